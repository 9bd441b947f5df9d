"""ctypes binding of oracle/libwt_oracle.so (and, when built, oracle/_ref/libwt_ref_frontend.so).

ORACLE — TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_long, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "libwt_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libwt_ref_frontend.so")

_fp, _ip64 = POINTER(c_float), POINTER(c_int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class FrontEnd:
    """The host-side stages; `prefix` selects the restatement ("wto_") or the reference's own
    functions compiled from /root/reference ("ref_")."""

    def __init__(self, path: str, prefix: str):
        self.lib = ctypes.CDLL(path)
        self.p = prefix
        L, p = self.lib, prefix
        g = lambda n: getattr(L, p + n)
        g("logmel").argtypes = [_fp, c_int, c_int, c_int, c_int, c_int, _fp, _fp]
        g("wav_read_legacy").argtypes = [c_char_p, _fp, c_long]
        g("wav_read_legacy").restype = c_long
        g("argmax_last").argtypes = [_fp, c_int64]
        g("argmax_last").restype = c_int64
        g("language_id").argtypes = [c_char_p]
        g("lang_code").argtypes = [c_int]
        g("lang_code").restype = c_char_p
        g("vocab_open").argtypes = [c_char_p, c_int]
        g("vocab_open").restype = c_void_p
        g("vocab_close").argtypes = [c_void_p]
        g("vocab_info").argtypes = [c_void_p, POINTER(c_int32)]
        g("vocab_filters_shape").argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32)]
        g("vocab_filters").argtypes = [c_void_p]
        g("vocab_filters").restype = _fp
        g("vocab_size").argtypes = [c_void_p]
        g("vocab_token").argtypes = [c_void_p, c_int, c_char_p, c_int]
        g("decode_text").argtypes = [c_void_p, _ip64, c_int, c_int, c_char_p, c_long]
        g("decode_text").restype = c_long
        g("remove_extra_spaces").argtypes = [c_char_p, c_char_p, c_long]
        g("remove_extra_spaces").restype = c_long

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    def logmel(self, pcm, filters, n_threads=4, fft_size=400, fft_step=160):
        pcm, filters = _f32(pcm).reshape(-1), _f32(filters)
        n_mel = filters.shape[0]
        out = np.zeros((n_mel, pcm.size // fft_step), np.float32)
        self._f("logmel")(pcm.ctypes.data_as(_fp), pcm.size, fft_size, fft_step, n_mel, n_threads,
                          filters.ctypes.data_as(_fp), out.ctypes.data_as(_fp))
        return out

    def wav_read_legacy(self, path):
        n = self._f("wav_read_legacy")(os.fsencode(path), None, 0)
        if n < 0:
            return np.zeros(0, np.float32)
        out = np.zeros(n, np.float32)
        self._f("wav_read_legacy")(os.fsencode(path), out.ctypes.data_as(_fp), n)
        return out

    def argmax_last(self, x):
        x = _f32(x).reshape(-1)
        return int(self._f("argmax_last")(x.ctypes.data_as(_fp), x.size))

    def language_id(self, code):
        return self._f("language_id")(code.encode())

    def lang_code(self, i):
        return self._f("lang_code")(i).decode()

    def language_count(self):
        return self._f("language_count")()

    def open_vocab(self, path, multilingual):
        return Vocab(self, path, multilingual)

    def remove_extra_spaces(self, s: str) -> str:
        buf = ctypes.create_string_buffer(len(s.encode()) + 8)
        n = self._f("remove_extra_spaces")(s.encode(), buf, len(buf))
        return buf.raw[:n].decode()


class Vocab:
    def __init__(self, fe: FrontEnd, path, multilingual):
        self.fe = fe
        self.h = fe._f("vocab_open")(os.fsencode(path), int(bool(multilingual)))
        if not self.h:
            raise FileNotFoundError(path)

    def close(self):
        if self.h:
            self.fe._f("vocab_close")(self.h)
            self.h = None

    def info(self):
        out = (c_int32 * 9)()
        self.fe._f("vocab_info")(self.h, out)
        keys = ("n_vocab", "eot", "sot", "translate", "transcribe", "prev", "solm", "not", "beg")
        return dict(zip(keys, list(out)))

    def filters(self):
        nm, nf = c_int32(0), c_int32(0)
        self.fe._f("vocab_filters_shape")(self.h, byref(nm), byref(nf))
        p = self.fe._f("vocab_filters")(self.h)
        return np.ctypeslib.as_array(p, shape=(nm.value, nf.value)).copy()

    def size(self):
        return self.fe._f("vocab_size")(self.h)

    def token(self, i):
        buf = ctypes.create_string_buffer(512)
        n = self.fe._f("vocab_token")(self.h, i, buf, len(buf))
        return None if n < 0 else buf.raw[:n]

    def decode(self, ids, omit_special=False):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        buf = ctypes.create_string_buffer(1 << 16)
        n = self.fe._f("decode_text")(self.h, ids.ctypes.data_as(_ip64), ids.size, int(omit_special), buf, len(buf))
        return None if n < 0 else buf.raw[:n]


def frontend() -> FrontEnd:
    return FrontEnd(ORACLE_LIB, "wto_")


def ref_frontend():
    """The reference's own front-end functions, or None where oracle/_ref was never built."""
    return FrontEnd(REF_LIB, "ref_") if os.path.exists(REF_LIB) else None


class Model:
    """CPU fp32 restatement of the encoder/decoder graphs + greedy loop (oracle/model.cpp)."""

    def __init__(self, wtw_path: str):
        L = ctypes.CDLL(ORACLE_LIB)
        L.wto_model_open.argtypes = [c_char_p]
        L.wto_model_open.restype = c_void_p
        L.wto_model_close.argtypes = [c_void_p]
        L.wto_model_dims.argtypes = [c_void_p, POINTER(c_int32)]
        L.wto_encode.argtypes = [c_void_p, _fp, _fp, c_int]
        L.wto_decode_greedy.argtypes = [c_void_p, _fp, _ip64, c_int, c_int, c_int64, c_int, c_int, c_int,
                                        _ip64, POINTER(c_int), _fp]
        L.wto_encdec_batch.argtypes = [c_void_p, _fp, c_int, _ip64, c_int, c_int, c_int64, c_int, c_int,
                                       c_int, _ip64, POINTER(c_int)]
        self.L = L
        self.h = L.wto_model_open(os.fsencode(wtw_path))
        if not self.h:
            raise FileNotFoundError(wtw_path)
        d = (c_int32 * 10)()
        L.wto_model_dims(self.h, d)
        keys = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")
        self.dims = dict(zip(keys, list(d)))

    def close(self):
        if self.h:
            self.L.wto_model_close(self.h)
            self.h = None

    def encode(self, mel, n_threads=8):
        mel = _f32(mel)
        out = np.zeros((self.dims["n_audio_ctx"], self.dims["n_audio_state"]), np.float32)
        self.L.wto_encode(self.h, mel.ctypes.data_as(_fp), out.ctypes.data_as(_fp), n_threads)
        return out

    def decode_greedy(self, enc_out, prompt, max_positions=30, eot=50257, stop_at_eot=True,
                      use_cache=True, n_threads=8, want_logits=False):
        enc_out = _f32(enc_out)
        prompt = np.ascontiguousarray(prompt, dtype=np.int64)
        ids = np.zeros(max_positions + 1, np.int64)
        n = c_int(0)
        steps_cap = max_positions - len(prompt) + 1
        logits = np.zeros((steps_cap, self.dims["n_vocab"]), np.float32) if want_logits else None
        steps = self.L.wto_decode_greedy(
            self.h, enc_out.ctypes.data_as(_fp), prompt.ctypes.data_as(_ip64), len(prompt), max_positions,
            eot, int(stop_at_eot), int(use_cache), n_threads, ids.ctypes.data_as(_ip64), byref(n),
            logits.ctypes.data_as(_fp) if logits is not None else None)
        return ids[: n.value].copy(), (logits[:steps] if logits is not None else None)

    def encdec_batch(self, mel, prompt, max_positions=30, eot=50257, stop_at_eot=True, use_cache=True,
                     n_threads=8):
        mel = _f32(mel)
        B = mel.shape[0]
        prompt = np.ascontiguousarray(prompt, dtype=np.int64)
        ids = np.zeros((B, max_positions + 1), np.int64)
        n = np.zeros(B, np.int32)
        self.L.wto_encdec_batch(self.h, mel.ctypes.data_as(_fp), B, prompt.ctypes.data_as(_ip64), len(prompt),
                                max_positions, eot, int(stop_at_eot), int(use_cache), n_threads,
                                ids.ctypes.data_as(_ip64), n.ctypes.data_as(POINTER(c_int)))
        return ids, n

"""whisper.tflite_amd — Python host mirror of the reference engine interface over the C ABI.

The product is the native library ``lib/libwhisper-tflite.so`` (hand-written gfx950
kernels behind ``include/wt_capi.h``).  This module only binds it with ctypes so that
tests, ``bench.py`` and ``__graft_entry__.py`` can drive it; it contains no compute and no
fallback: if the library is missing or no MI355X is present, calls fail loudly.

Interface mirrored (reference jerinphilip/whisper.tflite @ v2):
  * ``Engine.transcribe(samples)`` / ``Engine.transcribe(path)``  — whisper.h:159-163
  * ``create_engine(EngineType, model_prefix, vocab_path, multilingual)`` — whisper.h:259-260
  * ``EngineType`` — whisper.h:199-204

The directory is named like the reference's ``whisper.tflite/`` source directory, so it is
not importable with a plain ``import`` statement; load it with
``__graft_entry__.load_package()`` (importlib by path).
"""
from __future__ import annotations

import ctypes
import enum
import os
from ctypes import (POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_long, c_size_t,
                    c_uint64, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WT_LIB_PATH") or os.path.join(_HERE, "lib", "libwhisper-tflite.so")  # WT_LIB_PATH: A/B runs of two builds

WT_OK = 0
WT_MAX_IDS = 32
WT_PIPELINE_DEPTH = 24  # include/wt_capi.h
CHUNK_SAMPLES = 480000
STATUS_NAMES = {0: "WT_OK", 1: "WT_ERR_INVALID_ARG", 2: "WT_ERR_IO", 3: "WT_ERR_FORMAT",
                4: "WT_ERR_UNSUPPORTED", 5: "WT_ERR_DEVICE", 6: "WT_ERR_BUFFER"}

# Every symbol include/wt_capi.h and include/wt_debug.h declare.
CAPI_SYMBOLS = [
    "wt_device_alloc", "wt_device_free", "wt_device_upload", "wt_device_download", "wt_device_synchronize",
    "wt_engine_create", "wt_engine_destroy", "wt_last_error", "wt_engine_dims",
    "wt_engine_set_option", "wt_engine_get_option", "wt_engine_set_prompt", "wt_transcribe_pcm", "wt_transcribe_long_pcm", "wt_transcribe_file",
    "wt_logmel_batch", "wt_logmel_batch_dev", "wt_encdec_tokens_batch",
    "wt_encdec_tokens_batch_dev", "wt_transcribe_tokens_batch_dev", "wt_pipeline_submit_dev", "wt_pipeline_submit_pcm_dev", "wt_pipeline_collect",
    "wt_encdec_debug_batch",
    "wt_last_timings", "wt_last_kernel_stats", "wt_decode_text", "wt_language_id", "wt_lang_code", "wt_wav_read_legacy",
    "wt_vocab_info", "wt_filters", "wt_write_synthetic_weights", "wt_write_synthetic_vocab",
    "wt_vocab_open", "wt_vocab_close", "wt_vocab_get_info", "wt_vocab_get_filters", "wt_vocab_size", "wt_vocab_token",
    "wt_vocab_decode", "wt_log_mel_spectrogram", "wt_convert_tflite", "wt_shutdown",
]
DEBUG_SYMBOLS = [
    "wt_dbg_cross_absorbed", "wt_dbg_cross_absorbed_bf16", "wt_dbg_gemm_planes_ln", "wt_dbg_gemm", "wt_dbg_gemm_bench", "wt_dbg_dec_gemm_bench", "wt_dbg_dec_gemm", "wt_dbg_dec_ln_gemm", "wt_dbg_layernorm", "wt_dbg_encoder_attention",
    "wt_dbg_cross_attention", "wt_dbg_self_attention", "wt_dbg_interference", "wt_dbg_concurrency",
    "wt_dbg_gemm_planes", "wt_dbg_set_plane_gemm_mode", "wt_dbg_set_forced_ids", "wt_dbg_dec_gemm_bf16", "wt_dbg_dec_ln_gemm_bf16",
    "wt_dbg_self_attention_bf16", "wt_dbg_cross_attention_bf16", "wt_dbg_encoder_attention_planes", "wt_dbg_gemm_bf16", "wt_dbg_gemm_bf16_ln", "wt_dbg_encoder_attention_bf16",
]


class WtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {message}")
        self.code = code


class EngineType(enum.IntEnum):
    Monolith = 0
    EncDec = 1


class Dims(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class Timings(ctypes.Structure):
    _fields_ = [("logmel_ms", c_float), ("encoder_ms", c_float), ("cross_kv_ms", c_float),
                ("decoder_ms", c_float), ("total_ms", c_float), ("batch", c_int32),
                ("decoder_steps", c_int32)]


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", c_int32), ("reserved", c_int32),
                ("ms", ctypes.c_double), ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


_lib = None


def lib() -> ctypes.CDLL:
    """Loads the native library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()); "
                "there is no Python/CPU fallback for the engine")
        L = ctypes.CDLL(LIB_PATH)
        fp, ip64, ip32 = POINTER(c_float), POINTER(c_int64), POINTER(c_int32)
        L.wt_engine_create.argtypes = [c_int, c_char_p, c_char_p, c_int, c_int, POINTER(c_void_p)]
        L.wt_engine_destroy.argtypes = [c_void_p]
        L.wt_engine_destroy.restype = None
        L.wt_last_error.argtypes = [c_void_p]
        L.wt_last_error.restype = c_char_p
        L.wt_engine_dims.argtypes = [c_void_p, POINTER(Dims)]
        L.wt_engine_set_option.argtypes = [c_void_p, c_char_p, c_long]
        L.wt_engine_get_option.argtypes = [c_void_p, c_char_p, POINTER(c_long)]
        L.wt_engine_set_prompt.argtypes = [c_void_p, ip64, c_int]
        L.wt_transcribe_pcm.argtypes = [c_void_p, fp, c_size_t, c_char_p, c_size_t, POINTER(c_size_t)]
        L.wt_transcribe_long_pcm.argtypes = [c_void_p, fp, c_size_t, c_char_p, c_size_t, POINTER(c_size_t)]
        L.wt_transcribe_file.argtypes = [c_void_p, c_char_p, c_char_p, c_size_t, POINTER(c_size_t)]
        L.wt_logmel_batch.argtypes = [c_void_p, fp, c_int, fp]
        L.wt_logmel_batch_dev.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
        L.wt_encdec_tokens_batch.argtypes = [c_void_p, fp, c_int, ip64, ip32]
        L.wt_encdec_tokens_batch_dev.argtypes = [c_void_p, c_void_p, c_int, ip64, ip32]
        L.wt_transcribe_tokens_batch_dev.argtypes = [c_void_p, c_void_p, c_int, ip64, ip32]
        L.wt_pipeline_submit_dev.argtypes = [c_void_p, c_void_p, c_int]
        L.wt_pipeline_submit_pcm_dev.argtypes = [c_void_p, c_void_p, c_int]
        L.wt_pipeline_collect.argtypes = [c_void_p, ip64, ip32]
        L.wt_encdec_debug_batch.argtypes = [c_void_p, fp, c_int, ip64, ip32, fp, fp, c_int]
        L.wt_last_timings.argtypes = [c_void_p, POINTER(Timings)]
        L.wt_last_kernel_stats.argtypes = [c_void_p, POINTER(KernelStat), c_int]
        L.wt_decode_text.argtypes = [c_void_p, ip64, c_int, c_int, c_char_p, c_size_t, POINTER(c_size_t)]
        L.wt_language_id.argtypes = [c_char_p]
        L.wt_lang_code.argtypes = [c_int]
        L.wt_lang_code.restype = c_char_p
        L.wt_wav_read_legacy.argtypes = [c_char_p, fp, c_size_t, POINTER(c_size_t)]
        L.wt_vocab_info.argtypes = [c_void_p, ip32]
        L.wt_filters.argtypes = [c_void_p, fp, c_size_t, ip32, ip32]
        L.wt_write_synthetic_weights.argtypes = [c_char_p, c_char_p, c_uint64]
        L.wt_write_synthetic_vocab.argtypes = [c_char_p, c_int]
        L.wt_vocab_open.argtypes = [c_char_p, c_int, POINTER(c_void_p)]
        L.wt_vocab_close.argtypes = [c_void_p]
        L.wt_vocab_close.restype = None
        L.wt_vocab_get_info.argtypes = [c_void_p, ip32]
        L.wt_vocab_get_filters.argtypes = [c_void_p, fp, c_size_t, ip32, ip32]
        L.wt_vocab_size.argtypes = [c_void_p]
        L.wt_vocab_token.argtypes = [c_void_p, c_int, c_char_p, c_size_t, POINTER(c_size_t)]
        L.wt_vocab_decode.argtypes = [c_void_p, ip64, c_int, c_int, c_char_p, c_size_t, POINTER(c_size_t)]
        L.wt_log_mel_spectrogram.argtypes = [fp, c_int, fp, c_int, c_int, c_int, fp, c_size_t, POINTER(c_int)]
        L.wt_convert_tflite.argtypes = [c_char_p, c_char_p]
        L.wt_dbg_gemm.argtypes = [c_void_p, c_int, c_int, c_int, fp, fp, fp, fp, fp, c_int, c_int, fp]
        L.wt_dbg_gemm_planes.argtypes = [c_void_p, c_int, c_int, c_int, fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, fp,
                                         POINTER(c_float), c_int]
        L.wt_dbg_encoder_attention_planes.argtypes = [c_void_p, c_int, c_int, c_int, fp, c_int, fp, POINTER(c_float)]
        L.wt_dbg_cross_absorbed.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, fp, fp, c_int, POINTER(c_float)]
        L.wt_dbg_cross_absorbed_bf16.argtypes = L.wt_dbg_cross_absorbed.argtypes
        L.wt_dbg_gemm_planes_ln.argtypes = [c_void_p, c_int, c_int, fp, fp, fp, fp, fp, c_int, c_int, fp, fp, c_int, fp, fp, fp,
                                            POINTER(c_int)]
        L.wt_dbg_gemm_bf16.argtypes = [c_void_p, c_int, c_int, c_int, fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, fp,
                                       POINTER(c_float)]
        L.wt_dbg_encoder_attention_bf16.argtypes = [c_void_p, c_int, c_int, c_int, fp, c_int, fp, POINTER(c_float)]
        L.wt_dbg_gemm_bf16_ln.argtypes = [c_void_p, c_int, c_int, c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp, POINTER(c_int)]
        L.wt_dbg_gemm_bench.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float)]
        L.wt_dbg_dec_gemm_bench.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float)]
        L.wt_dbg_dec_gemm.argtypes = [c_void_p, c_int, c_int, c_int, c_int, fp, fp, fp, fp, fp, ip64]
        L.wt_dbg_dec_ln_gemm.argtypes = [c_void_p, c_int, c_int, c_int, fp, ip64, c_int, fp, fp,
                                         c_int, c_int, fp, fp, fp, fp, c_int, fp, fp]
        L.wt_dbg_layernorm.argtypes = [c_void_p, c_int, c_int, fp, fp, fp, fp]
        L.wt_device_alloc.argtypes = [c_void_p, ctypes.c_size_t, POINTER(c_void_p)]
        L.wt_device_free.argtypes = [c_void_p, c_void_p]
        L.wt_device_upload.argtypes = [c_void_p, c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t]
        L.wt_device_download.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_size_t, ctypes.c_size_t]
        L.wt_device_synchronize.argtypes = [c_void_p]
        L.wt_dbg_set_forced_ids.argtypes = [c_void_p, POINTER(c_int64), c_int]
        L.wt_dbg_dec_gemm_bf16.argtypes = L.wt_dbg_dec_gemm.argtypes
        L.wt_dbg_dec_ln_gemm_bf16.argtypes = L.wt_dbg_dec_ln_gemm.argtypes
        L.wt_dbg_encoder_attention.argtypes = [c_void_p, c_int, c_int, c_int, fp, fp]
        L.wt_dbg_cross_attention.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, fp, fp, fp, fp, fp]
        L.wt_dbg_self_attention.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, fp]
        L.wt_dbg_self_attention_bf16.argtypes = L.wt_dbg_self_attention.argtypes
        L.wt_dbg_cross_attention_bf16.argtypes = L.wt_dbg_cross_attention.argtypes
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(POINTER(c_float)) if a is not None else None


def write_synthetic_weights(path: str, arch: str = "tiny", seed: int = 0) -> None:
    rc = lib().wt_write_synthetic_weights(path.encode(), arch.encode(), seed)
    if rc != WT_OK:
        raise WtError(rc, lib().wt_last_error(None).decode())


def write_synthetic_vocab(path: str, n_tokens: int = 50257) -> None:
    rc = lib().wt_write_synthetic_vocab(path.encode(), n_tokens)
    if rc != WT_OK:
        raise WtError(rc, lib().wt_last_error(None).decode())


def language_id(code: str) -> int:
    return lib().wt_language_id(code.encode())


def lang_code(idx: int) -> str:
    return lib().wt_lang_code(idx).decode()


def wav_read_legacy(path: str) -> np.ndarray:
    n = c_size_t(0)
    rc = lib().wt_wav_read_legacy(path.encode(), None, 0, byref(n))
    if rc != WT_OK:
        return np.zeros(0, np.float32)  # the reference returns an empty vector
    out = np.zeros(n.value, np.float32)
    lib().wt_wav_read_legacy(path.encode(), _fp(out), out.size, byref(n))
    return out


class Vocab:
    """Host-side mirror of the reference's ``Vocab`` + ``Filters`` as ``Reader::read`` fills them
    (whisper.h:44-101, :236-248) and of ``decode`` (whisper.h:252-257).  Needs no GPU."""

    INFO_KEYS = ("n_vocab", "eot", "sot", "translate", "transcribe", "prev", "solm", "not", "beg")

    def __init__(self, vocab_path: str, multilingual: bool = True):
        self._v = c_void_p()
        rc = lib().wt_vocab_open(os.fsencode(vocab_path), int(bool(multilingual)), byref(self._v))
        if rc != WT_OK:
            self._v = c_void_p()
            raise WtError(rc, lib().wt_last_error(None).decode())

    def close(self) -> None:
        if getattr(self, "_v", None) and self._v.value:
            lib().wt_vocab_close(self._v)
            self._v = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        out = (c_int32 * 9)()
        lib().wt_vocab_get_info(self._v, out)
        return dict(zip(self.INFO_KEYS, list(out)))

    def filters(self) -> np.ndarray:
        nm, nf = c_int32(0), c_int32(0)
        total = lib().wt_vocab_get_filters(self._v, None, 0, byref(nm), byref(nf))
        out = np.zeros(total, np.float32)
        lib().wt_vocab_get_filters(self._v, _fp(out), out.size, byref(nm), byref(nf))
        return out.reshape(nm.value, nf.value)

    def size(self) -> int:
        return lib().wt_vocab_size(self._v)

    def token(self, idx: int) -> bytes:
        buf = ctypes.create_string_buffer(512)
        n = c_size_t(0)
        rc = lib().wt_vocab_token(self._v, int(idx), buf, len(buf), byref(n))
        if rc != WT_OK:
            raise WtError(rc, lib().wt_last_error(None).decode())
        return buf.raw[: n.value]

    def decode(self, ids, omit_special_tokens: bool = False) -> bytes:
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        buf = ctypes.create_string_buffer(1 << 16)
        n = c_size_t(0)
        rc = lib().wt_vocab_decode(self._v, ids.ctypes.data_as(POINTER(c_int64)), ids.size, int(omit_special_tokens),
                                   buf, len(buf), byref(n))
        if rc != WT_OK:
            raise WtError(rc, lib().wt_last_error(None).decode())
        return buf.raw[: n.value]


def log_mel_spectrogram(samples, filters, device_id: int = 0) -> np.ndarray:
    """Mirror of the free function ``whisper::log_mel_spectrogram`` (whisper.h:123): [n_mel][n_samples // 160]."""
    pcm = _f32(samples).reshape(-1)
    f = _f32(filters)
    n_len = c_int(0)
    out = np.zeros((f.shape[0], pcm.size // 160), np.float32)
    rc = lib().wt_log_mel_spectrogram(_fp(pcm), pcm.size, _fp(f), f.shape[0], f.shape[1], device_id, _fp(out),
                                      out.size, byref(n_len))
    if rc != WT_OK:
        raise WtError(rc, lib().wt_last_error(None).decode())
    return out


def convert_tflite(model_prefix: str, out_path: str) -> None:
    rc = lib().wt_convert_tflite(os.fsencode(model_prefix), os.fsencode(out_path))
    if rc != WT_OK:
        raise WtError(rc, lib().wt_last_error(None).decode())


class Engine:
    """Mirror of ``whisper::Engine`` / ``whisper::EncDec`` / ``whisper::Monolith`` (reference whisper.h:159-197)."""

    def __init__(self, model_prefix: str, vocab_path: str, multilingual: bool = True,
                 engine_type: EngineType = EngineType.EncDec, device_id: int = 0):
        self._h = c_void_p()
        rc = lib().wt_engine_create(int(engine_type), model_prefix.encode(), vocab_path.encode(),
                                    int(bool(multilingual)), device_id, byref(self._h))
        if rc != WT_OK:
            self._h = c_void_p()
            raise WtError(rc, lib().wt_last_error(None).decode())
        d = Dims()
        lib().wt_engine_dims(self._h, byref(d))
        self.dims = d

    # -- lifecycle -------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            lib().wt_engine_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != WT_OK:
            raise WtError(rc, lib().wt_last_error(self._h).decode())

    @property
    def handle(self) -> c_void_p:
        return self._h

    def set_option(self, key: str, value: int) -> None:
        self._check(lib().wt_engine_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = c_long(0)
        self._check(lib().wt_engine_get_option(self._h, key.encode(), byref(v)))
        return v.value

    def set_prompt(self, ids) -> None:
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        self._check(lib().wt_engine_set_prompt(self._h, ids.ctypes.data_as(POINTER(c_int64)), ids.size))

    # -- shapes ----------------------------------------------------------------------
    @property
    def mel_shape(self):
        return (self.dims.n_mels, 2 * self.dims.n_audio_ctx)

    @property
    def pcm_len(self) -> int:
        return 2 * self.dims.n_audio_ctx * 160

    # -- the reference's two virtuals ------------------------------------------------
    def transcribe(self, samples_or_path) -> str:
        buf = ctypes.create_string_buffer(16384)
        n = c_size_t(0)
        if isinstance(samples_or_path, (str, bytes, os.PathLike)):
            path = os.fsencode(samples_or_path)
            self._check(lib().wt_transcribe_file(self._h, path, buf, len(buf), byref(n)))
        else:
            pcm = _f32(samples_or_path).reshape(-1)
            self._check(lib().wt_transcribe_pcm(self._h, _fp(pcm), pcm.size, buf, len(buf), byref(n)))
        return buf.raw[: n.value].decode("utf-8", errors="replace")

    def transcribe_long(self, samples) -> str:
        """Long audio: consecutive 30 s windows, batched; per-window texts joined with '\\n'."""
        pcm = _f32(samples).reshape(-1)
        buf = ctypes.create_string_buffer(1 << 20)
        n = c_size_t(0)
        self._check(lib().wt_transcribe_long_pcm(self._h, _fp(pcm), pcm.size, buf, len(buf), byref(n)))
        return buf.raw[: n.value].decode("utf-8", errors="replace")

    # -- batch entry points ----------------------------------------------------------
    def logmel_batch(self, pcm) -> np.ndarray:
        pcm = _f32(pcm).reshape(-1, self.pcm_len)
        mel = np.empty((pcm.shape[0],) + self.mel_shape, np.float32)
        self._check(lib().wt_logmel_batch(self._h, _fp(pcm), pcm.shape[0], _fp(mel)))
        return mel

    def encdec_tokens_batch(self, mel):
        mel = _f32(mel).reshape((-1,) + self.mel_shape)
        B = mel.shape[0]
        ids = np.zeros((B, WT_MAX_IDS), np.int64)
        n = np.zeros(B, np.int32)
        self._check(lib().wt_encdec_tokens_batch(
            self._h, _fp(mel), B, ids.ctypes.data_as(POINTER(c_int64)), n.ctypes.data_as(POINTER(c_int32))))
        return ids, n

    def encdec_tokens_batch_dev(self, d_mel_ptr: int, batch: int):
        ids = np.zeros((batch, WT_MAX_IDS), np.int64)
        n = np.zeros(batch, np.int32)
        self._check(lib().wt_encdec_tokens_batch_dev(
            self._h, c_void_p(d_mel_ptr), batch, ids.ctypes.data_as(POINTER(c_int64)),
            n.ctypes.data_as(POINTER(c_int32))))
        return ids, n

    def transcribe_tokens_batch_dev(self, d_pcm_ptr: int, batch: int):
        ids = np.zeros((batch, WT_MAX_IDS), np.int64)
        n = np.zeros(batch, np.int32)
        self._check(lib().wt_transcribe_tokens_batch_dev(
            self._h, c_void_p(d_pcm_ptr), batch, ids.ctypes.data_as(POINTER(c_int64)),
            n.ctypes.data_as(POINTER(c_int32))))
        return ids, n

    def pipeline_submit_dev(self, d_mel_ptr: int, batch: int) -> None:
        self._check(lib().wt_pipeline_submit_dev(self._h, c_void_p(d_mel_ptr), batch))
        self._submitted = getattr(self, "_submitted", [])
        self._submitted.append(batch)

    def pipeline_submit_pcm_dev(self, d_pcm_ptr: int, batch: int) -> None:
        self._check(lib().wt_pipeline_submit_pcm_dev(self._h, c_void_p(d_pcm_ptr), batch))
        self._submitted = getattr(self, "_submitted", [])
        self._submitted.append(batch)

    def pipeline_collect(self):
        batch = self._submitted.pop(0)
        ids = np.zeros((batch, WT_MAX_IDS), np.int64)
        n = np.zeros(batch, np.int32)
        self._check(lib().wt_pipeline_collect(
            self._h, ids.ctypes.data_as(POINTER(c_int64)), n.ctypes.data_as(POINTER(c_int32))))
        return ids, n

    def logmel_batch_dev(self, d_pcm_ptr: int, batch: int, d_mel_ptr: int) -> None:
        self._check(lib().wt_logmel_batch_dev(self._h, c_void_p(d_pcm_ptr), batch, c_void_p(d_mel_ptr)))

    def encdec_debug_batch(self, mel, want_enc_out=True, want_logits=True, steps_cap=27):
        mel = _f32(mel).reshape((-1,) + self.mel_shape)
        B = mel.shape[0]
        ids = np.zeros((B, WT_MAX_IDS), np.int64)
        n = np.zeros(B, np.int32)
        enc = np.zeros((B, self.dims.n_audio_ctx, self.dims.n_audio_state), np.float32) if want_enc_out else None
        logits = np.zeros((B, steps_cap, self.dims.n_vocab), np.float32) if want_logits else None
        self._check(lib().wt_encdec_debug_batch(
            self._h, _fp(mel), B, ids.ctypes.data_as(POINTER(c_int64)), n.ctypes.data_as(POINTER(c_int32)),
            _fp(enc) if enc is not None else None, _fp(logits) if logits is not None else None, steps_cap))
        return ids, n, enc, logits

    def timings(self) -> Timings:
        t = Timings()
        self._check(lib().wt_last_timings(self._h, byref(t)))
        return t

    def kernel_stats(self) -> dict:
        arr = (KernelStat * 8)()
        n = lib().wt_last_kernel_stats(self._h, arr, 8)
        return {arr[i].name.decode(): {"launches": arr[i].launches, "ms": arr[i].ms, "flops": arr[i].flops,
                                       "bytes": arr[i].bytes} for i in range(min(n, 8))}

    # -- host helpers ----------------------------------------------------------------
    def decode_bytes(self, ids, omit_special_tokens: bool = False) -> bytes:
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        buf = ctypes.create_string_buffer(16384)
        n = c_size_t(0)
        self._check(lib().wt_decode_text(self._h, ids.ctypes.data_as(POINTER(c_int64)), ids.size,
                                         int(omit_special_tokens), buf, len(buf), byref(n)))
        return buf.raw[: n.value]

    def decode_text(self, ids, omit_special_tokens: bool = False) -> str:
        return self.decode_bytes(ids, omit_special_tokens).decode("utf-8", errors="replace")

    def vocab_info(self) -> dict:
        out = (c_int32 * 9)()
        self._check(lib().wt_vocab_info(self._h, out))
        keys = ("n_vocab", "eot", "sot", "translate", "transcribe", "prev", "solm", "not", "beg")
        return dict(zip(keys, list(out)))

    def filters(self) -> np.ndarray:
        nm, nf = c_int32(0), c_int32(0)
        total = lib().wt_filters(self._h, None, 0, byref(nm), byref(nf))
        out = np.zeros(total, np.float32)
        lib().wt_filters(self._h, _fp(out), out.size, byref(nm), byref(nf))
        return out.reshape(nm.value, nf.value)

    # -- kernel-level taps (include/wt_debug.h) --------------------------------------
    def dbg_gemm(self, A, W, bias=None, R=None, pos=None, epi=0):
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        N = W.shape[0]
        C = np.zeros((M, N), np.float32)
        bias = _f32(bias) if bias is not None else None
        R = _f32(R) if R is not None else None
        pos = _f32(pos) if pos is not None else None
        self._check(lib().wt_dbg_gemm(self._h, M, N, K, _fp(A), _fp(W), _fp(bias), _fp(R), _fp(pos),
                                      pos.shape[0] if pos is not None else 0, epi, _fp(C)))
        return C

    def dbg_gemm_planes(self, A, W, bias=None, R=None, pos=None, epi=1, planes_out=False, iters=0, n_cu=0):
        """The default encoder GEMM (fp16 planes).  Returns C, or (C, ms per launch) when iters > 0."""
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        N = W.shape[0]
        C = np.zeros((M, N), np.float32)
        bias = _f32(bias) if bias is not None else np.zeros(N, np.float32)
        R = _f32(R) if R is not None else None
        pos = _f32(pos) if pos is not None else None
        ms = c_float(0)
        self._check(lib().wt_dbg_gemm_planes(self._h, M, N, K, _fp(A), _fp(W), _fp(bias), _fp(R), _fp(pos),
                                             pos.shape[0] if pos is not None else 0, epi, int(planes_out), iters, _fp(C),
                                             byref(ms), int(n_cu)))
        return (C, ms.value) if iters > 0 else C

    def dbg_gemm_planes_ln(self, A, W, bias, ln_g, ln_b, R=None, pos=None, epi=5, n_cu=0, want_y32=True):
        """The plane GEMM (N = 384) with the LayerNorm of its output rows fused into the epilogue.  Returns
        (C, ln_out, ln_y32, fused)."""
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        assert W.shape[0] == 384
        C = np.zeros((M, 384), np.float32)
        ln = np.zeros((M, 384), np.float32)
        y32 = np.zeros((M, 384), np.float32) if want_y32 else None
        R = _f32(R) if R is not None else None
        pos = _f32(pos) if pos is not None else None
        fused = ctypes.c_int(0)
        self._check(lib().wt_dbg_gemm_planes_ln(self._h, M, K, _fp(A), _fp(W), _fp(_f32(bias)), _fp(R), _fp(pos),
                                                pos.shape[0] if pos is not None else 0, epi, _fp(_f32(ln_g)), _fp(_f32(ln_b)),
                                                int(n_cu), _fp(C), _fp(ln), _fp(y32) if y32 is not None else None, byref(fused)))
        return C, ln, y32, bool(fused.value)

    def dbg_cross_absorbed(self, qp, E, wv, bv, batch, heads, T, chunks, nq, iters=0, bf16=False):
        """Absorbed cross-attention + chunk combine + value projection: qp [nq * batch][heads * d], E [batch][T][d],
        wv [d][d], bv [d] -> [nq * batch][d] (and the average microseconds of the attention launch when iters > 0)"""
        qp, E, wv, bv = _f32(qp), _f32(E), _f32(wv), _f32(bv)
        d = heads * 64
        out = np.zeros((nq * batch, d), np.float32)
        us = c_float(0)
        fn = lib().wt_dbg_cross_absorbed_bf16 if bf16 else lib().wt_dbg_cross_absorbed
        self._check(fn(self._h, batch, heads, T, chunks, nq, _fp(qp), _fp(E), _fp(wv), _fp(bv), _fp(out), iters, byref(us)))
        return (out, us.value) if iters > 0 else out

    def dbg_encoder_attention_planes(self, qkv, batch, T, heads, iters=0):
        qkv = _f32(qkv)
        out = np.zeros((batch * T, heads * 64), np.float32)
        ms = c_float(0)
        self._check(lib().wt_dbg_encoder_attention_planes(self._h, batch, T, heads, _fp(qkv), iters, _fp(out), byref(ms)))
        return (out, ms.value) if iters > 0 else out

    def dbg_gemm_bf16(self, A, W, bias=None, R=None, pos=None, epi=1, bf16_out=False, iters=0):
        """The encoder GEMM of the bf16 storage mode (operands rounded to bf16 on the host side of the tap)."""
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        N = W.shape[0]
        C = np.zeros((M, N), np.float32)
        bias = _f32(bias) if bias is not None else np.zeros(N, np.float32)
        R = _f32(R) if R is not None else None
        pos = _f32(pos) if pos is not None else None
        ms = c_float(0)
        self._check(lib().wt_dbg_gemm_bf16(self._h, M, N, K, _fp(A), _fp(W), _fp(bias), _fp(R), _fp(pos),
                                           pos.shape[0] if pos is not None else 0, epi, int(bf16_out), iters, _fp(C),
                                           byref(ms)))
        return (C, ms.value) if iters > 0 else C

    def dbg_encoder_attention_bf16(self, qkv, batch, T, heads, iters=0):
        qkv = _f32(qkv)
        out = np.zeros((batch * T, heads * 64), np.float32)
        ms = c_float(0)
        self._check(lib().wt_dbg_encoder_attention_bf16(self._h, batch, T, heads, _fp(qkv), iters, _fp(out), byref(ms)))
        return (out, ms.value) if iters > 0 else out

    def dbg_gemm_bench(self, M, N, K, epi=1, variant=0, iters=10) -> float:
        ms = c_float(0)
        self._check(lib().wt_dbg_gemm_bench(self._h, M, N, K, epi, variant, iters, byref(ms)))
        return ms.value

    def device_array(self, a):
        """A copy of the numpy array `a` resident in this engine's HBM (wt_device_alloc + upload): the caller-owned input
        of the *_dev entry points.  Returns a DeviceArray (data_ptr(), download(), free())."""
        return DeviceArray(self, a)

    def device_synchronize(self):
        self._check(lib().wt_device_synchronize(self._h))

    def set_forced_ids(self, ids=None):
        """Teacher forcing (test tap): ids [clips][32] every following decode of that many clips follows instead of its
        own argmax; None switches it off."""
        if ids is None:
            self._check(lib().wt_dbg_set_forced_ids(self._h, None, 0))
            return
        a = np.ascontiguousarray(ids, dtype=np.int64)
        assert a.ndim == 2 and a.shape[1] == 32
        self._check(lib().wt_dbg_set_forced_ids(self._h, a.ctypes.data_as(POINTER(c_int64)), a.shape[0]))

    def dbg_gemm_bf16_ln(self, A, W, bias, R, ln_g, ln_b):
        """x = R + bias + A . W^T (bf16 operands) with the fused LayerNorm: returns (x, LayerNorm plane as float, fp32 LayerNorm, fused)"""
        A, W, bias, R, ln_g, ln_b = (_f32(v) for v in (A, W, bias, R, ln_g, ln_b))
        M, K = A.shape
        N = W.shape[0]
        C = np.zeros((M, N), np.float32)
        ln = np.zeros((M, N), np.float32)
        y32 = np.zeros((M, N), np.float32)
        fused = c_int(0)
        self._check(lib().wt_dbg_gemm_bf16_ln(self._h, M, N, K, _fp(A), _fp(W), _fp(bias), _fp(R), _fp(ln_g), _fp(ln_b), _fp(C), _fp(ln),
                                              _fp(y32), byref(fused)))
        return C, ln, y32, bool(fused.value)

    def dbg_dec_gemm_bench(self, kind, B, N, K, rows=None, iters=200) -> float:
        us = c_float(0)
        self._check(lib().wt_dbg_dec_gemm_bench(self._h, kind, B, N, K, rows or B, iters, byref(us)))
        return us.value

    def dbg_dec_gemm(self, X, W, bias=None, mode=0, R=None, bf16=False):
        """mode 0 bias, 1 bias+gelu, 2 residual (Y = R + bias + X.W^T), 3 logits + argmax.  bf16: the bf16 storage
        mode's instantiation (weights one bf16 plane, activations rounded to bf16 in registers)."""
        X, W = _f32(X), _f32(W)
        B, K = X.shape
        N = W.shape[0]
        bias = _f32(bias) if bias is not None else np.zeros(N, np.float32)
        R = _f32(R) if R is not None else None
        Y = np.zeros((B, N), np.float32)
        am = np.zeros(B, np.int64)
        fn = lib().wt_dbg_dec_gemm_bf16 if bf16 else lib().wt_dbg_dec_gemm
        self._check(fn(self._h, mode, B, N, K, _fp(X), _fp(W), _fp(bias), _fp(R), _fp(Y), am.ctypes.data_as(POINTER(c_int64))))
        return (Y, am) if mode == 3 else Y

    def dbg_dec_ln_gemm(self, W, bias, ln_g, ln_b, xin=None, ids=None, pos=0, tok_emb=None, pos_emb=None,
                        gelu=False, bf16=False):
        W, bias, ln_g, ln_b = _f32(W), _f32(bias), _f32(ln_g), _f32(ln_b)
        N, K = W.shape
        xin = _f32(xin) if xin is not None else None
        ids_a = np.ascontiguousarray(ids, dtype=np.int64) if ids is not None else None
        tok_emb = _f32(tok_emb) if tok_emb is not None else None
        pos_emb = _f32(pos_emb) if pos_emb is not None else None
        B = xin.shape[0] if xin is not None else ids_a.shape[0]
        Y = np.zeros((B, N), np.float32)
        xout = np.zeros((B, K), np.float32)
        fn = lib().wt_dbg_dec_ln_gemm_bf16 if bf16 else lib().wt_dbg_dec_ln_gemm
        self._check(fn(
            self._h, B, N, K, _fp(xin),
            ids_a.ctypes.data_as(POINTER(c_int64)) if ids_a is not None else None, pos, _fp(tok_emb), _fp(pos_emb),
            tok_emb.shape[0] if tok_emb is not None else 0, pos_emb.shape[0] if pos_emb is not None else 0,
            _fp(ln_g), _fp(ln_b), _fp(W), _fp(bias), int(gelu), _fp(Y), _fp(xout)))
        return Y, xout

    def dbg_layernorm(self, x, g, b):
        x, g, b = _f32(x), _f32(g), _f32(b)
        y = np.zeros_like(x)
        self._check(lib().wt_dbg_layernorm(self._h, x.shape[0], x.shape[1], _fp(x), _fp(g), _fp(b), _fp(y)))
        return y

    def dbg_encoder_attention(self, qkv, batch, T, heads):
        qkv = _f32(qkv)
        out = np.zeros((batch * T, heads * 64), np.float32)
        self._check(lib().wt_dbg_encoder_attention(self._h, batch, T, heads, _fp(qkv), _fp(out)))
        return out

    def dbg_cross_attention(self, x, ln_g, ln_b, wq, bq, kc, vc, chunks=2, nq=1, bf16=False):
        """x [nq*B][d] residual rows (row = p * B + b) -> attention output [nq*B][d]; the query projection
        q = LayerNorm(x) . wq^T + bq runs inside the kernel."""
        x, ln_g, ln_b, wq, bq, kc, vc = (_f32(a) for a in (x, ln_g, ln_b, wq, bq, kc, vc))
        B, H, T, _ = kc.shape
        out = np.zeros((nq * B, H * 64), np.float32)
        fn = lib().wt_dbg_cross_attention_bf16 if bf16 else lib().wt_dbg_cross_attention
        self._check(fn(self._h, B, H, T, chunks, nq, _fp(x), _fp(ln_g), _fp(ln_b), _fp(wq), _fp(bq), _fp(kc), _fp(vc), _fp(out)))
        return out

    def dbg_self_attention(self, qkv, kcache, vcache, pos, npos=1, bf16=False):
        qkv, kcache, vcache = _f32(qkv), _f32(kcache).copy(), _f32(vcache).copy()
        B, cap, d = kcache.shape
        out = np.zeros((npos * B, d), np.float32)
        fn = lib().wt_dbg_self_attention_bf16 if bf16 else lib().wt_dbg_self_attention
        self._check(fn(self._h, B, d // 64, cap, pos, npos, _fp(qkv), _fp(kcache), _fp(vcache), _fp(out)))
        return out, kcache, vcache


class DeviceArray:
    """Device buffer owned by the caller, allocated through the engine's C ABI (no HIP runtime on the Python side)."""

    def __init__(self, eng, a):
        a = np.ascontiguousarray(a)
        self._eng, self.shape, self.dtype, self.nbytes = eng, a.shape, a.dtype, a.nbytes
        p = c_void_p()
        eng._check(lib().wt_device_alloc(eng._h, a.nbytes, byref(p)))
        self._p = p
        eng._check(lib().wt_device_upload(eng._h, p, 0, a.ctypes.data_as(c_void_p), a.nbytes))

    def data_ptr(self) -> int:
        return self._p.value

    def download(self):
        out = np.empty(self.shape, self.dtype)
        self._eng._check(lib().wt_device_download(self._eng._h, out.ctypes.data_as(c_void_p), self._p, 0, self.nbytes))
        return out

    def free(self):
        if self._p:
            self._eng._check(lib().wt_device_free(self._eng._h, self._p))
            self._p = None


def create_engine(engine_type, model_prefix: str, vocab_path: str, multilingual: bool):
    """Mirror of ``whisper::create_engine`` (reference whisper.cpp:778-790): returns None
    (after a message on stderr) for an unknown or unsupported engine type."""
    import sys
    try:
        et = EngineType(int(engine_type))
    except ValueError:
        print("Unknown engine-type", file=sys.stderr)
        return None
    return Engine(model_prefix, vocab_path, multilingual, et)

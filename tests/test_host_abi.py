"""CPU: the C-ABI library loads without a GPU, exports every declared symbol, and its
host-side helpers (no device compute) agree with the oracle / reference goldens."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLD, ROOT


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wt_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    declared = _declared("wt_capi.h") + _declared("wt_debug.h")
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/ but not exported"
    assert sorted(pkg.CAPI_SYMBOLS) == _declared("wt_capi.h")
    assert sorted(pkg.DEBUG_SYMBOLS) == _declared("wt_debug.h")


def test_no_cpu_fallback_without_gpu(pkg, assets):
    """Creating an engine needs a gfx950 device; without one the ABI reports WT_ERR_DEVICE
    instead of computing anywhere else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    prefix, vocab = assets("micro")
    with pytest.raises(pkg.WtError) as e:
        pkg.Engine(prefix, vocab, True)
    assert e.value.code == 5  # WT_ERR_DEVICE


def test_error_conventions(pkg, assets, tmp_path):
    prefix, vocab = assets("micro")
    h = ctypes.c_void_p()
    L = pkg.lib()
    assert L.wt_engine_create(7, prefix.encode(), vocab.encode(), 1, 0, ctypes.byref(h)) == 1
    assert not h.value and b"Unknown engine-type" in L.wt_last_error(None)
    # missing vocab file -> IO error with the reference's message (mmap_file.cpp:16), for both engine types
    for et in (0, 1):
        rc = L.wt_engine_create(et, prefix.encode(), str(tmp_path / "nope.bin").encode(), 1, 0, ctypes.byref(h))
        assert rc == 2 and L.wt_last_error(None).startswith(b"Failed to open file")
    assert pkg.create_engine(9, prefix, vocab, True) is None
    L.wt_engine_destroy(None)  # no-op
    L.wt_vocab_close(None)
    v = ctypes.c_void_p()
    assert L.wt_vocab_open(str(tmp_path / "nope.bin").encode(), 1, ctypes.byref(v)) == 2 and not v.value
    (tmp_path / "short.bin").write_bytes(b"\0" * 20)
    assert L.wt_vocab_open(str(tmp_path / "short.bin").encode(), 1, ctypes.byref(v)) == 3 and not v.value
    # the free-function front end refuses geometries the kernels do not implement, without touching a device
    f = np.zeros((80, 201), np.float32)
    n_len = ctypes.c_int(0)
    out = np.zeros(10, np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = L.wt_log_mel_spectrogram(out.ctypes.data_as(fp), 10, f.ctypes.data_as(fp), 64, 201, 0, out.ctypes.data_as(fp), 10,
                                  ctypes.byref(n_len))
    assert rc == 4  # WT_ERR_UNSUPPORTED


def test_product_vocab_reader_and_decode_match_reference_golden(pkg, tmp_path):
    """Rows a3 / a7 on the PRODUCT side: the library's own reader (csrc/host_util.cpp parse_vocab) and
    decode_tokens, through the host-only wt_vocab_* entry points, against the vectors the reference's own
    Reader / decode produced (tests/golden/frontend_host.npz, generator tools/gen_golden.py;
    reference whisper.cpp:519-665, :218-226)."""
    g = np.load(os.path.join(GOLD, "frontend_host.npz"))
    path = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(path, 300)  # the file the goldens were made from
    v = pkg.Vocab(path, True)
    info = v.info()
    assert [info[k] for k in pkg.Vocab.INFO_KEYS] == list(g["info"])
    f = v.filters()
    assert f.shape == (80, 201) and np.array_equal(f.view(np.uint32), g["filters"].view(np.uint32))
    for i, b in zip(g["tok_ids"], g["tok_bytes"]):
        assert v.token(int(i)) == bytes(b), i
    assert v.size() == 51865
    assert v.decode(g["decode_ids"], False) == g["decode_text"].tobytes()
    assert v.decode(g["decode_ids"], True) == g["decode_text_omit"].tobytes()
    with pytest.raises(pkg.WtError):
        v.token(60000)
    with pytest.raises(pkg.WtError):
        v.decode([60000], False)
    v.close()
    en = pkg.Vocab(path, False)
    assert en.info()["eot"] == 50256 and en.info()["sot"] == 50257 and en.info()["not"] == 50362
    assert en.info()["n_vocab"] == 300 and en.size() == 51864
    assert en.token(50257) == b"<|startoftranscript_|>" and en.token(50362) == b"<|notimestamps|>"
    en.close()


def test_product_reader_agrees_with_oracle_on_every_token(pkg, orc, tmp_path):
    """Every id -> token of the product reader equals the reference-pinned oracle's, multilingual and English."""
    path = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(path, 1000)
    for multilingual in (True, False):
        a, b = pkg.Vocab(path, multilingual), orc.frontend().open_vocab(path, multilingual)
        assert a.size() == b.size()
        for i in list(range(0, 1000, 37)) + list(range(50250, a.size())):
            assert a.token(i) == b.token(i), (multilingual, i)
        a.close()
        b.close()


def test_language_table_matches_oracle(pkg, orc):
    fe = orc.frontend()
    for i in range(100):
        assert pkg.lang_code(i) == fe.lang_code(i)
        assert pkg.language_id(fe.lang_code(i)) == i
    assert pkg.language_id("de") == 2 and pkg.language_id("zz") == 100


def test_wav_reader_matches_reference_golden(pkg, tmp_path):
    g = np.load(os.path.join(GOLD, "frontend_host.npz"))
    p = tmp_path / "ramp.wav"
    p.write_bytes(g["wav_bytes"].tobytes())
    s = pkg.wav_read_legacy(str(p))
    assert np.array_equal(s.view(np.uint32), g["wav_samples"].view(np.uint32))
    assert len(pkg.wav_read_legacy(str(tmp_path / "missing.wav"))) == 0


def test_synthetic_vocab_file_round_trip(pkg, orc, tmp_path):
    """The asset writer produces the reference layout: the oracle's (reference-pinned) reader
    parses it, and the Slaney bank equals the HuggingFace/librosa one."""
    path = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(path, 1000)
    v = orc.frontend().open_vocab(path, True)
    f = v.filters()
    assert f.shape == (80, 201) and v.info()["n_vocab"] == 51865
    assert v.token(65) == b"A" and v.token(999) == b" t999"
    v.close()
    from transformers.audio_utils import mel_filter_bank
    hf = mel_filter_bank(num_frequency_bins=201, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                         sampling_rate=16000, norm="slaney", mel_scale="slaney").T
    assert np.abs(f - hf).max() < 1e-7
    raw = open(path, "rb").read()
    assert int.from_bytes(raw[:8], "little") == len(raw) - 8 and raw[8:12] == b"NESU"


def test_synthetic_weights_are_deterministic(pkg, tmp_path):
    from wtw import read_wtw
    a, b = str(tmp_path / "a.wtw"), str(tmp_path / "b.wtw")
    pkg.write_synthetic_weights(a, "micro", 3)
    pkg.write_synthetic_weights(b, "micro", 3)
    assert open(a, "rb").read() == open(b, "rb").read()
    dims, t = read_wtw(a)
    assert dims["n_audio_ctx"] == 100 and t["decoder.token_embedding.weight"].shape == (1024, 128)
    w = t["encoder.blocks.0.attn.query.weight"]
    assert abs(float(w.std()) - 128 ** -0.5) < 0.01 and abs(float(w.mean())) < 0.01
    assert abs(float(t["encoder.blocks.1.attn_ln.weight"].mean()) - 1.0) < 0.05
    with pytest.raises(pkg.WtError):
        pkg.write_synthetic_weights(a, "huge", 0)


def test_three_plane_bf16_split_is_exact():
    """The identity behind gemm_split_tile / encoder_attention_split (csrc/bf16_split.h), restated in
    numpy: h1 = x rounded to bf16 (half away from zero), h2 = (x - h1) truncated to bf16,
    h3 = x - h1 - h2.  The split leaves no remainder, every plane is a bf16 value, and the six plane
    products kept by the kernels miss a.b by < 2^-22 |a||b| (the dropped terms are a2b3 + a3b2 + a3b3
    with |a2| <= 2^-8 |a|, |a3| < 2^-15 |a|).  Holds for 2^-100 < |x| < 3.3e38."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-20, 20, 20000),
                        [0.0, 1.0, -1.0, 3.3e38, 1.0e-30, 1.0000001, 0.33333334, 1.00390625, 255.5]]).astype(np.float32)

    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    h1 = ((x.view(np.uint32) + np.uint32(0x8000)) & np.uint32(0xFFFF0000)).view(np.float32)
    r1 = x - h1
    h2 = trunc(r1)
    h3 = r1 - h2
    assert np.array_equal(trunc(h3), h3)  # the third plane needs no rounding
    assert np.array_equal(h1.astype(np.float64) + h2.astype(np.float64) + h3.astype(np.float64), x.astype(np.float64))
    a, b = x[:10000].astype(np.float64), x[10000:20000].astype(np.float64)
    pa = [v[:10000].astype(np.float64) for v in (h1, h2, h3)]
    pb = [v[10000:20000].astype(np.float64) for v in (h1, h2, h3)]
    kept = sum(pa[i] * pb[j] for i in range(3) for j in range(3) if i + j <= 2)
    rel = np.abs(kept - a * b) / np.abs(a * b)
    assert rel.max() < 2.0 ** -22 and rel.mean() < 2.0 ** -25

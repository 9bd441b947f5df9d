"""CPU: the oracle's host-side stages against (a) golden vectors produced by the REFERENCE's
own functions (tests/golden/frontend_*.npz, made by tools/gen_golden.py through
oracle/_ref/libwt_ref_frontend.so) and (b), when oracle/_ref is present, the reference
functions themselves, live."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLD, synth_pcm


@pytest.fixture(scope="module")
def gold_host():
    return np.load(os.path.join(GOLD, "frontend_host.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def gold_mel():
    return np.load(os.path.join(GOLD, "frontend_logmel.npz"), allow_pickle=False)


def _crc(mel):
    w = np.arange(mel.size, dtype=np.uint64).reshape(mel.shape) * 2654435761 % (1 << 32) + 1
    return np.uint64(np.bitwise_xor.reduce((mel.view(np.uint32).astype(np.uint64) * w).reshape(-1)))


@pytest.mark.parametrize("kind,n", [("noise", 480000), ("sweep", 480000), ("speechlike", 480000), ("noise", 32000)])
def test_logmel_bit_exact_vs_reference_golden(orc, gold_host, gold_mel, kind, n):
    key = f"{kind}_{n}"
    pcm = synth_pcm(kind, n, int(gold_mel[key + "_seed"]))
    mel = orc.frontend().logmel(pcm, gold_host["filters"], n_threads=3)
    assert mel.shape == (80, n // 160)
    if n <= 32000:
        assert np.array_equal(mel.view(np.uint32), gold_mel[key + "_full"].view(np.uint32))
    else:
        assert np.array_equal(mel[:, ::97].view(np.uint32), gold_mel[key + "_cols"].view(np.uint32))
        assert np.array_equal(mel[::13, :].view(np.uint32), gold_mel[key + "_rows"].view(np.uint32))
    assert _crc(mel) == gold_mel[key + "_crc"]  # whole-tensor bit pattern
    assert np.float32(mel.max()) == gold_mel[key + "_max"]


def test_logmel_thread_count_independent(orc, gold_host):
    pcm = synth_pcm("noise", 48000, 9)
    fe = orc.frontend()
    a = fe.logmel(pcm, gold_host["filters"], 1)
    b = fe.logmel(pcm, gold_host["filters"], 7)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_logmel_edge_cases(orc, gold_host):
    fe = orc.frontend()
    # silence: every bin clamps to 1e-10 -> log10 = -10 -> (-10 + 4) / 4
    z = fe.logmel(np.zeros(16000, np.float32), gold_host["filters"], 2)
    assert np.all(z == np.float32(-1.5))
    # ragged tail: n_samples not a multiple of the hop -> floor(n / 160) frames
    r = fe.logmel(synth_pcm("noise", 16000 + 77, 5), gold_host["filters"], 2)
    assert r.shape == (80, (16000 + 77) // 160)


def test_logmel_live_vs_reference(orc, gold_host):
    ref = orc.ref_frontend()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    fe = orc.frontend()
    for kind, n, seed in (("noise", 64000, 11), ("sweep", 480000, 12)):
        pcm = synth_pcm(kind, n, seed)
        assert np.array_equal(fe.logmel(pcm, gold_host["filters"], 4).view(np.uint32),
                              ref.logmel(pcm, gold_host["filters"], 4).view(np.uint32))


def test_argmax_tie_rule(orc, gold_host):
    fe = orc.frontend()
    cases = ([1, 3, 3, 2], [5, 5, 5, 5], [0, -1, -2, -3], [-0.0, 0.0, -0.0, -1])
    got = [fe.argmax_last(np.array(c, np.float32)) for c in cases]
    assert got == list(gold_host["argmax_cases"]) == [2, 3, 0, 2]


def test_language_table(orc, gold_host):
    fe = orc.frontend()
    assert fe.language_count() == int(gold_host["lang_count"]) == 100
    assert [fe.language_id(c) for c in ("en", "de", "yue", "xx")] == list(gold_host["lang_ids"]) == [0, 2, 99, 100]
    assert fe.lang_code(2) == "de"


def test_wav_read_legacy_quirks(orc, gold_host, tmp_path):
    p = tmp_path / "ramp.wav"
    p.write_bytes(gold_host["wav_bytes"].tobytes())
    s = orc.frontend().wav_read_legacy(str(p))
    ref = gold_host["wav_samples"]
    assert s.shape == ref.shape and np.array_equal(s.view(np.uint32), ref.view(np.uint32))
    # the quirk itself: count from the RIFF size field (file - 8) / 2, and the 8-byte 'data'
    # chunk header decoded as the first four samples
    assert len(s) == (44 + 2000 - 8) // 2
    d = np.frombuffer(b"data" + struct.pack("<I", 2000), "<i2").astype(np.float32) / np.float32(32767)
    assert np.array_equal(s[:4], d)
    assert np.all(s[-4:] == 0)  # tail past EOF stays zero
    assert len(orc.frontend().wav_read_legacy(str(tmp_path / "missing.wav"))) == 0
    (tmp_path / "bad.wav").write_bytes(b"RIFX" + bytes(60))
    assert len(orc.frontend().wav_read_legacy(str(tmp_path / "bad.wav"))) == 0


def test_vocab_reader_and_decode(orc, pkg, gold_host, tmp_path):
    path = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(path, 300)
    v = orc.frontend().open_vocab(path, True)
    info = v.info()
    assert [info[k] for k in ("n_vocab", "eot", "sot", "translate", "transcribe", "prev", "solm", "not", "beg")] == list(gold_host["info"])
    assert info["eot"] == 50257 and info["sot"] == 50258 and info["not"] == 50363 and info["transcribe"] == 50359
    assert np.array_equal(v.filters(), gold_host["filters"])
    for i, b in zip(gold_host["tok_ids"], gold_host["tok_bytes"]):
        assert v.token(int(i)) == bytes(b), i
    assert v.size() == 51865
    assert v.decode(gold_host["decode_ids"], False) == gold_host["decode_text"].tobytes()
    assert v.decode(gold_host["decode_ids"], True) == gold_host["decode_text_omit"].tobytes()
    assert v.decode(gold_host["decode_ids"], False).endswith(b"<|endoftranscript|>")  # stops AFTER eot
    v.close()
    en = orc.frontend().open_vocab(path, False)
    assert en.info()["eot"] == 50256 and en.info()["n_vocab"] == 300 and en.size() == 51864
    en.close()
    assert orc.frontend().remove_extra_spaces("a  b   c d") == "a b c d"

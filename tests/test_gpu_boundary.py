"""GPU (-m gpu): the drop-in boundary and the parity corners the path tests do not reach.

  * rows a3 / a7 on the PRODUCT side through an engine handle (golden vectors produced by the reference's own code);
  * BASELINE configs[0]: tiny.en weights + multilingual = false (English vocabulary ids), through the C ABI, the
    `encdec` CLI and the Monolith-compatible `minimal` entry;
  * the C ABI never aborts: unsupported weight files / options come back as status codes;
  * synchronous calls refuse to run over uncollected pipeline batches;
  * the fp16 two-plane encoder kernels on adversarial weight statistics, and the load-time fall-back;
  * the free-function front end (whisper::log_mel_spectrogram);
  * the RCCL branch of bench.py, executed once on hardware.
"""
import ctypes
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, DevBuf, synth_pcm

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
ENC_TOL = 1e-4
MEL_TOL = 1e-4


def _wav(path, pcm):
    pcm16 = np.clip(np.round(pcm * 32767), -32768, 32767).astype("<i2")
    path.write_bytes(b"RIFF" + struct.pack("<I", 36 + pcm16.nbytes) + b"WAVEfmt " +
                     struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" +
                     struct.pack("<I", pcm16.nbytes) + pcm16.tobytes())


# ----------------------------------------------------------------------------- a3 / a7 ---

def test_engine_vocab_tables_and_text_match_reference_golden(pkg, assets, tmp_path):
    """wt_vocab_info / wt_filters / wt_decode_text of an ENGINE handle (the tables the hot path uses) against
    tests/golden/frontend_host.npz: ids, filter bank bit-exact, every sampled token, decode with and without
    special tokens (reference whisper.cpp:519-665)."""
    g = np.load(os.path.join(GOLD, "frontend_host.npz"))
    prefix, _ = assets("micro")
    vocab = str(tmp_path / "v300.bin")
    pkg.write_synthetic_vocab(vocab, 300)  # the file the goldens were made from
    e = pkg.Engine(prefix, vocab, True)
    info = e.vocab_info()
    assert [info[k] for k in ("n_vocab", "eot", "sot", "translate", "transcribe", "prev", "solm", "not", "beg")] == list(g["info"])
    assert np.array_equal(e.filters().view(np.uint32), g["filters"].view(np.uint32))
    for i, b in zip(g["tok_ids"], g["tok_bytes"]):
        assert e.decode_bytes([int(i)]) == bytes(b), i
    assert e.decode_bytes(g["decode_ids"], False) == g["decode_text"].tobytes()
    assert e.decode_bytes(g["decode_ids"], True) == g["decode_text_omit"].tobytes()
    with pytest.raises(pkg.WtError):
        e.decode_bytes([70000])
    e.close()


# -------------------------------------------------------------------------- configs[0] ---

@pytest.fixture(scope="module")
def tiny_en(pkg, assets):
    prefix, vocab = assets("tiny.en")
    e = pkg.Engine(prefix, vocab, False)  # multilingual = false: English vocabulary ids (whisper.h:69-91)
    yield e, prefix, vocab
    e.close()


def test_config1_tiny_en_english_vocab_path(tiny_en, orc):
    """BASELINE configs[0] on the HIP path: whisper-tiny.en dims (n_vocab 51864), multilingual = false ->
    eot 50256, sot 50257, notimestamps 50362 (no transform_vocab_multilingual, whisper.cpp:218-226, :560-562);
    EncDec prompt [50257, 50259 + 2, 50359, 50362] (whisper.cpp:327-339).  Ids exact, logits < 1e-4, vs the oracle."""
    e, prefix, _ = tiny_en
    assert e.dims.n_vocab == 51864
    info = e.vocab_info()
    assert (info["eot"], info["sot"], info["not"], info["transcribe"], info["n_vocab"]) == (50256, 50257, 50362, 50359, 50257)
    prompt = [50257, 50261, 50359, 50362]
    e.set_option("stop_at_eot", 0)
    rng = np.random.default_rng(77)
    mel = rng.uniform(-1.0, 1.5, size=(2, 80, 3000)).astype(np.float32)
    ids, n, enc, logits = e.encdec_debug_batch(mel)
    assert list(ids[0, :4]) == prompt and list(n) == [31, 31]
    m = orc.Model(prefix + ".wtw")
    for b in range(2):
        enc_ref = m.encode(mel[b], 16)
        assert np.abs(enc[b] - enc_ref).max() < ENC_TOL
        ids_ref, lg = m.decode_greedy(enc_ref, prompt, 30, 50256, False, True, 16, True)
        assert np.abs(logits[b] - lg).max() < LOGIT_TOL
        assert list(ids[b, :31]) == list(ids_ref)
    # the EOT stop uses the ENGLISH eot id (50256): same ids and counts as the oracle with the stop enabled
    e.set_option("stop_at_eot", 1)
    ids_s, n_s = e.encdec_tokens_batch(mel)
    ids_o, n_o = m.encdec_batch(mel, prompt, 30, 50256, True, True, n_threads=16)
    assert list(n_s) == list(n_o)
    for b in range(2):
        assert list(ids_s[b, :n_s[b]]) == list(ids_o[b, :n_o[b]])
    m.close()


def test_config1_single_wav_through_encdec_cli_and_minimal(tiny_en, orc, tmp_path):
    """configs[0] as the reference runs it: ONE 30 s WAV through the `encdec` binary (same three flags; `--english`
    selects multilingual = false, which the reference hard-codes to true, app/encdec.cpp:47) and through `minimal`
    (reference app/minimal.cpp: Monolith, multilingual = false -> prompt [sot 50257, notimestamps 50362], the head
    of kGoldenGeneratedIDs, whisper.h:27-32).  The transcript is the last stdout line / sits between blank lines."""
    e, prefix, vocab = tiny_en
    exe = os.path.join(ROOT, "whisper.tflite_amd", "bin", "encdec")
    mini = os.path.join(ROOT, "whisper.tflite_amd", "bin", "minimal")
    if not (os.path.exists(exe) and os.path.exists(mini)):
        pytest.skip("CLI binaries not built")
    pcm = synth_pcm("sweep", 480000, 5)
    wav = tmp_path / "clip30.wav"
    _wav(wav, pcm)
    e.set_option("stop_at_eot", 1)
    want = e.transcribe(str(wav))
    # English ids: 50261 = sot + 1 + 3, so the reference's table names the "de" prompt token <|lang-es|> here
    # (whisper.cpp:596-598: lang_code(i - (token_sot + 1)))
    assert want.startswith("<|startoftranscript_|><|lang-es|><|transcribe|><|notimestamps|>")
    r = subprocess.run([exe, "--model-prefix", prefix, "--vocab", vocab, "--input", str(wav), "--english"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[-1] == want
    # Monolith-compatible entry: same kernels, the prompt HF generate() forces for an English-only model
    mono = e.__class__(prefix, vocab, False, engine_type=0)
    mono.set_option("stop_at_eot", 1)
    samples = orc.frontend().wav_read_legacy(str(wav))
    padded = np.zeros(480000, np.float32)
    padded[: min(len(samples), 480000)] = samples[:480000]
    mel = mono.logmel_batch(padded[None])
    ids, n = mono.encdec_tokens_batch(mel)
    assert list(ids[0, :2]) == [50257, 50362]
    m = orc.Model(prefix + ".wtw")
    ids_ref, _ = m.decode_greedy(m.encode(mel[0], 16), [50257, 50362], 30, 50256, True, True, 16, False)
    assert list(ids[0, :n[0]]) == list(ids_ref)
    m.close()
    text = mono.decode_text(ids[0, :n[0]])
    mono.close()
    r = subprocess.run([mini, prefix, vocab, str(wav)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    import re
    assert r.stdout == "\n" + re.sub(" +", " ", text) + "\n\n"
    e.set_option("stop_at_eot", 0)


def test_monolith_multilingual_prompt(pkg, assets):
    """EngineType::Monolith with a multilingual vocabulary: [sot, <|en|>, transcribe, notimestamps] — the forced
    decoder ids of the HF generate() graph the reference's Monolith runs (export/generate.py:24-30)."""
    prefix, vocab = assets("tiny")
    e = pkg.create_engine(0, prefix, vocab, True)
    assert e is not None
    e.set_option("stop_at_eot", 0)
    e.set_option("max_tokens", 6)
    mel = np.random.default_rng(3).uniform(-1.0, 1.5, size=(1, 80, 3000)).astype(np.float32)
    ids, n = e.encdec_tokens_batch(mel)
    assert list(ids[0, :4]) == [50258, 50259, 50359, 50363] and n[0] == 7
    e.close()


# ------------------------------------------------------------------- never abort behind the ABI ---

def test_unsupported_weight_files_and_options_return_status_codes(pkg, assets, tmp_path):
    """d_model 256 passes the reference-style file checks but has no kernel instantiation; zero heads would
    divide by zero; a context below the kernels' minimum; a wrapped tensor range; an unaligned payload; an
    unknown GEMM variant.  Each must come back as a status code with a message — the process survives
    (before: abort() inside launch_ln / launch_gemm_t, SIGFPE in upload_weights)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from wtw import read_wtw, write_wtw
    prefix, vocab = assets("micro")
    dims, t = read_wtw(prefix + ".wtw")
    L = pkg.lib()

    def create(path_prefix):
        h = ctypes.c_void_p()
        rc = L.wt_engine_create(1, path_prefix.encode(), vocab.encode(), 1, 0, ctypes.byref(h))
        assert not h.value
        return rc, L.wt_last_error(None).decode()

    def variant(name, **over):
        d = dict(dims)
        d.update(over)
        p = str(tmp_path / name)
        write_wtw(p + ".wtw", d, t)
        return p

    rc, msg = create(variant("d256", n_audio_state=256, n_text_state=256, n_audio_head=4, n_text_head=4))
    assert rc == 3 and "d_model" in msg                      # WT_ERR_FORMAT
    rc, msg = create(variant("heads0", n_audio_head=0))
    assert rc == 3 and "heads" in msg
    rc, msg = create(variant("ctx8", n_audio_ctx=8))
    assert rc == 3
    rc, msg = create(variant("tctx16", n_text_ctx=16))
    assert rc == 3
    rc, msg = create(variant("vocab0", n_vocab=0))
    assert rc == 3
    # tensor table: offset + nbytes wraps around 2^64 / payload not 4-byte aligned
    raw = bytearray(open(prefix + ".wtw", "rb").read())
    e0 = 128
    bad = bytearray(raw)
    struct.pack_into("<QQ", bad, e0 + 104, 2 ** 64 - 4, 8)
    (tmp_path / "wrap.wtw").write_bytes(bad)
    assert create(str(tmp_path / "wrap"))[0] == 3
    bad = bytearray(raw)
    off, nb = struct.unpack_from("<QQ", raw, e0 + 104)
    struct.pack_into("<QQ", bad, e0 + 104, off + 2, nb - 4)
    (tmp_path / "unaligned.wtw").write_bytes(bad)
    assert create(str(tmp_path / "unaligned"))[0] == 3
    (tmp_path / "trunc.wtw").write_bytes(raw[:1000])
    assert create(str(tmp_path / "trunc"))[0] == 3
    assert create(str(tmp_path / "missing"))[0] == 2          # WT_ERR_IO
    # options: variants that do not exist are refused when they are set, not when a kernel is launched
    e = pkg.Engine(prefix, vocab, True)
    for bad_v in (12, 1, 10, 11, 14, 17, 18, 19, -2):
        with pytest.raises(pkg.WtError) as ei:
            e.set_option("gemm_variant", bad_v)
        assert ei.value.code == 1
    for bad_v in (2, 3, 5, -1):
        with pytest.raises(pkg.WtError):
            e.set_option("attn_variant", bad_v)
    e.set_prompt([3, 5, 7, 11])
    mel = np.random.default_rng(1).uniform(-1, 1.5, size=(1,) + e.mel_shape).astype(np.float32)
    ids, n = e.encdec_tokens_batch(mel)  # and the engine works after the refused options
    assert n[0] >= 5
    # kernel-level taps: shapes outside a kernel's contract are errors too
    with pytest.raises(pkg.WtError):
        e.dbg_layernorm(np.zeros((4, 600), np.float32), np.ones(600, np.float32), np.zeros(600, np.float32))
    with pytest.raises(pkg.WtError):
        e.dbg_dec_gemm(np.zeros((4, 96), np.float32), np.zeros((64, 96), np.float32), mode=2, R=np.zeros((4, 64), np.float32))
    e.close()


def test_sync_call_with_batches_in_flight_is_refused_before_any_work(pkg, assets):
    """With the pipeline FULL (WT_PIPELINE_DEPTH uncollected batches) a synchronous call used to enqueue its encoder first — onto
    the oldest uncollected slot, overwriting that batch — and throw only at decode().  Now every synchronous entry
    point checks first: error, nothing enqueued, and the later collects return the ORIGINAL ids (also when the
    refused call carried a larger batch than the submitted ones)."""
    prefix, vocab = assets("micro")
    e = pkg.Engine(prefix, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_prompt([3, 5, 7, 11])
    rng = np.random.default_rng(21)
    D = pkg.WT_PIPELINE_DEPTH
    mels = [rng.uniform(-1.0, 1.5, size=(3,) + e.mel_shape).astype(np.float32) for _ in range(D)]
    big = rng.uniform(-1.0, 1.5, size=(9,) + e.mel_shape).astype(np.float32)
    want = [e.encdec_tokens_batch(m) for m in mels]
    e.encdec_tokens_batch(big)  # grows the workspace now, not while batches are in flight
    dev = [DevBuf(m) for m in mels]
    d_big = DevBuf(big)
    pcm = DevBuf(np.zeros((9, e.pcm_len), np.float32))
    for d in dev:
        e.pipeline_submit_dev(d.data_ptr(), 3)
    assert e.get_option("in_flight") == D
    for call in (lambda: e.encdec_tokens_batch(big), lambda: e.encdec_tokens_batch_dev(d_big.data_ptr(), 9),
                 lambda: e.encdec_debug_batch(big), lambda: e.transcribe(np.zeros(1000, np.float32)),
                 lambda: e.logmel_batch(np.zeros((1, e.pcm_len), np.float32)),
                 lambda: e.transcribe_tokens_batch_dev(pcm.data_ptr(), 9),
                 lambda: e.transcribe_long(np.zeros(1000, np.float32))):
        with pytest.raises(pkg.WtError) as ei:
            call()
        assert ei.value.code == 1 and "collect" in str(ei.value)
    with pytest.raises(pkg.WtError):
        e.pipeline_submit_dev(dev[0].data_ptr(), 3)  # one submit too many: pipeline full
    assert e.get_option("in_flight") == D
    for k in range(D):
        ids, n = e.pipeline_collect()
        assert ids.shape == (3, 32) and np.array_equal(ids, want[k][0]) and np.array_equal(n, want[k][1]), k
    ids, n = e.encdec_tokens_batch(mels[2])  # synchronous calls work again
    assert np.array_equal(ids, want[2][0])
    e.close()


# ----------------------------------------------------- fp16 two-plane kernels, adversarial weights ---

def _adversarial_tiny(assets, tmp_path, name, ln_gain=30.0, heavy=True, v_row_scale=1.0):
    """whisper-tiny random-init weights with the statistics a trained checkpoint can have and N(0, 1/fan_in) does
    not (tools/wtw.py adversarial_weights: LayerNorm-gain outliers, LayerNorm shifts, heavy-tailed rows, one value
    channel `v_row_scale` x larger with its out-projection column that much smaller)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from wtw import adversarial_weights
    prefix, vocab = assets("tiny")
    p = str(tmp_path / name)
    adversarial_weights(prefix + ".wtw", p + ".wtw", ln_gain=ln_gain, heavy=heavy, v_row_scale=v_row_scale)
    return p, vocab


def _encoder_errors(pkg, orc, prefix, vocab, mel):
    """max |encoder output - fp64 reference| of the default path, of the two fp32-storage forms and of the CPU oracle
    (tests/fp64_encoder.py: the same graph in numpy float64 — the arbiter where fp32 implementations disagree), the
    output scale, and the engine's fall-back count"""
    from fp64_encoder import encoder_fp64
    from wtw import read_wtw
    e = pkg.Engine(prefix, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_option("max_tokens", 5)
    fallbacks = e.get_option("f16_fallbacks")
    enc = {}
    for name, gv, av in (("default", -1, 4), ("bf16x3", 16, 1), ("fp32_mfma", 0, 0)):
        e.set_option("gemm_variant", gv)
        e.set_option("attn_variant", av)
        enc[name] = e.encdec_debug_batch(mel, want_logits=False)[2][0]
    e.close()
    m = orc.Model(prefix + ".wtw")
    enc["oracle"] = m.encode(mel[0], 16)
    m.close()
    dims, t = read_wtw(prefix + ".wtw")
    ref = encoder_fp64(dims, t, mel[0])
    assert all(np.isfinite(v).all() for v in enc.values())
    return {k: float(np.abs(v - ref).max()) for k, v in enc.items()}, float(np.abs(ref).max()), fallbacks


def test_fp16_split_on_outlier_weights_is_as_accurate_as_fp32_mfma(pkg, assets, orc, tmp_path):
    """LayerNorm gains x30 on a few channels and heavy-tailed rows inflate the weight-derived bounds the default
    two-plane fp16 kernels take their scales from, and make the network itself ill-conditioned: every fp32
    implementation then sits 1e-4 .. 1e-3 of the output scale away from the exact (fp64) result, the CPU oracle
    included, and two of them differ from each other by as much (summation order).  The arbiter is therefore the graph
    evaluated in float64 (tests/fp64_encoder.py; it agrees with the oracle to 1e-6 on N(0, 1/fan_in) weights,
    tests/test_oracle_model.py): each GPU form must be as close to it as an fp32 implementation can be — within 3x of
    the better of the oracle's and the fp32-MFMA kernels' own distance."""
    mel = np.random.default_rng(9).uniform(-1.0, 1.5, size=(1, 80, 3000)).astype(np.float32)
    for name, kw in (("outliers", dict(ln_gain=30.0, heavy=True)), ("gains-only", dict(ln_gain=30.0, heavy=False)),
                     ("tails-only", dict(ln_gain=1.0, heavy=True))):
        prefix, vocab = _adversarial_tiny(assets, tmp_path, "tiny-" + name, **kw)
        err, scale, fallbacks = _encoder_errors(pkg, orc, prefix, vocab, mel)
        print(name, {k: f"{v / scale:.2e}" for k, v in err.items()}, "scale", scale)
        assert fallbacks == 0, name  # still inside the slack the fp16 form is used for
        fp32_level = min(err["oracle"], err["fp32_mfma"])
        assert err["oracle"] < 2e-3 * scale and err["fp32_mfma"] < 2e-3 * scale, (name, err, scale)  # fp32-sized at all
        for form in ("default", "bf16x3", "fp32_mfma"):
            assert err[form] < 3.0 * fp32_level + 1e-6 * scale, (name, form, err, scale)


def test_fp16_split_falls_back_per_contraction_when_a_bound_is_far_above_typical(pkg, assets, orc, tmp_path):
    """One output channel of a value projection 10^4 x larger than the others (and its out-projection column that much
    smaller) puts the weight-derived bound of V — the scale of the fp16 planes of the whole tensor — more than 2^12
    above V's typical magnitude: typical elements would lose their second fp16 plane to the subnormal range.  The
    engine must give THOSE contractions (layer 0's attention and its out-projection) the bf16 three-plane kernels by
    itself at load time (f16_fallbacks >= 2) — and only those: the other 18 GEMMs (of 19: conv1, conv2, four per
    layer, and the cross-KV projection of a synchronous one-clip call — pipelined batches run the absorbed
    cross-attention and have 18) and 3 attentions of the pass stay on the plane kernels (wt_last_kernel_stats), the
    hand-over being the fp32 output form of the qkv plane GEMM.
    The result stays at the fp32 instruction's error level."""
    mel = np.random.default_rng(10).uniform(-1.0, 1.5, size=(1, 80, 3000)).astype(np.float32)
    prefix, vocab = _adversarial_tiny(assets, tmp_path, "tiny-vrow", ln_gain=1.0, heavy=False, v_row_scale=1.0e4)
    err, scale, fallbacks = _encoder_errors(pkg, orc, prefix, vocab, mel)  # against the fp64 graph
    assert fallbacks >= 2  # layer 0: the attention and its out-projection
    assert err["default"] < 3.0 * min(err["oracle"], err["fp32_mfma"]) + 1e-6 * scale, (err, scale)
    e = pkg.Engine(prefix, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_option("max_tokens", 5)
    e.set_option("kernel_timers", 1)
    e.encdec_tokens_batch(mel)
    ks = e.kernel_stats()
    assert ks["gemm_planes"]["launches"] == 19 - 1 and ks["gemm_split16_tile"]["launches"] == 1, ks
    assert ks["encoder_attention_planes"]["launches"] == 3 and ks["encoder_attention_split"]["launches"] == 1, ks
    assert ks["f32_to_planes"]["launches"] == 0  # fall-back feeds fall-back here: no conversion needed
    # a fall-back producer in front of a plane consumer: forcing only the attention off the plane kernel makes every
    # layer hand its fp32 result to the plane out-projection through f32_to_planes
    e.set_option("attn_variant", 1)
    ids_a, _, enc_a, _ = e.encdec_debug_batch(mel, want_logits=False)
    ks = e.kernel_stats()
    assert ks["encoder_attention_split"]["launches"] == 4 and ks["encoder_attention_planes"]["launches"] == 0, ks
    assert ks["f32_to_planes"]["launches"] == 3 and ks["gemm_planes"]["launches"] == 18, ks
    e.set_option("attn_variant", 4)
    ids_b, _, enc_b, _ = e.encdec_debug_batch(mel, want_logits=False)
    assert np.abs(enc_a - enc_b).max() < 6.0 * min(err["oracle"], err["fp32_mfma"]) + 2e-6 * scale
    e.close()


def test_trained_like_weight_statistics_in_every_layer(pkg, assets, orc, tmp_path):
    """tools/wtw.py trained_like_weights: log-normal LayerNorm gains with 6..20 x outlier channels, heavy-tailed weight rows
    with log-normal norms and two "massive" residual channels (40 x) in EVERY encoder layer — the statistics the
    headline's random-init weights do not have.  The load-time slack check decides per contraction which ones leave the
    fp16-plane kernels (reported: f16_fallbacks of f16_contractions, the smallest slack in bits); whatever it decides, the
    encoder must sit at the fp32 error level against the graph in float64, like every fp32 implementation."""
    from wtw import trained_like_weights
    src, vocab = assets("tiny")
    prefix = str(tmp_path / "tiny-trained-like")
    trained_like_weights(src + ".wtw", prefix + ".wtw")
    mel = np.random.default_rng(11).uniform(-1.0, 1.5, size=(1, 80, 3000)).astype(np.float32)
    err, scale, fallbacks = _encoder_errors(pkg, orc, prefix, vocab, mel)
    e = pkg.Engine(prefix, vocab, True)
    n_c, slack = e.get_option("f16_contractions"), e.get_option("f16_min_slack_millibits") / 1000.0
    e.close()
    print("trained-like:", fallbacks, "of", n_c, "contractions leave the plane kernels; min slack", slack, "bits;",
          {k: f"{v / scale:.2e}" for k, v in err.items()})
    assert n_c == 23 and 0 <= fallbacks <= n_c
    assert (slack < 0) == (fallbacks > 0)
    fp32_level = min(err["oracle"], err["fp32_mfma"])
    assert fp32_level < 2e-3 * scale
    # (the massive channels make the network ill-conditioned: fp32 implementations differ from each other by ~1e-4 of the
    # output scale here; the six-product fall-back form, not on the default path for these weights, measured 3.2 x the oracle's
    # distance)
    for form, factor in (("default", 3.0), ("fp32_mfma", 3.0), ("bf16x3", 5.0)):
        assert err[form] < factor * fp32_level + 1e-6 * scale, (form, err, scale)


def test_fall_back_in_every_position_keeps_ids(pkg, assets, orc):
    """Every producer -> consumer hand-over of the mixed encoder (plane -> fall-back: fp32 output of the plane GEMM /
    LayerNorm; fall-back -> plane: f32_to_planes), exercised by flagging contractions one at a time through the test
    hook WT_FORCE_FALLBACK (bit i = contraction i in launch order: conv1, conv2, then per layer qkv, attention, out,
    fc1, fc2, and last the operand of the decoder's cross-attention — flagged, it sends the decoder back to the
    cross-KV cache filled by the full-range GEMM): ids and encoder output must equal the all-plane path's."""
    prefix, vocab = assets("tiny")
    rng = np.random.default_rng(12)
    mel = rng.uniform(-1.0, 1.5, size=(2, 80, 3000)).astype(np.float32)
    e = pkg.Engine(prefix, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_option("max_tokens", 8)
    ids0, n0, enc0, _ = e.encdec_debug_batch(mel, want_logits=False)
    # contraction indices of layer 0: qkv 2, attention 3, out 4, fc1 5, fc2 6; layer l adds 5 l; cross-KV is 22
    for mask in (1 << 0, 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5, 1 << 6, 1 << 22, (1 << 2) | (1 << 4), (1 << 7) | (1 << 11),
                 (1 << 23) - 1):
        e.set_option("force_fallback", mask)
        assert e.get_option("f16_fallbacks") == bin(mask).count("1")
        ids, n, enc, _ = e.encdec_debug_batch(mel, want_logits=False)
        assert np.array_equal(ids, ids0) and np.array_equal(n, n0), hex(mask)
        assert np.abs(enc - enc0).max() < 2e-5, hex(mask)
    e.set_option("force_fallback", 0)
    assert e.get_option("f16_fallbacks") == 0
    e.close()


# --------------------------------------------------------------- .tflite model files (f1) ---

def test_engine_accepts_the_reference_tflite_pair(pkg, orc, tmp_path):
    """`--model-prefix` as the reference's users have it: only <prefix>.encoder.tflite and <prefix>.decoder.tflite
    exist (whisper.cpp:743-744).  wt_engine_create extracts the weights (int8 dynamic-range de-quantised) into
    <prefix>.wtw and runs; ids and logits equal the oracle's on exactly those de-quantised weights.
    PARITY UNPINNED against TFLite itself: the pair comes from tests/tflite_writer.py (no .tflite exists here)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_tflite_extract import make_pair
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named=False)
    vocab = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(vocab, 1000)
    assert not os.path.exists(prefix + ".wtw")
    e = pkg.Engine(prefix, vocab, True)
    assert os.path.exists(prefix + ".wtw")
    e.set_option("stop_at_eot", 0)
    e.set_prompt([3, 5, 7, 11])
    mel = np.random.default_rng(2).uniform(-1.0, 1.5, size=(2,) + e.mel_shape).astype(np.float32)
    ids, n, enc, logits = e.encdec_debug_batch(mel)
    e.close()
    m = orc.Model(prefix + ".wtw")
    for b in range(2):
        enc_ref = m.encode(mel[b])
        assert np.abs(enc[b] - enc_ref).max() < ENC_TOL
        ids_ref, lg = m.decode_greedy(enc_ref, [3, 5, 7, 11], 30, -1, False, True, 4, True)
        assert np.abs(logits[b] - lg).max() < LOGIT_TOL
        assert list(ids[b, :31]) == list(ids_ref)
    m.close()


# ------------------------------------------------------------------ front end as a free function ---

def test_log_mel_spectrogram_free_function(pkg, orc):
    """whisper::log_mel_spectrogram (whisper.h:123) without an engine: wt_log_mel_spectrogram runs the same kernels
    through a front-end-only context.  Full 30 s clip vs the reference-produced golden; a 2 s clip (n_len = 200
    frames: the maximum runs over those frames only) vs the golden and the oracle; unsupported geometry refused."""
    g = np.load(os.path.join(GOLD, "frontend_logmel.npz"))
    host = np.load(os.path.join(GOLD, "frontend_host.npz"))
    filters = host["filters"]
    pcm = synth_pcm("noise", 480000, int(g["noise_480000_seed"]))
    mel = pkg.log_mel_spectrogram(pcm, filters)
    assert mel.shape == (80, 3000)
    assert np.abs(mel[:, ::97] - g["noise_480000_cols"]).max() < MEL_TOL
    assert np.abs(mel[::13, :] - g["noise_480000_rows"]).max() < MEL_TOL
    short = synth_pcm("noise", 32000, int(g["noise_32000_seed"]))
    mel_s = pkg.log_mel_spectrogram(short, filters)
    assert mel_s.shape == (80, 200)
    assert np.abs(mel_s - g["noise_32000_full"]).max() < MEL_TOL
    odd = short[:31999]  # n_samples not a multiple of the hop: n_len = 199
    assert np.abs(pkg.log_mel_spectrogram(odd, filters) - orc.frontend().logmel(odd, filters, 4)).max() < MEL_TOL
    with pytest.raises(pkg.WtError) as ei:
        pkg.log_mel_spectrogram(np.zeros(480001, np.float32), filters)
    assert ei.value.code == 4
    assert pkg.log_mel_spectrogram(np.zeros(0, np.float32), filters).shape == (80, 0)


# -------------------------------------------------------------------------- RCCL on hardware ---

def test_bench_rccl_branch_executes_on_one_rank(tmp_path):
    """The N > 1 path of bench.py (init_process_group("nccl") = RCCL, device-tensor all_gather of the id records,
    barrier, all_reduce of the elapsed time) executed on ONE rank in a fresh child process, so that the driver's
    8-GPU run is not the first time it runs.  Ids must equal those of the same run without collectives."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "9", "--warmup", "1",
            "--no-cpu-baseline", "--no-fp32-leg", "--emit-ids"]
    r1 = subprocess.run(base + ["--rehearse-nccl"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r1.returncode == 0, r1.stderr[-3000:]
    a = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run(base, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r2.returncode == 0, r2.stderr[-3000:]
    b = json.loads(r2.stdout.strip().splitlines()[-1])
    assert a["collectives"] >= 1 and b["collectives"] == 0
    assert a["n_gpus"] == 1 and a["steps"] == 9 and a["value"] > 0 and a["unit"] == "audio-sec/s"
    assert a["gathered_records"] == 32 * 9 and a["ids_crc"] == b["ids_crc"]
    assert "roofline" in a and "config" in a


def test_bench_two_ranks_share_the_gpu_and_the_gather_cadence(tmp_path):
    """Rehearsal of the driver's multi-GPU launch on the one GPU there is: two ranks as fresh child processes of
    torch.distributed.run (gloo collectives, both on cuda:0 — `--single-device --backend gloo`), i.e. concurrent engine
    creation, per-rank temporary assets, shard_range(rank, 2, 64), the all_gather cadence (8 batches per collective) and
    the max-over-ranks timing.  The gathered records must be exactly the records of the same 64 global clips decoded by
    ONE process (order-independent digest), and a single-rank run must map ONE libamdhip64 (the engine allocates every
    device buffer itself; torch is not imported at N = 1)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "9", "--warmup", "1", "--no-cpu-baseline", "--no-fp32-leg", "--emit-ids"]
    two = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device"] + common
    r2 = subprocess.run(two, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r2.returncode == 0, r2.stderr[-3000:]
    lines = [l for l in r2.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r2.stdout[-2000:]  # rank 0 prints the one line
    a = json.loads(lines[-1])
    one = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "64", "--report-maps"] + common
    r1 = subprocess.run(one, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r1.returncode == 0, r1.stderr[-3000:]
    b = json.loads(r1.stdout.strip().splitlines()[-1])
    assert a["n_gpus"] == 2 and a["config"]["global_batch"] == 64 and a["scaling"] == "weak"
    assert a["collectives"] == 2 and b["collectives"] == 0          # 9 batches: one gather of 8, one of 1
    assert a["gathered_records"] == 64 * 9 == b["gathered_records"]
    assert a["ids_digest"] == b["ids_digest"] and a["ids_digest"] != 0
    assert a["value"] > 0 and b["value"] > 0
    assert len(b["hip_runtimes_mapped"]) == 1, b["hip_runtimes_mapped"]


def test_device_buffers_of_the_c_abi(pkg, assets):
    """wt_device_alloc / upload / download / free / synchronize (include/wt_capi.h, round 4): a host program without a HIP
    runtime of its own keeps its inputs resident through the engine.  Round trip at an offset, the *_dev entry points take
    the pointer (same ids as the host-buffer call), bad arguments are refused with a status, not a fault."""
    from ctypes import byref, c_void_p
    prefix, vocab = assets("micro")
    e = pkg.Engine(prefix, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_prompt([3, 5, 7, 11])
    rng = np.random.default_rng(31)
    mel = rng.uniform(-1.0, 1.5, size=(3,) + e.mel_shape).astype(np.float32)
    d = e.device_array(mel)
    assert np.array_equal(d.download(), mel)
    L = pkg.lib()
    patch = rng.standard_normal(64).astype(np.float32)
    assert L.wt_device_upload(e._h, c_void_p(d.data_ptr()), 1024, patch.ctypes.data_as(c_void_p), patch.nbytes) == 0
    back = np.empty(64, np.float32)
    assert L.wt_device_download(e._h, back.ctypes.data_as(c_void_p), c_void_p(d.data_ptr()), 1024, back.nbytes) == 0
    assert np.array_equal(back, patch)
    assert L.wt_device_upload(e._h, c_void_p(d.data_ptr()), 0, mel.ctypes.data_as(c_void_p), mel.nbytes) == 0
    e.device_synchronize()
    want = e.encdec_tokens_batch(mel)
    got = e.encdec_tokens_batch_dev(d.data_ptr(), 3)
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
    p = c_void_p()
    assert L.wt_device_alloc(e._h, 0, byref(p)) == 0 and p.value  # an empty request still returns a buffer the caller frees
    assert L.wt_device_free(e._h, p) == 0
    assert L.wt_device_alloc(None, 16, byref(p)) != 0         # no engine
    assert L.wt_device_upload(e._h, None, 0, mel.ctypes.data_as(c_void_p), 16) != 0
    assert L.wt_device_download(e._h, None, c_void_p(d.data_ptr()), 0, 16) != 0
    d.free()
    d.free()  # idempotent on the Python side
    e.close()

"""Generates tests/golden/*.npz — run in the build container only (needs /root/reference for
the front-end vectors and the locally installed `transformers` for the model vectors).

  frontend_*.npz : outputs of the REFERENCE's own functions (oracle/_ref/libwt_ref_frontend.so,
                   compiled from /root/reference by oracle/build_ref.sh) on seeded inputs.
  model_*.npz    : outputs of HuggingFace transformers' WhisperForConditionalGeneration (an
                   independent implementation of the architecture the reference's exporter
                   traces, export/generate_onnx.py:85-120) loaded with this repo's deterministic
                   synthetic weights.  The reference's own model arithmetic (TFLite runtime +
                   .tflite graph) is absent, so model parity is pinned to this second
                   implementation, not to the reference ("parity unpinned" vs TFLite itself).

Usage: python tools/gen_golden.py [--skip-tiny]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402
from wtw import read_wtw  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def synth_pcm(kind, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    if kind == "noise":
        return np.clip(rng.normal(0, 0.1, n), -1, 1).astype(np.float32)
    if kind == "sweep":  # sine sweep 100 Hz -> 6 kHz over a noise floor
        f = 100.0 + (6000.0 - 100.0) * t / t[-1]
        return (0.5 * np.sin(2 * np.pi * np.cumsum(f) / 16000.0) + rng.normal(0, 1e-3, n)).astype(np.float32)
    if kind == "speechlike":  # amplitude-modulated harmonics, silence at both ends
        x = np.zeros(n)
        for h in range(1, 12):
            x += np.sin(2 * np.pi * 140.0 * h * t + rng.uniform(0, 6.28)) / h
        env = (np.sin(2 * np.pi * 3.0 * t) > 0) * np.hanning(n)
        return (0.2 * x * env + rng.normal(0, 3e-4, n)).astype(np.float32)
    raise ValueError(kind)


def frontend_goldens():
    orc = ge.load_oracle()
    pkg = ge.load_package()
    ref = orc.ref_frontend()
    assert ref is not None, "oracle/_ref not built (need /root/reference)"
    with tempfile.TemporaryDirectory() as tmp:
        vocab_path = os.path.join(tmp, "v.bin")
        pkg.write_synthetic_vocab(vocab_path, 300)
        v = ref.open_vocab(vocab_path, True)
        filters = v.filters()
        info = v.info()
        toks = {i: v.token(i) for i in (0, 65, 299, 300, 50256, 50257, 50258, 50259, 50261, 50357, 50358, 50359,
                                        50360, 50361, 50362, 50363, 50364, 50365, 51864)}
        ids = np.array([50258, 50261, 50359, 50363, 65, 66, 299, 50257, 70], np.int64)
        text = v.decode(ids, False)
        text_omit = v.decode(ids, True)
        v.close()
        # WAV: canonical 44-byte header, 1000 samples of a ramp; legacy reader quirks included
        wav = os.path.join(tmp, "ramp.wav")
        ramp = (np.arange(1000) - 500) / 600.0
        import struct
        pcm16 = np.clip(np.round(ramp * 32767), -32768, 32767).astype("<i2")
        with open(wav, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + pcm16.nbytes) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" +
                    struct.pack("<I", pcm16.nbytes) + pcm16.tobytes())
        wav_bytes = np.frombuffer(open(wav, "rb").read(), np.uint8)
        wav_samples = ref.wav_read_legacy(wav)
    np.savez_compressed(
        os.path.join(GOLD, "frontend_host.npz"), filters=filters,
        info=np.array([info[k] for k in ("n_vocab", "eot", "sot", "translate", "transcribe", "prev", "solm", "not", "beg")], np.int32),
        tok_ids=np.array(sorted(toks), np.int32),
        tok_bytes=np.array([toks[i] for i in sorted(toks)], dtype=object).astype("S64"),
        decode_ids=ids, decode_text=np.frombuffer(text, np.uint8), decode_text_omit=np.frombuffer(text_omit, np.uint8),
        wav_bytes=wav_bytes, wav_samples=wav_samples,
        lang_ids=np.array([ref.language_id(c) for c in ("en", "de", "yue", "xx")], np.int32),
        lang_count=np.int32(ref.language_count()),
        argmax_cases=np.array([ref.argmax_last(np.array(c, np.float32)) for c in ([1, 3, 3, 2], [5, 5, 5, 5], [0, -1, -2, -3], [-0.0, 0.0, -0.0, -1])], np.int64))
    # log-mel: full 80x3000 is 960 KB per clip; keep slices + statistics for 30 s clips and the
    # complete tensor for a 2 s clip
    out = {}
    for kind, n, seed in (("noise", 480000, 1), ("sweep", 480000, 2), ("speechlike", 480000, 3), ("noise", 32000, 4)):
        pcm = synth_pcm(kind, n, seed)
        mel = ref.logmel(pcm, filters, 4)
        key = f"{kind}_{n}"
        out[key + "_seed"] = np.int64(seed)
        if n <= 32000:
            out[key + "_full"] = mel
        else:
            out[key + "_cols"] = np.ascontiguousarray(mel[:, ::97])
            out[key + "_rows"] = np.ascontiguousarray(mel[::13, :])
        out[key + "_sum"] = np.float64(mel.astype(np.float64).sum())
        out[key + "_max"] = np.float32(mel.max())
        w = np.arange(mel.size, dtype=np.uint64).reshape(mel.shape) * 2654435761 % (1 << 32) + 1
        out[key + "_crc"] = np.uint64(np.bitwise_xor.reduce((mel.view(np.uint32).astype(np.uint64) * w).reshape(-1)))
    np.savez_compressed(os.path.join(GOLD, "frontend_logmel.npz"), **out)
    print("frontend goldens written")


def hf_model(dims, tensors):
    import torch
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(
        vocab_size=dims["n_vocab"], num_mel_bins=dims["n_mels"], d_model=dims["n_audio_state"],
        encoder_layers=dims["n_audio_layer"], decoder_layers=dims["n_text_layer"],
        encoder_attention_heads=dims["n_audio_head"], decoder_attention_heads=dims["n_text_head"],
        encoder_ffn_dim=4 * dims["n_audio_state"], decoder_ffn_dim=4 * dims["n_text_state"],
        max_source_positions=dims["n_audio_ctx"], max_target_positions=dims["n_text_ctx"],
        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1,
        suppress_tokens=None, begin_suppress_tokens=None)
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = {}

    def T(n):
        return torch.from_numpy(np.array(tensors[n]))

    def attn(src, dst):
        for a, b in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
            sd[f"{dst}.{b}.weight"] = T(f"{src}.{a}.weight")
            if a != "key":
                sd[f"{dst}.{b}.bias"] = T(f"{src}.{a}.bias")

    def ln(src, dst):
        sd[dst + ".weight"] = T(src + ".weight")
        sd[dst + ".bias"] = T(src + ".bias")

    for n in ("conv1", "conv2"):
        sd[f"model.encoder.{n}.weight"] = T(f"encoder.{n}.weight")
        sd[f"model.encoder.{n}.bias"] = T(f"encoder.{n}.bias")
    sd["model.encoder.embed_positions.weight"] = T("encoder.positional_embedding")
    for i in range(dims["n_audio_layer"]):
        s, d = f"encoder.blocks.{i}", f"model.encoder.layers.{i}"
        ln(s + ".attn_ln", d + ".self_attn_layer_norm")
        attn(s + ".attn", d + ".self_attn")
        ln(s + ".mlp_ln", d + ".final_layer_norm")
        for a, b in (("mlp.0", "fc1"), ("mlp.2", "fc2")):
            sd[f"{d}.{b}.weight"] = T(f"{s}.{a}.weight")
            sd[f"{d}.{b}.bias"] = T(f"{s}.{a}.bias")
    ln("encoder.ln_post", "model.encoder.layer_norm")
    sd["model.decoder.embed_tokens.weight"] = T("decoder.token_embedding.weight")
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    sd["model.decoder.embed_positions.weight"] = T("decoder.positional_embedding")
    for i in range(dims["n_text_layer"]):
        s, d = f"decoder.blocks.{i}", f"model.decoder.layers.{i}"
        ln(s + ".attn_ln", d + ".self_attn_layer_norm")
        attn(s + ".attn", d + ".self_attn")
        ln(s + ".cross_attn_ln", d + ".encoder_attn_layer_norm")
        attn(s + ".cross_attn", d + ".encoder_attn")
        ln(s + ".mlp_ln", d + ".final_layer_norm")
        for a, b in (("mlp.0", "fc1"), ("mlp.2", "fc2")):
            sd[f"{d}.{b}.weight"] = T(f"{s}.{a}.weight")
            sd[f"{d}.{b}.bias"] = T(f"{s}.{a}.bias")
    ln("decoder.ln", "model.decoder.layer_norm")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m for m in missing), missing  # HF has no key bias either
    return model


def argmax_last(x):
    x = np.asarray(x)
    return int(len(x) - 1 - np.argmax(x[::-1]))


def model_goldens(arch, seed, n_clips, mel_seed):
    import torch
    pkg = ge.load_package()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, f"{arch}.wtw")
        pkg.write_synthetic_weights(path, arch, seed)
        dims, tensors = read_wtw(path)
        model = hf_model(dims, tensors)
    rng = np.random.default_rng(mel_seed)
    mel = rng.uniform(-1.0, 1.5, size=(n_clips, dims["n_mels"], 2 * dims["n_audio_ctx"])).astype(np.float32)
    prompt = [50258, 50261, 50359, 50363] if dims["n_vocab"] > 50364 else [3, 5, 7, 11]
    out = {"arch": arch, "seed": np.int64(seed), "mel_seed": np.int64(mel_seed), "prompt": np.array(prompt, np.int64)}
    torch.set_num_threads(8)
    with torch.no_grad():
        enc = model.model.encoder(torch.from_numpy(mel)).last_hidden_state
        all_ids, top2, margins, last_logits = [], [], [], []
        for b in range(n_clips):
            ids = list(prompt)
            t2, lg = [], []
            for i in range(len(prompt) - 1, 30):  # reference loop bounds, whisper.cpp:367
                dec = model.model.decoder(input_ids=torch.tensor([ids]), encoder_hidden_states=enc[b:b + 1]).last_hidden_state
                logits = (dec[0, -1] @ model.proj_out.weight.T).numpy()
                nxt = argmax_last(logits)
                order = np.argsort(logits)[-2:][::-1]
                t2.append([order[0], order[1], logits[order[0]], logits[order[1]]])
                lg.append(logits)
                ids.append(nxt)
            all_ids.append(ids)
            top2.append(t2)
            last_logits.append(np.stack(lg))
    enc = enc.numpy()
    out["ids"] = np.array(all_ids, np.int64)
    out["top2"] = np.array(top2, np.float64)
    if arch == "micro":
        out["mel"] = mel
        out["enc_out"] = enc
        out["logits"] = np.stack(last_logits)
    else:
        out["enc_head"] = enc[:, 0:4, 0:8]
        out["enc_tail"] = enc[:, -4:, -8:]
        out["enc_rows"] = enc[:, ::250, :]
        out["enc_l2"] = np.sqrt((enc.astype(np.float64) ** 2).sum(axis=(1, 2)))
        out["logits_cols"] = np.stack(last_logits)[:, :, ::997]
    np.savez_compressed(os.path.join(GOLD, f"model_{arch}.npz"), **out)
    print(f"model goldens for {arch}: min top-2 margin {np.min(out['top2'][..., 2] - out['top2'][..., 3]):.4g}; ids[0] {all_ids[0][:10]}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-tiny", action="store_true")
    ap.add_argument("--skip-frontend", action="store_true")
    ap.add_argument("--only-frontend", action="store_true")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if not a.skip_frontend:
        frontend_goldens()
    if a.only_frontend:
        sys.exit(0)
    model_goldens("micro", 0, 3, 1234)
    if not a.skip_tiny:
        model_goldens("tiny", 0, 2, 1234)

"""Error of the encoder-attention variants against fp64 (GPU box): python tools/attn_accuracy.py"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "micro", 0)
eng = pkg.Engine(prefix, vocab, True)


def attn_ref(q, k, v):
    s = q @ k.T / 8.0
    s -= s.max(1, keepdims=True)
    p = np.exp(s)
    return (p / p.sum(1, keepdims=True)) @ v


names = {0: "fp32 MFMA", 1: "bf16 x3 split", 4: "fp16 x2 split", 3: "bf16 operands"}
for B, T, H, scale in [(2, 100, 2, 1.0), (1, 1500, 2, 1.0), (1, 1500, 2, 3.0)]:
    rng = np.random.default_rng(B * 1000 + T + H)
    qkv = (rng.standard_normal((B * T, 3 * 64 * H)) * scale).astype(np.float32)
    q64 = qkv.astype(np.float64).reshape(B, T, 3, H, 64)
    ref = np.stack([np.stack([attn_ref(q64[b, :, 0, h], q64[b, :, 1, h], q64[b, :, 2, h]) for h in range(H)], 1)
                    for b in range(B)])  # [B][T][H][64]
    print(f"B={B} T={T} H={H} operand scale {scale}: |ref| rms {np.sqrt((ref ** 2).mean()):.3f}")
    for v, name in names.items():
        eng.set_option("attn_variant", v)
        out = eng.dbg_encoder_attention(qkv, B, T, H).reshape(B, T, H, 64).astype(np.float64)
        err = np.abs(out - ref)
        print(f"  {name:14s}: max {err.max():.3e}  rms {np.sqrt((err ** 2).mean()):.3e}  p99.9 {np.quantile(err, 0.999):.3e}", flush=True)

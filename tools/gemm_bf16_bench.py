"""The bf16 storage mode's encoder GEMM and attention on whisper-base / whisper-tiny shapes (GPU box):
python tools/gemm_bf16_bench.py [base|tiny] [batch] -> TF/s per shape (WT_BF16_TILE=128/256/384 forces a tile)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
rng = np.random.default_rng(0)
arch = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if arch == "base" else 32)
d, L, Ld = (512, 6, 6) if arch == "base" else (384, 4, 4)
shapes = [("conv1", B * 3000, d, 256, 3, True, 1), ("conv2", B * 1500, d, 3 * d, 1, False, 1), ("qkv", B * 1500, 3 * d, d, 1, True, L),
          ("out", B * 1500, d, d, 1, False, L), ("fc1", B * 1500, 4 * d, d, 3, True, L), ("fc2", B * 1500, d, 4 * d, 1, False, L),
          ("cross-kv", B * 1500, 2 * Ld * d, d, 1, True, 1)]
total = 0.0
for name, M, N, K, epi, bf_out, reps in shapes:
    A = rng.standard_normal((M, K), dtype=np.float32)
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K)))
    _, ms = eng.dbg_gemm_bf16(A, W, np.zeros(N, np.float32), epi=epi, bf16_out=bf_out, iters=10)
    total += ms * reps
    print(f"{name:9s} {M}x{N}x{K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s  (x{reps} per batch)", flush=True)
    del A, W
print(f"encoder GEMM ms per batch: {total:.3f}")
T, H = 1500, d // 64
qkv = rng.standard_normal((B * T, 3 * 64 * H), dtype=np.float32)
_, ms = eng.dbg_encoder_attention_bf16(qkv, B, T, H, iters=5)
print(f"attention B={B}: {ms * 1e3:.1f} us per layer  {4.0 * B * H * T * T * 64 / ms / 1e9:.1f} TF/s")

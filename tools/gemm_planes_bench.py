"""The default encoder GEMM (fp16 planes) on the encoder's shapes (GPU box): python tools/gemm_planes_bench.py
-> TF/s algorithmic per shape, and the encoder attention per layer."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("conv1", B * 3000, 384, 256, 3, True), ("conv2", B * 1500, 384, 1152, 1, False), ("qkv", B * 1500, 1152, 384, 1, True),
          ("out", B * 1500, 384, 384, 1, False), ("fc1", B * 1500, 1536, 384, 3, True), ("fc2", B * 1500, 384, 1536, 1, False),
          ("cross-kv", B * 1500, 3072, 384, 1, False)]
total = 0.0
for name, M, N, K, epi, planes in shapes:
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    _, ms = eng.dbg_gemm_planes(A, W, np.zeros(N, np.float32), epi=epi, planes_out=planes, iters=10)
    reps = {"conv1": 1, "conv2": 1, "cross-kv": 1}.get(name, 4)
    total += ms * reps
    print(f"{name:9s} {M}x{N}x{K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s algorithmic  (x{reps} per batch)", flush=True)
print(f"encoder GEMM ms per batch: {total:.3f}")
T, H = 1500, 6
qkv = rng.standard_normal((B * T, 3 * 64 * H)).astype(np.float32)
_, ms = eng.dbg_encoder_attention_planes(qkv, B, T, H, iters=5)
print(f"attention B={B}: {ms * 1e3:.1f} us per layer  {4.0 * B * H * T * T * 64 / ms / 1e9:.1f} TF/s algorithmic")

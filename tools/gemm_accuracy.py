"""Error of the GEMM variants against fp64 on random operands (GPU box):
python tools/gemm_accuracy.py"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
rng = np.random.default_rng(5)
for M, N, K in [(512, 384, 384), (512, 384, 1536), (384, 1152, 384)]:
    A = rng.normal(0, 1, (M, K)).astype(np.float32)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    ref32 = A @ W.T  # numpy/BLAS fp32
    print(f"M={M} N={N} K={K}  |ref| rms {np.sqrt((ref ** 2).mean()):.3f}")
    print(f"  numpy fp32      : max abs err {np.abs(ref32 - ref).max():.3e}  rms {np.sqrt(((ref32 - ref) ** 2).mean()):.3e}")
    for v, name in [(0, "fp32 MFMA      "), (10, "bf16 x6 split  "), (11, "bf16 rounded   ")]:
        eng.set_option("gemm_variant", v)
        C = eng.dbg_gemm(A, W).astype(np.float64)
        print(f"  {name} : max abs err {np.abs(C - ref).max():.3e}  rms {np.sqrt(((C - ref) ** 2).mean()):.3e}  "
              f"mean signed {np.mean(C - ref):+.2e}", flush=True)
names = {0: "fp32 128x128", 4: "fp32 64x128", 10: "split-3", 11: "bf16"}
shapes = [(48000, 384, 384, 5, "out-proj"), (48000, 1152, 384, 1, "qkv"), (48000, 1536, 384, 3, "fc1"),
          (48000, 384, 1536, 5, "fc2"), (48000, 384, 1152, 3, "conv2"), (96000, 384, 256, 3, "conv1"),
          (48000, 3072, 384, 1, "cross-kv")]
print("shape".ljust(28) + "".join(n.rjust(14) for n in names.values()))
for M, N, K, epi, label in shapes:
    row = f"{label} {M}x{N}x{K}".ljust(28)
    for v in names:
        ms = eng.dbg_gemm_bench(M, N, K, epi=epi, variant=v, iters=8)
        row += f"{2.0 * M * N * K / ms / 1e9:10.1f} TF ".rjust(14)
    print(row, flush=True)

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
    for v, name in [(0, "fp32 MFMA      "), (13, "bf16 x3 split  "), (16, "bf16 x3 2blk/CU"), (17, "fp16 x2 split  "), (18, "fp16 x2 2blk/CU"), (11, "bf16 rounded   ")]:
        eng.set_option("gemm_variant", v)
        C = eng.dbg_gemm(A, W).astype(np.float64)
        print(f"  {name} : max abs err {np.abs(C - ref).max():.3e}  rms {np.sqrt(((C - ref) ** 2).mean()):.3e}  "
              f"mean signed {np.mean(C - ref):+.2e}", flush=True)

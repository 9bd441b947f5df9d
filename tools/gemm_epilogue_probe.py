"""Where a plane-GEMM launch spends its time (GPU box): python tools/gemm_epilogue_probe.py [B]
The same contraction with different epilogues (fp32 out / plane out / + GELU / + residual) and, through K, different
main-loop lengths: the differences isolate the GELU, the plane split and the store traffic from the MFMA loop."""
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
M = B * 1500
print(f"tile override WT_PLANE_TILE={os.environ.get('WT_PLANE_TILE', '-')}", flush=True)
for name, N, K in (("fc1-shape", 1536, 384), ("qkv-shape", 1152, 384), ("out-shape", 384, 384), ("fc2-shape", 384, 1536),
                   ("K32", 1536, 32), ("K768", 1536, 768)):
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if N == 384 else None
    row = []
    for label, epi, planes, res in (("bias->f32", 1, False, None), ("bias->planes", 1, True, None),
                                    ("bias+gelu->planes", 3, True, None), ("bias+resid->f32", 5, False, R)):
        if res is None and epi == 5:
            continue
        if epi == 5 and R is None:
            continue
        try:
            _, ms = eng.dbg_gemm_planes(A, W, np.zeros(N, np.float32), R=res, epi=epi, planes_out=planes, iters=10)
            row.append(f"{label} {ms * 1e3:7.1f} us ({2.0 * M * N * K / ms / 1e9:6.1f} TF/s)")
        except Exception as e:  # epilogue combination not instantiated
            row.append(f"{label} n/a")
    print(f"{name:10s} {M}x{N}x{K}: " + " | ".join(row), flush=True)

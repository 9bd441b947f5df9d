"""A/B of the plane-GEMM schedules in ONE process (GPU box): python tools/gemm_mode_ab.py [modes=0,1] [rounds=3] [B=32]
Alternates the modes of wt_dbg_set_plane_gemm_mode per shape and round, prints microseconds per launch (median and min)
and checks that every mode returns the bits of mode 0."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
rng = np.random.default_rng(0)
L = pkg.lib()
shapes = [("conv1", B * 3000, 384, 256, 3, True, 1), ("conv2", B * 1500, 384, 1152, 1, False, 1), ("qkv", B * 1500, 1152, 384, 1, True, 4),
          ("out", B * 1500, 384, 384, 5, False, 4), ("fc1", B * 1500, 1536, 384, 3, True, 4), ("fc2", B * 1500, 384, 1536, 5, False, 4)]
only = os.environ.get("AB_SHAPES")
tot = {m: 0.0 for m in modes}
for name, M, N, K, epi, planes, reps in shapes:
    if only and name not in only.split(","):
        continue
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if epi & 4 else None
    bias = rng.standard_normal(N).astype(np.float32)
    t = {m: [] for m in modes}
    ref = None
    for r in range(rounds):
        for m in modes:
            assert L.wt_dbg_set_plane_gemm_mode(m) == 0
            C, ms = eng.dbg_gemm_planes(A, W, bias, R=R, epi=epi, planes_out=planes, iters=10)
            t[m].append(ms * 1e3)
            if ref is None:
                ref = C
            elif r == 0 and not np.array_equal(ref, C):
                print(f"  !! mode {m} differs from mode {modes[0]}: max |d| = {np.abs(ref - C).max():.3e}", flush=True)
    line = f"{name:6s} {M}x{N}x{K} epi={epi}:"
    for m in modes:
        med, mn = float(np.median(t[m])), min(t[m])
        tot[m] += med * reps
        line += f"  mode{m} {med:7.1f} us (min {mn:7.1f}) {2.0 * M * N * K / med / 1e6:6.1f} TF/s"
    print(line, flush=True)
print("encoder GEMM ms per batch (medians): " + "  ".join(f"mode{m} {tot[m] / 1e3:.3f}" for m in modes))

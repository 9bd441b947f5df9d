import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
rng = np.random.default_rng(1)
M, N, K = 256, 128, 256
for sa, sw in [(1.0, 1.0), (0.05, 1.0), (1.0, 0.001), (0.05, 0.001), (0.001, 1.0)]:
    A = (rng.standard_normal((M, K)) * sa).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K) * sw).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    for v in (0, 13, 17):
        eng.set_option("gemm_variant", v)
        C = eng.dbg_gemm(A, W).astype(np.float64)
        print(f"A scale {sa} W scale {sw} variant {v}: rel rms err {np.sqrt(((C-ref)**2).mean())/np.sqrt((ref**2).mean()):.3e}")

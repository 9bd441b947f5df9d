"""One GEMM shape / variant for profiling: python tools/gemm_one.py M N K epi variant [iters]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
M, N, K, epi, variant = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 4
ms = eng.dbg_gemm_bench(M, N, K, epi=epi, variant=variant, iters=iters)
print(f"{M}x{N}x{K} variant {variant}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TF")

"""Decoder GEMM launch-time sweep (run on the GPU box): microseconds per back-to-back launch."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
for label, kind, N, K in (("out-proj resid", 0, 384, 384), ("fc2 resid", 0, 384, 1536), ("combine+resid", 2, 384, 384)):
    print(label.ljust(16) + "".join(f"  w{w}: {eng.dbg_dec_gemm_bench(kind, 32, N, K, w):6.1f} us" for w in (4, 8, 16)), flush=True)
for label, N in (("LN qkv", 1152), ("LN q", 384), ("LN fc1", 1536)):
    print(label.ljust(16) + f"  {eng.dbg_dec_gemm_bench(1, 32, N, 384):6.1f} us", flush=True)

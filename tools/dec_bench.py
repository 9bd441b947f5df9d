"""Decoder GEMM launch times (GPU box): python tools/dec_bench.py  -> us per launch, tiny shapes, 32 and 128 rows."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
for label, kind, N, K in (("out-proj resid", 0, 384, 384), ("fc2 resid", 0, 384, 1536), ("combine+resid", 2, 384, 384),
                          ("LN qkv", 1, 1152, 384), ("LN fc1", 1, 1536, 384)):
    print(label.ljust(16) + "".join(f"  rows {r:3d}: {eng.dbg_dec_gemm_bench(kind, 32, N, K, r):6.1f} us" for r in (32, 64, 128)), flush=True)

"""Absorbed cross-attention launch time against the key-chunk count (GPU box): python tools/cross_abs_bench.py [B]"""
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
H, T = 6, 1500
d = 64 * H
E = rng.standard_normal((B, T, d)).astype(np.float32)
wv = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
bv = np.zeros(d, np.float32)
for nq in (1, 2):
    qp = (rng.standard_normal((nq * B, H * d)) * (3.0 / np.sqrt(d))).astype(np.float32)
    for chunks in (1, 2, 4, 6, 8, 12, 16):
        _, us = eng.dbg_cross_absorbed(qp, E, wv, bv, B, H, T, chunks, nq, iters=20)
        tiles = -(-((T + 31) // 32) // chunks)
        print(f"B={B} nq={nq} chunks={chunks:2d} ({B * chunks:4d} blocks x {tiles:2d} tiles): {us:7.1f} us  "
              f"{B * T * d * 4 / us / 1e6:6.2f} TB/s of E", flush=True)

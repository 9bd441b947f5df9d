"""Main-loop ablations of the 192 x 384 plane-GEMM tile (GPU box): WT_GEMM_ABL=0..3 python tools/gemm_ablate.py
0 = the kernel, 1 = no LDS-DMA after the first k-tile, 2 = no MFMAs, 3 = LDS-DMA only.  Results of 1..3 are garbage by
construction; only the times matter.  K = 1536 (48 k-tiles: the loop dominates) and K = 384, fp32 output."""
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
M = 48000
for N, K in ((384, 1536), (1536, 384), (384, 384)):
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    _, ms = eng.dbg_gemm_planes(A, W, np.zeros(N, np.float32), epi=1, planes_out=False, iters=10, n_cu=256)
    print(f"ABL={os.environ.get('WT_GEMM_ABL', '0')} {M}x{N}x{K}: {ms * 1e3:8.1f} us ({2.0 * M * N * K / ms / 1e9:6.1f} TF/s algorithmic)", flush=True)

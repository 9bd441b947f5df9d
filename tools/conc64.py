"""Decoder concurrency probe at B = 32 and B = 64 (GPU box): python tools/conc64.py"""
import ctypes
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "tiny", 0)
eng = pkg.Engine(prefix, vocab, True)
eng.set_option("stop_at_eot", 0)
lib = pkg.lib()
rng = np.random.default_rng(0)
for B in (64, 32):
    d_mel = torch.from_numpy(rng.uniform(-1, 1.5, size=(B, 80, 3000)).astype(np.float32)).cuda()
    for _ in range(9):
        eng.pipeline_submit_dev(d_mel.data_ptr(), B)
        eng.pipeline_collect()

    def conc(n_dec, n_enc):
        d = (ctypes.c_float * 4)()
        e = ctypes.c_float()
        rc = lib.wt_dbg_concurrency(eng._h, ctypes.c_void_p(d_mel.data_ptr()), B, n_dec, n_enc, d, ctypes.byref(e))
        assert rc == 0, rc
        return [round(d[i], 2) for i in range(n_dec)], round(e.value, 2)

    conc(1, 1)
    for n_dec, n_enc in [(1, 0), (2, 0), (3, 0), (0, 2), (1, 2), (2, 3), (3, 4)]:
        d, e = conc(n_dec, n_enc)
        print(f"B={B}: {n_dec} decodes + {n_enc} encoder passes: decode ms {d}, encoder passes {e} ms"
              + (f" ({e / n_enc:.2f} each)" if n_enc else ""), flush=True)

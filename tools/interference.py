"""What does a chain of dependent launches on a second stream cost the encoder, and vice versa?
python tools/interference.py   (GPU box)"""
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
lib = pkg.lib()
B = 32
rng = np.random.default_rng(0)
d_mel = torch.from_numpy(rng.uniform(-1, 1.5, size=(B, 80, 3000)).astype(np.float32)).cuda()
eng.encdec_tokens_batch_dev(d_mel.data_ptr(), B)


def probe(n_enc, chain, blocks):
    a, b = ctypes.c_float(), ctypes.c_float()
    rc = lib.wt_dbg_interference(eng._h, ctypes.c_void_p(d_mel.data_ptr()), B, n_enc, chain, blocks,
                                 ctypes.byref(a), ctypes.byref(b))
    assert rc == 0, rc
    return a.value, b.value


probe(1, 10, 1)
for n_enc, chain, blocks in [(4, 0, 1), (0, 6000, 1), (0, 6000, 64), (4, 6000, 1), (4, 6000, 64), (4, 6000, 512),
                            (4, 1500, 1), (4, 600, 1)]:
    e, c = probe(n_enc, chain, blocks)
    print(f"encoders {n_enc} chain {chain:5d} x {blocks:3d} blocks: encoder {e / max(n_enc, 1):7.3f} ms each, "
          f"chain {c:8.3f} ms = {1e3 * c / max(chain, 1):6.2f} us per launch", flush=True)


def conc(n_dec, n_enc):
    d = (ctypes.c_float * 4)()
    e = ctypes.c_float()
    rc = lib.wt_dbg_concurrency(eng._h, ctypes.c_void_p(d_mel.data_ptr()), B, n_dec, n_enc, d, ctypes.byref(e))
    assert rc == 0, eng.last_error() if hasattr(eng, "last_error") else rc
    return [round(d[i], 2) for i in range(n_dec)], round(e.value, 2)


for _ in range(9):
    eng.pipeline_submit_dev(d_mel.data_ptr(), B)
    eng.pipeline_collect()
conc(1, 1)
for n_dec, n_enc in [(1, 0), (2, 0), (3, 0), (4, 0), (0, 2), (1, 2), (2, 3), (3, 4), (4, 4)]:
    d, e = conc(n_dec, n_enc)
    print(f"{n_dec} decodes + {n_enc} encoder passes: decode ms {d}, encoder passes {e} ms"
          + (f" ({e / n_enc:.2f} each)" if n_enc else ""), flush=True)

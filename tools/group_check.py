"""dec_group 4 at 32-clip batches (128 rows per decoder pass, ONE key chunk per clip in the absorbed cross-attention) against the
synchronous call (eight chunks): where ids differ, how close were the fp32 logits of the two candidates?"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from conftest import DevBuf
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "tiny", 0)
e = pkg.Engine(prefix, vocab, True)
e.set_option("stop_at_eot", 0)
rng = np.random.default_rng(96128)
mels = [rng.uniform(-1.0, 1.5, size=(32, 80, 3000)).astype(np.float32) for _ in range(4)]
want = [e.encdec_tokens_batch(m) for m in mels]
dev = [DevBuf(m) for m in mels]
e.set_option("dec_group", 4)
for k in range(4): e.pipeline_submit_dev(dev[k].data_ptr(), 32)
got = [e.pipeline_collect() for _ in range(4)]
for chunks in (1, 2, 8):
    e.set_option("abs_chunks", chunks)
    ids_c, _ = e.encdec_tokens_batch(mels[2])
    print("synchronous call with abs_chunks", chunks, "equals the default synchronous ids:", np.array_equal(ids_c, want[2][0]), "equals the group-of-4 ids:", np.array_equal(ids_c, got[2][0]))
e.set_option("abs_chunks", 0)
for k in range(4):
    for b in range(32):
        if not np.array_equal(want[k][0][b], got[k][0][b]):
            pos = int(np.argmax(want[k][0][b] != got[k][0][b]))
            ids, n, enc, lg = e.encdec_debug_batch(mels[k][b:b + 1])
            row = lg[0, pos - 4]
            a, c = int(want[k][0][b][pos]), int(got[k][0][b][pos])
            print(f"batch {k} clip {b} position {pos}: synchronous id {a} (logit {row[a]:.7f}), grouped id {c} (logit {row[c]:.7f}), margin {row[a] - row[c]:.2e}, top-2 margin {np.sort(row)[-1] - np.sort(row)[-2]:.2e}")

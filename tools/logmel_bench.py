"""Log-mel front end alone (GPU box): python tools/logmel_bench.py -> ms per 32-clip batch, and the largest deviation
from the oracle's (bit-exact restatement of the reference's) log-mel on a noise clip."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "tiny", 0)
eng = pkg.Engine(prefix, vocab, True)
B = 32
pcm = np.clip(np.random.default_rng(7).normal(0.0, 0.1, size=(B, eng.pcm_len)), -1, 1).astype(np.float32)
d_pcm = eng.device_array(pcm)
d_mel = eng.device_array(np.zeros((B,) + eng.mel_shape, np.float32))
for _ in range(3):
    eng.logmel_batch_dev(d_pcm.data_ptr(), B, d_mel.data_ptr())
t0 = time.perf_counter()
for _ in range(20):
    eng.logmel_batch_dev(d_pcm.data_ptr(), B, d_mel.data_ptr())
print(f"log-mel: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per batch of {B}")
mel = d_mel.download()
orc = ge.load_oracle()
ref = orc.frontend().logmel(pcm[0], eng.filters(), 8)
print("max |mel - oracle| on clip 0:", float(np.abs(mel[0] - ref).max()))

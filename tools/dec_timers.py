"""Per-launch-class device time of one decoder chain (GPU box), eager launches with an event pair around each:
WT_DEC_KERNEL_TIMERS=1 python tools/dec_timers.py [cross_absorb 0|1] [batch]   (the engine prints the table on stderr)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WT_DEC_KERNEL_TIMERS"] = "1"
import __graft_entry__ as ge

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "tiny", 0)
eng = pkg.Engine(prefix, vocab, True)
eng.set_option("stop_at_eot", 0)
eng.set_option("use_graphs", 0)
eng.set_option("cross_absorb", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mel = np.random.default_rng(1).uniform(-1.0, 1.5, size=(B,) + eng.mel_shape).astype(np.float32)
for _ in range(3):
    ids, n = eng.encdec_tokens_batch(mel)
t = eng.timings()
print(f"cross_absorb={eng.get_option('cross_absorb')} batch={B}: encoder {t.encoder_ms:.3f} ms, cross-kv {t.cross_kv_ms:.3f} ms, "
      f"decoder {t.decoder_ms:.3f} ms (eager, with event pairs)", flush=True)

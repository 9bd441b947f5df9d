"""Does a second engine in the same process run the pipeline as fast as the first?  (GPU box)
python tools/second_engine_probe.py [keep]   keep = do not close the first engine before creating the second"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import bench

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "tiny", 0)
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
engines = []
for i in range(3):
    e = pkg.Engine(prefix, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_option("kernel_timers", 4)
    d_mel = torch.from_numpy(bench.synthetic_mel(0, 32, e.mel_shape)).cuda()
    torch.cuda.synchronize()
    dt, _, n, det = bench.timed_leg(e, d_mel.data_ptr(), 32, 40, 12, 10, torch.cuda.synchronize)
    t = e.timings()
    print(f"engine {i}: {32 * 40 * 30 / dt:9.1f} audio-sec/s  enc {t.encoder_ms:.2f} ms dec chain {t.decoder_ms:.2f} ms  " +
          " ".join(f"{k}={v['avg_launch_us']:.0f}us" for k, v in det.items()), flush=True)
    if keep:
        engines.append(e)
    else:
        e.close()

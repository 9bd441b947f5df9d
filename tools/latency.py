"""Single-clip latency of the reference's own entry point (Engine::transcribe(samples)) and of
small batches, on the GPU box: python tools/latency.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "tiny", 0)
eng = pkg.Engine(prefix, vocab, True)
if os.environ.get("WT_NO_GRAPHS"):
    eng.set_option("use_graphs", 0)
rng = np.random.default_rng(0)
pcm = np.clip(rng.normal(0, 0.1, 480000), -1, 1).astype(np.float32)
eng.transcribe(pcm)
for stop in (1, 0):
    eng.set_option("stop_at_eot", stop)
    eng.transcribe(pcm)  # the first call of a configuration captures its decoder graphs (every pipeline slot at once)
    t0 = time.perf_counter()
    for _ in range(10):
        eng.transcribe(pcm)
    dt = (time.perf_counter() - t0) / 10
    t = eng.timings()
    print(f"transcribe(1 clip, stop_at_eot={stop}): {1e3 * dt:.2f} ms wall  (logmel {t.logmel_ms:.2f}  encoder {t.encoder_ms:.2f}  "
          f"cross-kv {t.cross_kv_ms:.2f}  decoder {t.decoder_ms:.2f} ms, {t.decoder_steps} argmax steps)", flush=True)
eng.set_option("stop_at_eot", 0)
for absorb in ((1, 0) if os.environ.get("WT_LATENCY_BOTH_FORMS") else (None,)):
    if absorb is not None:
        eng.set_option("cross_absorb", absorb)
        print(f"cross_absorb = {absorb}")
    for B in (1, 2, 4, 8, 16, 32):
        mel = rng.uniform(-1, 1.5, size=(B, 80, 3000)).astype(np.float32)
        eng.encdec_tokens_batch(mel)
        t0 = time.perf_counter()
        for _ in range(5):
            eng.encdec_tokens_batch(mel)
        dt = (time.perf_counter() - t0) / 5
        t = eng.timings()
        print(f"encdec B={B:2d}: {1e3 * dt:7.2f} ms wall  encoder {t.encoder_ms:6.2f}  cross-kv {t.cross_kv_ms:5.2f}  decoder {t.decoder_ms:6.2f} ms "
              f"-> {B * 30 / dt:9.0f} audio-sec/s", flush=True)

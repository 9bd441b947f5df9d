"""Encoder error of the three contraction forms on weights with trained-checkpoint statistics (GPU box):
python tools/outlier_probe.py [gain ...]  -> max |enc - oracle| per form, relative to the output scale."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge
from wtw import read_wtw, write_wtw

pkg, orc = ge.load_package(), ge.load_oracle()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "tiny", 0)
dims, t = read_wtw(prefix + ".wtw")
mel = np.random.default_rng(9).uniform(-1.0, 1.5, size=(1, 80, 3000)).astype(np.float32)
for spec in (sys.argv[1:] or ["30:6:1", "1000:1:0", "30:6:0", "1:0:1"]):
    gain, nch, heavy = spec.split(":")
    gain, nch, heavy = float(gain), int(nch), int(heavy)
    rng = np.random.default_rng(1234)
    out = {}
    for k, v in t.items():
        a = np.array(v, dtype=np.float32)
        if (k.startswith("encoder.") and k.endswith("_ln.weight")) or k == "encoder.ln_post.weight":
            if nch:
                a[rng.choice(a.size, nch, replace=False)] *= gain
        elif heavy and k.startswith("encoder.blocks.") and k.endswith(".weight") and a.ndim == 2:
            mask = rng.random(a.shape) < 0.02
            a[mask] *= rng.uniform(8, 40, int(mask.sum())).astype(np.float32)
        out[k] = a
    p = os.path.join(tmp, "adv")
    write_wtw(p + ".wtw", dims, out)
    m = orc.Model(p + ".wtw")
    ref = m.encode(mel[0], 16)
    m.close()
    e = pkg.Engine(p, vocab, True)
    e.set_option("stop_at_eot", 0)
    e.set_option("max_tokens", 5)
    res = {}
    for name, gv, av in (("fp16x2", -1, 4), ("bf16x3", 16, 1), ("fp32mfma", 0, 0)):
        e.set_option("gemm_variant", gv)
        e.set_option("attn_variant", av)
        _, _, enc, _ = e.encdec_debug_batch(mel, want_logits=False)
        res[name] = enc[0]
    fb = e.get_option("f16_fallbacks")
    e.close()
    sc = float(np.abs(ref).max())
    print(f"gain x{gain:g} on {nch} channels, heavy tails {heavy}: |enc| max {sc:.3g}, fallbacks {fb}; " +
          "  ".join(f"{k} err {np.abs(v - ref).max():.3g} ({np.abs(v - ref).max() / sc:.2g} rel)" for k, v in res.items()) +
          f"  fp16x2-vs-bf16x3 {np.abs(res['fp16x2'] - res['bf16x3']).max():.3g}", flush=True)

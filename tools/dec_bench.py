"""Decoder GEMM launch times (GPU box): python tools/dec_bench.py  -> us per launch, tiny shapes, 32 and 128 rows."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
for label, kind, N, K in (("out-proj resid", 0, 384, 384), ("fc2 resid", 0, 384, 1536), ("combine+resid", 2, 384, 384),
                          ("LN qkv", 1, 1152, 384), ("LN fc1", 1, 1536, 384)):
    print(label.ljust(16) + "".join(f"  rows {r:3d}: {eng.dbg_dec_gemm_bench(kind, 32, N, K, r):6.1f} us" for r in (32, 64, 128)), flush=True)
# the persistent logits kernel at whisper's vocabulary.  Round 4: 18.0 / 28.4 / 41.5 / 40.2 us for 32 / 64 / 96 / 128 rows
# (512 resident blocks) — time follows the rows, not the 80 MB of weights: an 8-wavefront form in which the two row
# tiles of a pair share every weight fetch through the vector L1 measured the same 28.0 us (and 141.1 against 141.4 k
# audio-sec/s end to end); a tile iteration is its load latency, one barrier and the argmax shuffles, not the stream.
for r in (32, 64, 96, 128):
    print(f"LN logits rows {r:3d}: {eng.dbg_dec_gemm_bench(3, 32, 51865, 384, r):6.1f} us  "
          f"(WT_LOGITS_BLOCKS={os.environ.get('WT_LOGITS_BLOCKS', 'default 512')})", flush=True)

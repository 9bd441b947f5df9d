"""GEMM tile-variant sweep on the encoder's shapes (run on the GPU box):
python tools/gemm_bench.py -> TFLOP/s per (shape, variant)."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
tmp = tempfile.mkdtemp()
prefix, vocab = ge._assets(tmp, "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
names = {0: "128x128x32", 1: "128x128x64", 2: "128x128x32db", 3: "128x64x32", 4: "64x128x32", 5: "128x64x32db", 6: "64x64x32", 7: "128x128pf2", 8: "128x64pf2", 9: "192x128x32"}
shapes = [(48000, 384, 384, 5, "out-proj"), (48000, 1152, 384, 1, "qkv"), (48000, 1536, 384, 3, "fc1"),
          (48000, 384, 1536, 5, "fc2"), (48000, 384, 1152, 3, "conv2"), (96000, 384, 256, 3, "conv1"),
          (48000, 3072, 384, 1, "cross-kv")]
variants = [int(v) for v in sys.argv[1:]] or list(names)
print("shape".ljust(28) + "".join(names[v].rjust(14) for v in variants))
for M, N, K, epi, label in shapes:
    row = f"{label} {M}x{N}x{K}".ljust(28)
    for v in variants:
        ms = eng.dbg_gemm_bench(M, N, K, epi=epi, variant=v, iters=8)
        row += f"{2.0 * M * N * K / ms / 1e9:10.1f} TF ".rjust(14)
    print(row, flush=True)

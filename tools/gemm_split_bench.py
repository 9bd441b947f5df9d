"""The selectable fp32-storage GEMM variants (k_gemm.hip) on the encoder shapes (GPU box):
python tools/gemm_split_bench.py [variants...]; the default plane GEMM: tools/gemm_planes_bench.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
prefix, vocab = ge._assets(tempfile.mkdtemp(), "micro", 0)
eng = pkg.Engine(prefix, vocab, True)
names = {0: "fp32 128x128", 11: "bf16 operands", 13: "bf16x3 k16", 16: "bf16x3 2blk/CU", 17: "fp16x2", 18: "fp16x2 2blk"}  # the variants the library keeps (k_gemm.hip)
shapes = [(48000, 384, 384, 5, "out-proj"), (48000, 1152, 384, 1, "qkv"), (48000, 1536, 384, 3, "fc1"),
          (48000, 384, 1536, 5, "fc2"), (48000, 384, 1152, 3, "conv2"), (96000, 384, 256, 3, "conv1"),
          (48000, 3072, 384, 1, "cross-kv")]
variants = [int(v) for v in sys.argv[1:]] or list(names)
print("shape".ljust(28) + "".join(names.get(v, str(v)).rjust(14) for v in variants))
tot = {v: 0.0 for v in variants}
mult = {"out-proj": 4, "qkv": 4, "fc1": 4, "fc2": 4, "conv2": 1, "conv1": 1, "cross-kv": 1}
for M, N, K, epi, label in shapes:
    row = f"{label} {M}x{N}x{K}".ljust(28)
    for v in variants:
        ms = eng.dbg_gemm_bench(M, N, K, epi=epi, variant=v, iters=8)
        tot[v] += ms * mult[label]
        row += f"{2.0 * M * N * K / ms / 1e9:10.1f} TF ".rjust(14)
    print(row, flush=True)
print("encoder GEMM ms per batch".ljust(28) + "".join(f"{tot[v]:11.2f} ms".rjust(14) for v in variants))

# encoder attention A/B of two library builds on ONE box: bash tools/ab_attn.sh tools/bin/lib_a.so tools/bin/lib_b.so
# (fp32-accurate form in bursts, bf16 form in bursts; three alternating rounds)
set -o pipefail
for i in 1 2 3; do
  for l in "$@"; do
    echo -n "$l: "; WT_LIB_PATH=$PWD/$l python tools/gemm_planes_bench.py 2>/dev/null | tail -1 || exit 1
    echo -n "$l bf16: "; WT_LIB_PATH=$PWD/$l python tools/gemm_bf16_bench.py 2>/dev/null | tail -1 || exit 1
  done
done

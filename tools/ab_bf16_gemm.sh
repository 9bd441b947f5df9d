# bf16 encoder GEMM A/B of two library builds on ONE box: bash tools/ab_bf16_gemm.sh tools/bin/lib_a.so tools/bin/lib_b.so
set -o pipefail
for i in 1 2 3; do
  for l in "$@"; do
    echo "$l:"; WT_LIB_PATH=$PWD/$l python tools/gemm_bf16_bench.py 2>/dev/null | tail -9 | head -8 || exit 1
  done
done

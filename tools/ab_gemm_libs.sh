# plane GEMM A/B of two library builds on ONE box: bash tools/ab_gemm_libs.sh tools/bin/lib_a.so tools/bin/lib_b.so
set -o pipefail
for i in 1 2 3; do
  for l in "$@"; do
    echo "$l:"; WT_LIB_PATH=$PWD/$l python tools/gemm_planes_bench.py 2>/dev/null || exit 1
  done
done

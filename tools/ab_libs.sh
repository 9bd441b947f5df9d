# end-to-end A/B of library builds on ONE box: bash tools/ab_libs.sh tools/bin/lib_a.so tools/bin/lib_g.so
set -o pipefail
for i in 1 2 3; do
  for l in "$@"; do
    WT_LIB_PATH=$PWD/$l python bench.py --steps 60 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'], d['stage_ms_per_step']['encoder_ms'], d['roofline_isolated']['gemm_planes']['avg_launch_us'], d['roofline_isolated']['encoder_attention_planes']['avg_launch_us'])" || exit 1
  done
done

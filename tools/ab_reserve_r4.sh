# CUs per XCD kept free of pipelined encoder work, final round-4 build, on ONE box: bash tools/ab_reserve_r4.sh
set -o pipefail
for i in 1 2; do
  for r in 4 0 1 2 3 5 6; do
    WT_ENC_CU_RESERVE=$r python bench.py --steps 120 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve $r', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
  done
done

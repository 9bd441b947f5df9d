# batches per decoder chain x CUs per XCD kept free of encoder work x batches in flight, on ONE box: bash tools/ab_group.sh
set -o pipefail
for i in 1 2; do
  for cfg in "2 4 10" "2 4 14" "3 4 16" "3 3 16" "3 2 16" "4 4 20" "4 3 20" "4 2 20" "4 1 20" "4 0 20" "4 2 24"; do
    set -- $cfg
    WT_ENC_CU_RESERVE=$2 python bench.py --steps 120 --dec-group $1 --depth $3 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $1 reserve $2 depth $3', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
  done
done

# the driver's 20-step command by CU reservation and batches in flight, ONE box: bash tools/ab_short_run.sh
set -o pipefail
for i in 1 2 3; do
  for cfg in "4 10" "5 10" "6 10" "4 8" "4 7" "5 8" "4 12"; do
    set -- $cfg
    WT_ENC_CU_RESERVE=$1 python bench.py --gpus 1 --steps 20 --warmup 5 --depth $2 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve $1 depth $2', d['value'], d['ms_per_step'])" || exit 1
  done
done

# the driver's 20-step command by announced last batches, on ONE box: bash tools/ab_tail_r4.sh
set -o pipefail
for i in 1 2 3; do
  for t in 2 3 4; do
    WT_STREAM_PROBE_TRACE= python bench.py --gpus 1 --steps 20 --warmup 5 --tail $t --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail $t', d['value'], d['ms_per_step'])" || exit 1
  done
done

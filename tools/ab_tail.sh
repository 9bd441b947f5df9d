set -o pipefail
for t in 0 2 1 3 0 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --tail $t --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail=$t', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
done

# A/B of the driver's command: encoder passes of the pipeline fill on the whole chip (default) against always masked
set -o pipefail
for i in 1 2 3; do
  for m in 0 1; do
    WT_ENC_MASK_ALWAYS=$m python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask_always=$m', d['value'], d['ms_per_step'])" || exit 1
  done
done

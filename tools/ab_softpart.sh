set -o pipefail
for p in 0 64 96 128 0; do
  if [ $p = 0 ]; then unset WT_DEC_PARTITION; else export WT_DEC_PARTITION=$p; fi
  python bench.py --gpus 1 --steps 60 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dec_partition=$p', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
done

set -o pipefail
for n in 3 4 5 3; do
  for st in 20 60; do
    WT_DEC_STREAMS=$n python bench.py --gpus 1 --steps $st --warmup 5 --depth 12 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dec_streams=$n steps=$st', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
  done
done

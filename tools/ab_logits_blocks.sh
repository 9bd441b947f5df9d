set -o pipefail
for n in 512 0 512 0 512 0; do
  WT_LOGITS_BLOCKS=$n python bench.py --gpus 1 --steps 60 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('logits_blocks=$n', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
done

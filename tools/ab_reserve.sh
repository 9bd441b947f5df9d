# CUs per XCD left to the decoders (steady state, 60 steps), alternating
set -o pipefail
for i in 1 2; do
  for r in 8 7 6; do
    WT_ENC_CU_RESERVE=$r python bench.py --gpus 1 --steps 60 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve=$r', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
  done
done

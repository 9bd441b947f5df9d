# CUs per XCD left to the decoders, on the driver's 20-step command (the drain weighs more there)
set -o pipefail
for r in 8 6 10 12 8; do
  WT_ENC_CU_RESERVE=$r python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve=$r', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
done

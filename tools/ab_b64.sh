set -o pipefail
for i in 1 2; do
  python bench.py --gpus 1 --steps 60 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32 paired', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
  python bench.py --gpus 1 --steps 30 --batch 64 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b64', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
done

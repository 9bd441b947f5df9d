set -o pipefail
for a in "--steps 40 --depth 5" "--steps 40 --depth 10" "--steps 100 --depth 10" "--steps 40 --depth 5 --warmup 6"; do
  python bench.py --arch base --batch 64 --bf16 $a --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
done

# cross-attention key chunks per clip in the paired pipeline: driver's command (20 steps) and a long run
set -o pipefail
for c in 0 1 0 1; do
  for st in 20 60; do
    python bench.py --gpus 1 --steps $st --warmup 5 --abs-chunks $c --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks=$c steps=$st', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['avg_launch_us'])" || exit 1
  done
done

# round 4: CUs of the pipelined encoder stream (WT_ENC_CU_KEEP) — 250 lets the 192-row ping-pong tiles of the N = 384 shapes run
# in one round; 224 (reserve 4, default) makes the tile choice take the 256-row kernel there
set -o pipefail
for i in 1 2; do
  for k in 224 240 248 250 256; do
    WT_ENC_CU_KEEP=$k python bench.py --gpus 1 --steps 80 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keep=$k', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['avg_launch_us'])" || exit 1
  done
done

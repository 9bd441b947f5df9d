# round 4: CU reservation and cross-attention key chunks after the loader-wave rewrite of the absorbed cross-attention
set -o pipefail
run() { # label, env..., args...
  local label=$1; shift
  env "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" || exit 1
}
for i in 1 2; do
  for r in 8 6 5 4 3; do
    run "reserve=$r" WT_ENC_CU_RESERVE=$r python bench.py --gpus 1 --steps 80 --no-cpu-baseline --no-fp32-leg 2>/dev/null
  done
done
for i in 1 2; do
  for c in 0 1 2 3 4 6 8; do
    run "abs_chunks=$c" python bench.py --gpus 1 --steps 80 --abs-chunks $c --no-cpu-baseline --no-fp32-leg 2>/dev/null
  done
done

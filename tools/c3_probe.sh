# configs[3] (whisper-base, batch 64, bf16): absorbed against cached cross-attention, key chunks, CU reservation
set -o pipefail
run() {  # label, env assignments, bench args
  env $2 python bench.py --arch base --batch 64 --bf16 --steps 40 --no-cpu-baseline --no-fp32-leg $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['frac'], d['roofline']['isolated']['frac'])" || exit 1
}
run cached A=1 "--cross-absorb 0"
run absorbed A=1 ""
run absorbed_c1 A=1 "--abs-chunks 1"
run absorbed_c3 A=1 "--abs-chunks 3"
run absorbed_c4 A=1 "--abs-chunks 4"
run absorbed_res4 WT_ENC_CU_RESERVE=4 ""
run absorbed_res6 WT_ENC_CU_RESERVE=6 ""
run absorbed A=1 ""

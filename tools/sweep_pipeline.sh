# pipeline knobs A/B on one box (GPU)
set -o pipefail
run() { name=$1; shift; env "$@" python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fp32-leg ${EXTRA} > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/sw_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['stage_ms_per_step'], flush=True)
except Exception as e:
    print(n,'ERR',e, open(f'gpurun_out/sw_{n}.err').read()[-300:])
PY
}
EXTRA="--abs-chunks 4" run b32_c4_res8 WT_ENC_CU_RESERVE=8
EXTRA="--batch 64 --abs-chunks 4" run b64_c4_res8 WT_ENC_CU_RESERVE=8
EXTRA="--batch 64 --abs-chunks 2" run b64_c2_res8 WT_ENC_CU_RESERVE=8
EXTRA="--batch 64 --abs-chunks 2" run b64_c2_res4 WT_ENC_CU_RESERVE=4
EXTRA="--batch 64 --abs-chunks 4" run b64_c4_res4 WT_ENC_CU_RESERVE=4
EXTRA="--batch 64 --abs-chunks 2" run b64_c2_res6 WT_ENC_CU_RESERVE=6
EXTRA="--batch 64 --abs-chunks 2 --depth 3" run b64_c2_res4_d3 WT_ENC_CU_RESERVE=4
EXTRA="--batch 64 --abs-chunks 2" run b64_c2_res4_dec2 WT_ENC_CU_RESERVE=4 WT_DEC_STREAMS=2

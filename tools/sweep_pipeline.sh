# pipeline knobs A/B on one box (GPU)
set -o pipefail
run() { name=$1; shift; env "$@" python bench.py --steps 72 --warmup 12 --no-cpu-baseline --no-fp32-leg ${EXTRA} > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/sw_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['stage_ms_per_step'], flush=True)
except Exception as e:
    print(n,'ERR',e, open(f'gpurun_out/sw_{n}.err').read()[-300:])
PY
}
EXTRA="--depth 6" run d6_res8 WT_ENC_CU_RESERVE=8
EXTRA="--depth 8" run d8_res8 WT_ENC_CU_RESERVE=8
EXTRA="--depth 10" run d10_res8 WT_ENC_CU_RESERVE=8
EXTRA="--depth 12" run d12_res8 WT_ENC_CU_RESERVE=8
EXTRA="--depth 10" run d10_res4 WT_ENC_CU_RESERVE=4
EXTRA="--depth 10" run d10_res6 WT_ENC_CU_RESERVE=6
EXTRA="--depth 10 --abs-chunks 3" run d10_res8_c3 WT_ENC_CU_RESERVE=8
EXTRA="--depth 10 --abs-chunks 4" run d10_res8_c4 WT_ENC_CU_RESERVE=8
EXTRA="--depth 10" run d10_res10 WT_ENC_CU_RESERVE=10

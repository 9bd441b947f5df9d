# pipeline knobs A/B on one box (GPU): reserve (CUs per XCD kept free of pipelined encoder work), strict partition
set -o pipefail
run() { name=$1; shift; env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-leg ${EXTRA} > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/sw_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['stage_ms_per_step'], flush=True)
except Exception as e:
    print(n,'ERR',e, open(f'gpurun_out/sw_{n}.err').read()[-300:])
PY
}
EXTRA="" run res4 WT_ENC_CU_RESERVE=4
EXTRA="" run res8 WT_ENC_CU_RESERVE=8
EXTRA="" run res8part WT_ENC_CU_RESERVE=8 WT_DEC_PARTITION=1
EXTRA="" run res10part WT_ENC_CU_RESERVE=10 WT_DEC_PARTITION=1
EXTRA="" run res6part WT_ENC_CU_RESERVE=6 WT_DEC_PARTITION=1
EXTRA="" run res8part_dec4 WT_ENC_CU_RESERVE=8 WT_DEC_PARTITION=1 WT_DEC_STREAMS=4
EXTRA="" run res8_cc2 WT_ENC_CU_RESERVE=8
EXTRA="--cross-chunks 2" run res8part_cc2 WT_ENC_CU_RESERVE=8 WT_DEC_PARTITION=1
EXTRA="" run res9 WT_ENC_CU_RESERVE=9
EXTRA="" run res8b WT_ENC_CU_RESERVE=8

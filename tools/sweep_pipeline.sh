set -o pipefail
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg ${EXTRA} > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/sw_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:(v['avg_launch_us'],v['ms_per_step']) for k,v in (d.get('roofline_isolated') or {}).items()}, flush=True)
except Exception as e:
    print(n,'ERR',e, open(f'gpurun_out/sw_{n}.err').read()[-300:])
PY
}
EXTRA="--arch base --batch 64 --bf16 --depth 5" run c3_res4_d5 WT_ENC_CU_RESERVE=4
EXTRA="--arch base --batch 64 --bf16 --depth 5" run c3_res8_d5 WT_ENC_CU_RESERVE=8
EXTRA="--arch base --batch 64 --bf16 --depth 10" run c3_res4_d10 WT_ENC_CU_RESERVE=4
EXTRA="--arch base --batch 64 --bf16 --depth 10" run c3_res8_d10 WT_ENC_CU_RESERVE=8
EXTRA="--arch base --batch 64 --bf16 --depth 3" run c3_res4_d3 WT_ENC_CU_RESERVE=4

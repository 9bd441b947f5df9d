set -o pipefail
root=$PWD; out=$root/gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "planes" 2>&1 | tail -2 || exit 1
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg > $out/r02_b8.json 2>/dev/null || exit 1
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b8.json")); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"]); 
for k,v in d["roofline_isolated"].items(): print(k, v["achieved"], v["avg_launch_us"], v["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/prof_lds -- python3 bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-fp32-leg > /dev/null 2> $out/pmc_lds.err || exit 2
python3 tools/pmc_summary.py $out/prof_lds > $out/r02_pmc_lds.json
rm -rf $out/prof_lds
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r02_pmc_lds.json"))
for k,v in d.items():
    if "planes" in k:
        g=lambda c: v.get(c,{}).get("per_launch",0)
        print(k[:60], "conf", g("SQ_LDS_BANK_CONFLICT"), "idx_active", g("SQ_LDS_IDX_ACTIVE"), "inst_active", g("SQ_ACTIVE_INST_LDS"), "insts", g("SQ_INSTS_LDS"))
PY

python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "decoder or dec_" 2>&1 | tail -3
python -m pytest tests/test_gpu_path.py -m gpu -q 2>&1 | tail -3
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg > gpurun_out/r02_b12.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b12.json")); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-pipeline > gpurun_out/r02_b12s.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b12s.json")); print("sync", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-pipeline > /dev/null 2>&1
python3 tools/prof_summary.py gpurun_out/prof_x | grep -E "dec_|self_att|cross_att|select" 
rm -rf gpurun_out/prof_x

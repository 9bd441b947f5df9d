set -o pipefail
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "planes or attention" 2>&1 | tail -3 || exit 1
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg > gpurun_out/r02_b9.json 2>/dev/null || exit 1
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b9.json")); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"]); 
for k,v in d["roofline_isolated"].items(): print(k, v["achieved"], v["avg_launch_us"], v["ms_per_step"])
PY

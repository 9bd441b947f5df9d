python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "cross" 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg > gpurun_out/r02_b15.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b15.json")); print("nt", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-pipeline > gpurun_out/r02_b15s.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b15s.json")); print("sync", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY

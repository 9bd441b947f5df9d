python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg > gpurun_out/r02_b16.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b16.json")); print("nt2", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
done

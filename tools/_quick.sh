python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "decoder or dec_" 2>&1 | tail -3
python -m pytest tests/test_gpu_path.py tests/test_gpu_boundary.py -m gpu -q 2>&1 | tail -3
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg > gpurun_out/r02_b13.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b13.json")); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-pipeline > gpurun_out/r02_b13s.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_b13s.json")); print("sync", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
python bench.py --arch base --batch 64 --bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c4d.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_c4d.json")); print("c4", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY

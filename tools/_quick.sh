python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "decoder or dec_" 2>&1 | tail -3
python -m pytest tests/test_gpu_path.py -m gpu -q 2>&1 | tail -3
python bench.py --arch base --batch 64 --bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c4c.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_c4c.json")); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
python bench.py --arch base --batch 64 --bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > gpurun_out/r02_c4cs.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/r02_c4cs.json")); print("sync", d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY

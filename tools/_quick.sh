set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bf16" 2>&1 | tail -15 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_path.py -m gpu -q -k "bf16_storage" 2>&1 | tail -25

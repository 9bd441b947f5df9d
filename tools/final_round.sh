# everything a round's profiles/ needs, in one GPU call (≈6 minutes): default + configs[3] profile sets, the serial
# per-kernel table, the bench lines, single-clip latency
set -o pipefail
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1 || { tail -5 gpurun_out/r03_collect.log; exit 1; }
echo "r03 profiles done"
bash tools/collect_profiles.sh r03_c4 "--arch base --batch 64 --bf16" gemm_bf16_planes > gpurun_out/r03_c4_collect.log 2>&1 || { tail -5 gpurun_out/r03_c4_collect.log; exit 2; }
echo "r03_c4 profiles done"
bash tools/serial_stats.sh r03_serial > gpurun_out/r03_serial.log 2>&1 || exit 3
echo "serial done"
bash tools/agree_check.sh r03_agree > gpurun_out/r03_agree.txt 2>&1 || exit 4
echo "agree done"
python tools/latency.py > gpurun_out/r03_latency.txt 2>&1 || exit 5
echo "latency done"
python tools/gemm_planes_bench.py > gpurun_out/r03_gemm_bursts.txt 2>&1 || exit 6
bash tools/final_bench.sh > gpurun_out/r03_final.log 2>&1 || { tail -5 gpurun_out/r03_final.log; exit 7; }
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_cmd.json 2> /dev/null || exit 8
tail -8 gpurun_out/r03_final.log

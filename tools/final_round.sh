# everything a round's profiles/ needs, in one GPU call (≈6 minutes): default + configs[3] profile sets, the serial
# per-kernel table, the bench lines, single-clip latency
set -o pipefail
R=${ROUND:-r04}
bash tools/collect_profiles.sh ${R} > gpurun_out/${R}_collect.log 2>&1 || { tail -5 gpurun_out/${R}_collect.log; exit 1; }
echo "${R} profiles done"
bash tools/collect_profiles.sh ${R}_c4 "--arch base --batch 64 --bf16" gemm_bf16_planes > gpurun_out/${R}_c4_collect.log 2>&1 || { tail -5 gpurun_out/${R}_c4_collect.log; exit 2; }
echo "${R}_c4 profiles done"
bash tools/serial_stats.sh ${R}_serial > gpurun_out/${R}_serial.log 2>&1 || exit 3
echo "serial done"
bash tools/agree_check.sh ${R}_agree > gpurun_out/${R}_agree.txt 2>&1 || exit 4
echo "agree done"
python tools/latency.py > gpurun_out/${R}_latency.txt 2>&1 || exit 5
echo "latency done"
python tools/gemm_planes_bench.py > gpurun_out/${R}_gemm_bursts.txt 2>&1 || exit 6
bash tools/final_bench.sh > gpurun_out/${R}_final.log 2>&1 || { tail -5 gpurun_out/${R}_final.log; exit 7; }
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_cmd.json 2> /dev/null || exit 8
tail -8 gpurun_out/${R}_final.log

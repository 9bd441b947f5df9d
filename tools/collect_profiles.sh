#!/bin/bash
# Round-N profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r04                                   (default workload: BASELINE configs[1])
#   bash tools/collect_profiles.sh r04_c4 "--arch base --batch 64 --bf16" gemm_bf16_planes   (configs[3])
# 1. rocprofv3 --kernel-trace --stats over the default bench command (per-kernel averages)
# 2. separate --pmc passes (never combined with other trace domains) over a serial 2-step run
# Raw output goes to gpurun_out/prof_<tag>_*; the summaries are written to gpurun_out/<tag>_*.{csv,json}
# and copied into profiles/ by hand afterwards.
set -o pipefail
tag=${1:-r04}
extra=${2:-}
dominant=${3:-gemm_planes}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
BENCH="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg $extra"
SERIAL="bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-fp32-leg $extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -- python3 $BENCH > $out/${tag}_prof_bench.json 2> $out/${tag}_prof_bench.err || exit 1
cp $out/prof_${tag}_stats/*/*_kernel_stats.csv $out/${tag}_bench_default_kernel_stats.csv
python3 tools/prof_summary.py $out/prof_${tag}_stats 39 > $out/${tag}_kernel_summary.txt
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_fetch -- python3 $SERIAL > /dev/null 2> $out/${tag}_pmc_fetch.err || exit 2
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_write -- python3 $SERIAL > /dev/null 2> $out/${tag}_pmc_write.err || exit 3
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/prof_${tag}_sq1 -- python3 $SERIAL > /dev/null 2> $out/${tag}_pmc_sq1.err || exit 4
echo "sq1 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/prof_${tag}_sq2 -- python3 $SERIAL > /dev/null 2> $out/${tag}_pmc_sq2.err || exit 5
echo "sq2 done"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/prof_${tag}_grbm -- python3 $SERIAL > /dev/null 2> $out/${tag}_pmc_grbm.err || exit 6
echo "grbm done"
python3 tools/pmc_traffic.py $dominant $out/prof_${tag}_fetch $out/prof_${tag}_write > $out/${tag}_pmc_traffic.json
python3 tools/pmc_summary.py $out/prof_${tag}_fetch $out/prof_${tag}_write $out/prof_${tag}_sq1 $out/prof_${tag}_sq2 $out/prof_${tag}_grbm > $out/${tag}_pmc_counters.json
# keep the merge-back small: the raw csv traces are only needed on the box
rm -rf $out/prof_${tag}_fetch $out/prof_${tag}_write $out/prof_${tag}_sq1 $out/prof_${tag}_sq2 $out/prof_${tag}_grbm
find $out/prof_${tag}_stats -name "*kernel_trace.csv" -size +20M -delete
echo "summaries written"

# per-kernel durations of strictly serial steps (no pipeline, nothing shares the chip): rocprofv3 --kernel-trace --stats
set -o pipefail
root=$PWD; out=$root/gpurun_out; tag=${1:-r04_serial}
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -- python3 bench.py --steps 4 --warmup 2 --no-pipeline --no-cpu-baseline --no-fp32-leg > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
python3 tools/prof_summary.py $out/prof_${tag} 39 > $out/${tag}_kernel_summary.txt
find $out/prof_${tag} -name "*kernel_trace.csv" -size +20M -delete
head -32 $out/${tag}_kernel_summary.txt

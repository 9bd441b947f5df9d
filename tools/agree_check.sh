# bench.py's per-launch figure for the dominant kernel class against rocprofv3 --kernel-trace --stats of the same command
set -o pipefail
root=$PWD; out=$root/gpurun_out; tag=${1:-r04_agree}
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
python3 - <<PY
import csv, glob, json
d = json.loads(open("$out/${tag}_bench.json").read().strip().splitlines()[-1])
print("bench: value", d["value"], "timed avg_launch_us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], "isolated", d["roofline"]["isolated"])
for cls in ("gemm_planes", "encoder_attention_planes"):
    print("  detail", cls, d["roofline_detail"][cls]["avg_launch_us"], d["roofline_isolated"][cls]["avg_launch_us"])
f = glob.glob("$out/prof_${tag}/*/*_kernel_stats.csv")[0]
tot = {}
for r in csv.DictReader(open(f)):
    for cls in ("gemm_planes", "encoder_attention_planes"):
        if cls in r["Name"]:
            t = tot.setdefault(cls, [0, 0.0]); t[0] += int(r["Calls"]); t[1] += float(r["TotalDurationNs"])
for cls, (n, ns) in tot.items():
    print("rocprofv3:", cls, n, "launches avg", round(ns / n / 1e3, 1), "us")
PY
find $out/prof_${tag} -name "*kernel_trace.csv" -size +20M -delete

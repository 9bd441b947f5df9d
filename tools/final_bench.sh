# the bench lines of a round (GPU box): default workload with all legs, configs[3] in bf16 and in the default mode
set -o pipefail
R=${ROUND:-r04}
python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err || { tail -5 gpurun_out/${R}_bench.err; exit 1; }
echo "default done"
python bench.py --arch base --batch 64 --bf16 --steps 100 > gpurun_out/${R}_c4_bench.json 2> gpurun_out/${R}_c4_bench.err || { tail -5 gpurun_out/${R}_c4_bench.err; exit 1; }
echo "c4 done"
python bench.py --arch base --batch 64 --steps 50 --no-cpu-baseline --no-fp32-leg > gpurun_out/${R}_c4_f32accurate_bench.json 2>/dev/null || exit 1
python - <<PY
import json
for f in ("${R}_bench","${R}_c4_bench","${R}_c4_f32accurate_bench"):
    d=json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("isolated"), d["roofline"]["traffic"], "dec", d["decoder_roofline"]["frac"], d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("ids_match_gpu"))
    print("   fp32 leg", d.get("encoder_fp32_mfma") and (d["encoder_fp32_mfma"]["value"], d["encoder_fp32_mfma"]["bf16x3_split"]["value"]), "frontend", d.get("with_frontend") and d["with_frontend"]["value"],
          "outlier", d.get("outlier_weights") and (d["outlier_weights"]["value"], d["outlier_weights"]["f16_fallbacks"]), "c3 leg", d.get("configs3_bf16_base") and d["configs3_bf16_base"]["value"])
PY

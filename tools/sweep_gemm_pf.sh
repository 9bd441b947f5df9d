# A/B of the plane GEMM's L2 prefetch distance (GPU box): per-shape times for WT_GEMM_PF = 0 (off), 2, 4, 8
for pf in 0 4 2 8 0 4; do
  echo "== WT_GEMM_PF=$pf"
  WT_GEMM_PF=$pf python tools/gemm_planes_bench.py | grep -v attention
done

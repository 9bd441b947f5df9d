// mfma_valu_overlap — do the matrix pipe and the vector ALU of a SIMD overlap ACROSS wavefronts when each wavefront
// alternates between an MFMA phase and a VALU phase (the shape of a flash-attention tile: QK MFMAs, softmax VALU, PV MFMAs)?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap && tools/bin/mfma_valu_overlap
// One block per CU, W wavefronts per SIMD.  Per iteration a wavefront issues NM v_mfma_f32_32x32x16_f16 (4 independent
// accumulators) and NV dependent-free v_fma_f32 (8 independent chains), in phases.  Printed: microseconds per iteration
// for MFMA only, VALU only and both, for W = 1, 2, 3: with perfect overlap both = max(mfma, valu) once W >= 2.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

template <int NM, int NV, bool STAGGER>
__global__ __launch_bounds__(768) void phases(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (threadIdx.x + 3 * e));
  }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = 0.5f + 0.001f * threadIdx.x + e;
  const float m = 0.999f, c = 0.001f;
  // STAGGER: wavefronts of one SIMD start in different phases (wave slot = (threadIdx.x / 64) / 4)
  const int slot = (threadIdx.x >> 6) >> 2;
  if (STAGGER && (slot & 1)) {
#pragma unroll
    for (int k = 0; k < NV / 8; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], m, c);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NM / 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NV / 8; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], m, c);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.0f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, bool ST>
float run(int w, int n_cu, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((phases<NM, NV, ST>), dim3(n_cu), dim3(256 * w), 0, 0, out, 50);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((phases<NM, NV, ST>), dim3(n_cu), dim3(256 * w), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)n_cu * 768 * 4);
  printf("%d CUs; per iteration and wavefront: 48 MFMA 32x32x16 f16, 240 v_fma_f32 (an attention tile's mix)\n", n_cu);
  for (int w = 1; w <= 3; ++w) {
    const float tm = run<48, 0, false>(w, n_cu, out), tv = run<0, 240, false>(w, n_cu, out), tb = run<48, 240, false>(w, n_cu, out),
                ts = run<48, 240, true>(w, n_cu, out);
    printf("W=%d waves/SIMD: mfma only %.3f us  valu only %.3f us  both %.3f us  both, staggered start %.3f us  (sum %.3f, max %.3f)\n", w,
           tm, tv, tb, ts, tm + tv, tm > tv ? tm : tv);
  }
  return 0;
}

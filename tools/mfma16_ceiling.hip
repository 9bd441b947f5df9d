// mfma16_ceiling — what the 16-bit matrix pipe of this MI355X delivers under sustained load (the practical ceiling
// the plane GEMM / attention fractions should be read against; the 2500 TF/s datasheet peak assumes 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma16_ceiling.hip -o tools/bin/mfma16_ceiling && tools/bin/mfma16_ceiling
// Every wavefront issues v_mfma_f32_32x32x16_f16 (or _bf16) back to back on NACC independent accumulators with
// operands in registers: no LDS, no memory.  Reported: TF/s for 1 and 2 wavefronts per SIMD and the core clock
// (s_memtime cycles over s_memrealtime's 100 MHz ticks) seen by wavefront 0 during the run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <bool BF, int NACC>
__global__ __launch_bounds__(512) void burn(float* out, unsigned long long* clk, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (threadIdx.x + 3 * e));
  }
  unsigned long long c0 = 0, t0 = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    c0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (BF) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
      }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = __builtin_readcyclecounter() - c0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - t0;
  }
  float s = 0.0f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BF>
void run(const char* name, int waves_per_simd, int n_cu) {
  const int threads = 256 * waves_per_simd, iters = 4000;
  constexpr int NACC = 8;
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, (size_t)n_cu * threads * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((burn<BF, NACC>), dim3(n_cu), dim3(threads), 0, 0, out, clk, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((burn<BF, NACC>), dim3(n_cu), dim3(threads), 0, 0, out, clk, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flops = 2.0 * 32 * 32 * 16 * (double)iters * 4 * NACC * (threads / 64) * n_cu;
  printf("%s  %d wave(s)/SIMD: %7.1f TF/s over %.2f ms, core clock %.3f GHz\n", name, waves_per_simd, flops / ms / 1e9, ms,
         (double)h[0] / ((double)h[1] * 10.0));
  hipFree(out);
  hipFree(clk);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
  for (int w = 1; w <= 2; ++w) {
    run<false>("f16 ", w, p.multiProcessorCount);
    run<true>("bf16", w, p.multiProcessorCount);
  }
  return 0;
}

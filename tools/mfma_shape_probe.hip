// mfma_shape_probe — under the chip's power management, which 16-bit MFMA shape delivers more for the plane GEMM's
// arithmetic: v_mfma_f32_32x32x16_f16 (3 x 3 tiles per wave) or v_mfma_f32_16x16x32_f16 (6 x 6 tiles)?  Same 96 x 96
// wave tile, same three plane products per k-step (hi.lo + lo.hi + hi.hi), operands in registers (no LDS, no memory in
// the loop), RANDOM operand data shaped like the planes (hi = fp16(x), lo = fp16(x - hi), x ~ N(0,1) * 1024) or
// constants.  MI355X_MICROARCH.md (DVFS give-back, item 7) reports 1.12-1.15 x for the 16x16x32 bf16 loop on random data.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe && tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

// operands: per lane 12 A fragments and 12 B fragments of 8 halfs (hi 0..5, lo 6..11), from `data`
template <int SHAPE, int WAVES>
__global__ __launch_bounds__(256 * WAVES) void burn(const half8* data, float* out, unsigned long long* clk, int iters) {
  half8 ah[6], al[6], bh[6], bl[6];
  const half8* d = data + (size_t)(threadIdx.x & 255) * 24;
  for (int i = 0; i < 6; ++i) ah[i] = d[i], al[i] = d[6 + i], bh[i] = d[12 + i], bl[i] = d[18 + i];
  unsigned long long c0 = 0, t0 = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    c0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memrealtime();
  }
  float s = 0.0f;
  if (SHAPE == 32) {
    f32x16 acc[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // a 32-deep k-tile = two 16-deep steps with different fragments
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i + 3 * ks], bl[j + 3 * ks], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i + 3 * ks], bh[j + 3 * ks], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i + 3 * ks], bh[j + 3 * ks], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    f32x4 acc[6][6];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = __builtin_readcyclecounter() - c0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - t0;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int WAVES>
void run(const char* what, const half8* data, int n_cu) {
  const int threads = 256 * WAVES, iters = 3000;
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, (size_t)n_cu * threads * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((burn<SHAPE, WAVES>), dim3(n_cu), dim3(threads), 0, 0, data, out, clk, 3000);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((burn<SHAPE, WAVES>), dim3(n_cu), dim3(threads), 0, 0, data, out, clk, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flops = 2.0 * 96 * 96 * 32 * 3 * (double)iters * (threads / 64) * n_cu;  // executed 16-bit FLOPs
  printf("%-28s %dx%d, %d wave(s)/SIMD: %7.1f TF/s executed (%6.1f algorithmic) over %.2f ms, core clock %.3f GHz\n", what, SHAPE, SHAPE,
         WAVES, flops / ms / 1e9, flops / ms / 3e9, ms, (double)h[0] / ((double)h[1] * 10.0));
  hipFree(out);
  hipFree(clk);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  const size_t n = 256 * 24 * 8;
  std::vector<unsigned short> planes(n), consts(n);
  // per lane: A hi x6, A lo x6, B hi x6, B lo x6 fragments of 8 halfs
  for (size_t t = 0; t < 256; ++t)
    for (int f = 0; f < 12; ++f)
      for (int e = 0; e < 8; ++e) {
        const int op = f / 6;  // 0 A, 1 B
        const float x = nd(rng) * (op == 0 ? 1024.0f : 64.0f);
        const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
        const size_t base = (t * 24 + (op * 12) + (f % 6)) * 8 + e;
        std::memcpy(&planes[base], &hi, 2);
        std::memcpy(&planes[base + 6 * 8], &lo, 2);
        const _Float16 c = (_Float16)(0.001f * (float)(t + e));
        std::memcpy(&consts[base], &c, 2);
        std::memcpy(&consts[base + 6 * 8], &c, 2);
      }
  half8 *d_planes, *d_consts;
  hipMalloc(&d_planes, n * 2);
  hipMalloc(&d_consts, n * 2);
  hipMemcpy(d_planes, planes.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_consts, consts.data(), n * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 1>("random planes", d_planes, p.multiProcessorCount);
    run<16, 1>("random planes", d_planes, p.multiProcessorCount);
    run<32, 2>("random planes", d_planes, p.multiProcessorCount);
    run<16, 2>("random planes", d_planes, p.multiProcessorCount);
    run<32, 2>("smooth constants", d_consts, p.multiProcessorCount);
    run<16, 2>("smooth constants", d_consts, p.multiProcessorCount);
  }
  return 0;
}

// mfma_probe — isolates what bounds the fp32-MFMA GEMM main loop on gfx950.
// Builds as a standalone binary: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o mfma_probe
// Each kernel issues the same number of v_mfma_f32_32x32x2_f32 per wave (4 independent 32x32
// accumulators) and adds one ingredient of the real kernel at a time:
//   0 pure MFMA, operands in registers
//   1 + LDS fragment reads (4 x ds_read_b128 per 16 MFMA), no barrier
//   2 + __syncthreads every 64 MFMA
//   3 + ds_write_b128 of a fresh tile (8 per thread) every 64 MFMA
//   4 + global loads feeding those writes (the whole staging path), L2-resident source
//   5 same, but every block streams its own fresh 16 KB per k-tile (HBM-resident source)
//   6 = 5 + the GEMM's naive epilogue (64 dword stores per lane, bias + residual read)
//   7 = 5 + LDS-transposed epilogue (16 float4 stores per lane, float4 residual reads)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ dst,
                                              int iters) {
  __shared__ __attribute__((aligned(16))) float As[128 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5, wm = wid >> 1, wn = wid & 1;
  for (int i = tid; i < 128 * 36; i += 256) {
    As[i] = src[i & 4095];
    Bs[i] = src[(i * 7) & 4095];
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  f32x4 a0 = {src[lane], src[lane + 64], src[lane + 128], src[lane + 192]};
  f32x4 a1 = a0 * 1.5f, b0 = a0 * 0.5f, b1 = a0 * 0.25f;
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  const float* gp = src + ((long)blockIdx.x * 128 + srow) * 32 + scol;
  f32x4 ra[4], rb[4];
  for (int i = 0; i < 4; ++i) ra[i] = rb[i] = a0;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long off = MODE >= 5 ? ((long)blockIdx.x * iters + it) * 8192 - (long)blockIdx.x * 4096
                                   : (long)((it & 15) * 4096);
        ra[i] = *reinterpret_cast<const f32x4*>(gp + (long)(i * 32) * 32 + off);
        rb[i] = *reinterpret_cast<const f32x4*>(gp + (long)(i * 32) * 32 + off + 2048);
      }
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      if (MODE >= 1) {
        const int kof = kq * 8 + 4 * lh;
        a0 = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + l31) * 36 + kof]);
        a1 = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + 32 + l31) * 36 + kof]);
        b0 = *reinterpret_cast<const f32x4*>(&Bs[(wn * 64 + l31) * 36 + kof]);
        b1 = *reinterpret_cast<const f32x4*>(&Bs[(wn * 64 + 32 + l31) * 36 + kof]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
      }
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(&As[(srow + 32 * i) * 36 + scol]) = ra[i];
        *reinterpret_cast<f32x4*>(&Bs[(srow + 32 * i) * 36 + scol]) = rb[i];
      }
      __syncthreads();
    }
  }
  if (MODE == 6) {
    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); N = 384
    const long m0 = (long)(blockIdx.x / 3) * 128, n0 = (blockIdx.x % 3) * 128;
    for (int mi = 0; mi < 2; ++mi)
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        for (int ni = 0; ni < 2; ++ni) {
          const long n = n0 + wn * 64 + ni * 32 + l31;
          float v = acc[mi][ni][r] + src[n];
          v += dst[m * 384 + n];
          dst[m * 384 + n] = v;
        }
      }
    return;
  }
  if (MODE == 7) {
    const long m0 = (long)(blockIdx.x / 3) * 128, n0 = (blockIdx.x % 3) * 128;
    __syncthreads();  // all waves are done with As/Bs
    float* stage = As + wid * (32 * 68);  // per-wave [32 rows][64 cols + 4]; As+Bs = 9216 floats >= 4*2176
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(src + n0 + wn * 64 + c4);
    for (int mi = 0; mi < 2; ++mi) {
      for (int ni = 0; ni < 2; ++ni)
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + ni * 32 + l31] = acc[mi][ni][r];
      // wave-private staging: no block barrier needed, only LDS ordering inside the wave
      __builtin_amdgcn_s_waitcnt(0xC07F);
      for (int i = 0; i < 8; ++i) {
        const int row = i * 4 + rr;
        f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * 68 + c4]);
        const long o = (m0 + wm * 64 + mi * 32 + row) * 384 + n0 + wn * 64 + c4;
        const f32x4 res = *reinterpret_cast<const f32x4*>(dst + o);
        v += bias + res;
        *reinterpret_cast<f32x4*>(dst + o) = v;
      }
    }
    return;
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  dst[(long)blockIdx.x * 256 + tid] = s;
}

static size_t g_src_floats = 0, g_dst_floats = 0;

template <int MODE>
void run(const float* src, float* dst, int blocks, int iters) {
  // host-side bounds check of everything the kernel will index (a first version of this probe
  // faulted the GPU by streaming past the source buffer)
  if (MODE >= 5 && (size_t)blocks * iters * 8192 + 2 * 8192 > g_src_floats) {
    printf("mode %d blocks %d iters %d: skipped (source buffer too small)\n", MODE, blocks, iters);
    return;
  }
  if (MODE >= 6 && ((size_t)(blocks / 3 + 1) * 128 * 384 > g_dst_floats || blocks % 3 != 0)) {
    printf("mode %d blocks %d: skipped (output buffer too small / blocks not a multiple of 3)\n", MODE, blocks);
    return;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, src, dst, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 5.0 * blocks * 4.0 * iters * 64.0 * 4096.0;
  printf("mode %d blocks %5d (%.1f per CU) iters %4d : %7.3f ms  %6.1f TFLOP/s\n", MODE, blocks, blocks / 256.0,
         iters, ms / 5, flops / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  float *src, *dst;
  const size_t n = 352u << 20;  // 1.4 GB: mode 5 streams 32 KB per block per k-tile
  hipMalloc(&src, n * 4);
  hipMalloc(&dst, 256u << 20);
  g_src_floats = n;
  g_dst_floats = (256u << 20) / 4;  // modes 6/7 write a [blocks/3*128][384] fp32 output
  std::vector<float> h(n);
  unsigned x = 12345;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = float(int(x >> 9) - (1 << 22)) * (1.0f / (1 << 22));
  }
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  if (getenv("PROBE_FULL")) {
    for (int blocks : {256, 512, 768, 1024, 3072}) {
      run<0>(src, dst, blocks, 48);
      run<1>(src, dst, blocks, 48);
      run<2>(src, dst, blocks, 48);
      run<3>(src, dst, blocks, 48);
      run<4>(src, dst, blocks, 48);
      if (blocks <= 1024) run<5>(src, dst, blocks, 32);
    }
  }
  for (int blocks : {768, 1125, 2250, 3375}) {
    run<5>(src, dst, blocks, 12);
    run<6>(src, dst, blocks, 12);
    run<7>(src, dst, blocks, 12);
  }
  run<5>(src, dst, 768, 48);
  run<7>(src, dst, 768, 48);
  return 0;
}

// cu_stream_probe — how many GB/s can ONE compute unit stream from HBM, by method and by bytes in flight?  The pipeline's
// step is a sum of CU x time: a streaming decoder kernel (the absorbed cross-attention reads 8 GB of encoder
// planes per batch) costs bytes / (per-CU rate) whatever its grid is.
//   hipcc --offload-arch=gfx950 -O3 tools/cu_stream_probe.hip -o tools/bin/cu_stream_probe && tools/bin/cu_stream_probe
// N blocks (one per CU: N <= 256), each streams its own contiguous 16 MB slice of a 4 GB buffer once per launch:
//   regs W x U   W wavefronts, U global_load_dwordx4 in flight per lane (1 KiB contiguous per wave-instruction), summed
//   stores W x U  W wavefronts, U global_store_dwordx4 per lane and loop step, the same slices written
//   dma L: S x KB  L loader wavefronts (a wavefront holds at most 63 memory instructions in flight: vmcnt is 6 bits),
//                LDS-DMA (global_load_lds_dwordx4, 1 KiB contiguous per instruction) into a ring of S stages of KB KiB;
//                four consumer wavefronts read every stage from LDS (ds_read_b128) and sum
// Printed: GB/s per CU and TB/s chip-wide for N = 32, 64, 128, 256.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr size_t kSlice = 16u << 20;

template <int W, int U>
__global__ __launch_bounds__(64 * W) void stream_regs(const u32x4* __restrict__ src, unsigned* out) {
  const u32x4* p = src + (size_t)blockIdx.x * (kSlice / 16) + threadIdx.x;
  constexpr int kStep = 64 * W;  // 16-byte units per block-wide load
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = 0; i < kSlice / 16 / kStep; i += U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (i + u) * kStep);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

// stores: W wavefronts, U dwordx4 stores per lane and loop step (1 KiB contiguous per wave-instruction)
template <int W, int U>
__global__ __launch_bounds__(64 * W) void stream_stores(u32x4* __restrict__ dst, unsigned seed) {
  u32x4* p = dst + (size_t)blockIdx.x * (kSlice / 16) + threadIdx.x;
  constexpr int kStep = 64 * W;
  const u32x4 v = {seed, seed + threadIdx.x, seed ^ blockIdx.x, 7u};
  for (size_t i = 0; i < kSlice / 16 / kStep; i += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) p[(i + u) * kStep] = v;
  }
}

__device__ __forceinline__ void lds_dma16_sgpr(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// loader waves 4 .. 3 + L keep S - 1 stages in flight; consumers (waves 0-3) read stage t while t + 1 .. t + S - 1 load
template <int S, int KB, int L>
__global__ __launch_bounds__(256 + 64 * L) void stream_dma(const unsigned char* __restrict__ src, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const unsigned long long base = reinterpret_cast<unsigned long long>(src) + (unsigned long long)blockIdx.x * kSlice;
  constexpr int kStage = KB * 1024, kTiles = kSlice / kStage, PER = KB / L;
  static_assert(KB % L == 0 && (S - 2) * PER <= 63, "vmcnt");
  auto issue = [&](int t) {
    const unsigned long long sb = base + (unsigned long long)t * kStage;
    const unsigned dst = lds_base + (unsigned)((t % S) * kStage);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int q = (wid - 4) * PER + j;
      lds_dma16_sgpr((unsigned)(q * 1024 + lane * 16), sb, dst + q * 1024);
    }
  };
  u32x4 acc = {0, 0, 0, 0};
  if (wid >= 4) {
#pragma unroll
    for (int t = 0; t < S - 1; ++t) issue(t);
  }
  for (int t = 0; t < kTiles; ++t) {
    if (wid >= 4) {
      if (t + S - 1 <= kTiles) wait_vmcnt<(S - 2) * PER>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // stage t has landed; everybody is done with stage t - 1
    if (wid >= 4) {
      if (t + S - 1 < kTiles) issue(t + S - 1);
    } else {
      const u32x4* st = reinterpret_cast<const u32x4*>(smem + (t % S) * kStage);
#pragma unroll
      for (int i = 0; i < kStage / 16 / 256; ++i) acc += st[i * 256 + tid];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

template <class F>
static void run(const char* name, F launch) {
  const int ns[] = {32, 64, 128, 256};
  printf("%-14s", name);
  for (int n : ns) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(n);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int it = 0; it < 5; ++it) launch(n);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)kSlice * 5 / (ms * 1e-3) / 1e9;
    printf("  N=%3d: %6.1f GB/s per CU %5.2f TB/s", n, gbs, gbs * n / 1e3);
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  unsigned char* buf;
  unsigned* out;
  const size_t bytes = 256 * kSlice;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, bytes);
  const u32x4* b4 = reinterpret_cast<const u32x4*>(buf);
  run("regs 4 x 4", [&](int n) { stream_regs<4, 4><<<n, 256>>>(b4, out); });
  run("regs 4 x 8", [&](int n) { stream_regs<4, 8><<<n, 256>>>(b4, out); });
  run("regs 8 x 8", [&](int n) { stream_regs<8, 8><<<n, 512>>>(b4, out); });
  run("regs 16 x 4", [&](int n) { stream_regs<16, 4><<<n, 1024>>>(b4, out); });
  run("regs 16 x 8", [&](int n) { stream_regs<16, 8><<<n, 1024>>>(b4, out); });
  run("regs 16 x 16", [&](int n) { stream_regs<16, 16><<<n, 1024>>>(b4, out); });
  u32x4* w4 = reinterpret_cast<u32x4*>(buf);
  run("stores 4 x 4", [&](int n) { stream_stores<4, 4><<<n, 256>>>(w4, 1u); });
  run("stores 8 x 8", [&](int n) { stream_stores<8, 8><<<n, 512>>>(w4, 2u); });
  run("stores 16 x 8", [&](int n) { stream_stores<16, 8><<<n, 1024>>>(w4, 3u); });
#define DMA(S, KB, L)                                                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_dma<S, KB, L>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  run("dma " #L ": " #S " x " #KB, [&](int n) { stream_dma<S, KB, L><<<n, 256 + 64 * L, S * KB * 1024>>>(buf, out); });
  DMA(3, 16, 1)
  DMA(3, 32, 1)
  DMA(5, 16, 1)
  DMA(5, 32, 2)
  DMA(9, 16, 2)
  DMA(5, 32, 4)
  return 0;
}

// dma_probe — what does the plane GEMM's LDS-DMA staging cost by the SHAPE of the global segments an instruction
// fetches?  (gfx950; hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/bin/dma_probe && tools/bin/dma_probe)
// One 512-thread block per row tile of 192 rows (250 blocks, fc2's shape: K = 1536, 48 k-tiles of 32), the staging of
// gemm_planes_tile and nothing else: per k-tile 72 global_load_lds_dwordx4 (1 KiB each: 24 for the block's own A rows,
// 48 for the W tile every block shares) into a two-stage ring, vmcnt(0), barrier.  Variants:
//   seg 64   an instruction = 16 rows x 64 B of ONE plane (the layout in HBM today: hi plane, lo plane, row-major)
//   seg 128  an instruction = 8 rows x 128 B: hi and lo of a row's 32-deep k-tile next to each other (whole cache lines)
//   seg 1024 an instruction = 1 KiB contiguous (k-tile-blocked operand)
// Printed: microseconds per launch, GB/s per CU, and wave 0's average cycles per k-tile spent issuing / waiting.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void lds_dma16_sgpr(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

constexpr int K = 1536, BK = 32, NKT = K / BK, BM = 192, BN = 384, M = 48000;
constexpr int QPW = 9, NW = 8, STAGE = 72 * 1024;

template <int SEG, int ISSUERS>
__global__ __launch_bounds__(512) void stage_only(const unsigned char* A, const unsigned char* W, long long* stats, int nkt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const long m0 = (long)blockIdx.x * BM;
  constexpr int PER = 72 / ISSUERS;  // instructions per issuing wave
  unsigned voff[PER];
  const unsigned char* ubase[PER];
  unsigned kstep;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int q = wid + ISSUERS * j;  // 0..23 A, 24..71 W
    const bool is_a = q < 24;
    const int qq = is_a ? q : q - 24;
    long off;
    if (SEG == 64) {  // qq: A 0..11 hi, 12..23 lo; W 0..23 hi, 24..47 lo; 16 rows per instruction
      const int half = is_a ? 12 : 24;
      const bool lo = qq >= half;
      const long row = (is_a ? m0 : 0) + 16 * (qq % half) + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const long plane = is_a ? (long)M * K * 2 : (long)BN * K * 2;
      off = (lo ? plane : 0) + row * (K * 2) + chunk * 16;
      kstep = 64;
    } else if (SEG == 128) {  // 8 rows per instruction, both planes of a row's k-tile in one 128-byte line
      const long row = (is_a ? m0 : 0) + 8 * qq + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      off = row * (K * 4) + chunk * 16;
      kstep = 128;
    } else {  // 1 KiB blocks: [row group of 8][k-tile][1024]
      const long grp = (is_a ? m0 / 8 : 0) + qq;
      off = grp * (NKT * 1024) + lane * 16;
      kstep = 1024;
    }
    voff[j] = (unsigned)off;
    ubase[j] = is_a ? A : W;
  }
  long long t_issue = 0, t_wait = 0;
  auto issue = [&](int kt, int buf) {
    if (wid < ISSUERS) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const unsigned long long sb = reinterpret_cast<unsigned long long>(ubase[j]) + (size_t)kt * kstep;
        lds_dma16_sgpr(voff[j], sb, lds_base + (unsigned)(buf * STAGE + (wid + ISSUERS * j) * 1024));
      }
    }
  };
  issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long t1 = __builtin_readcyclecounter();
    if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
    const long long t2 = __builtin_readcyclecounter();
    t_wait += t1 - t0;
    t_issue += t2 - t1;
  }
  if (tid == 0) {
    stats[2 * blockIdx.x] = t_issue;
    stats[2 * blockIdx.x + 1] = t_wait;
  }
}

template <int SEG, int ISSUERS>
void run(const unsigned char* A, const unsigned char* W, long long* stats, const char* what) {
  const int blocks = M / BM;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_only<SEG, ISSUERS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stage_only<SEG, ISSUERS>), dim3(blocks), dim3(512), 2 * STAGE, 0, A, W, stats, NKT);
  hipEventRecord(e0, 0);
  const int iters = 10;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((stage_only<SEG, ISSUERS>), dim3(blocks), dim3(512), 2 * STAGE, 0, A, W, stats, NKT);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = 1e3 * ms / iters;
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), stats, h.size() * 8, hipMemcpyDeviceToHost);
  double ti = 0, tw = 0;
  for (int b = 0; b < blocks; ++b) ti += h[2 * b], tw += h[2 * b + 1];
  const double bytes = (double)blocks * NKT * 72 * 1024;
  printf("%-44s %7.1f us  %6.1f GB/s per CU (%5.2f TB/s)  wave0 per k-tile: issue %6.0f wait %6.0f cycles (memtime ticks)\n", what, us,
         bytes / blocks / us * 1e-3, bytes / us * 1e-6, ti / blocks / NKT, tw / blocks / NKT);
}

int main() {
  const size_t a_bytes = (size_t)M * K * 4 + (1 << 20), w_bytes = (size_t)BN * K * 4 + (1 << 20);
  unsigned char *A, *W;
  long long* stats;
  hipMalloc(&A, a_bytes);
  hipMalloc(&W, w_bytes);
  hipMalloc(&stats, 4096 * 8);
  hipMemset(A, 1, a_bytes);
  hipMemset(W, 1, w_bytes);
  for (int rep = 0; rep < 2; ++rep) {
    run<64, 8>(A, W, stats, "seg 64 (16 rows x 64 B), 8 waves issue");
    run<128, 8>(A, W, stats, "seg 128 (8 rows x 128 B), 8 waves issue");
    run<1024, 8>(A, W, stats, "seg 1024 (contiguous), 8 waves issue");
    run<64, 4>(A, W, stats, "seg 64, 4 waves issue 18 each");
    run<128, 4>(A, W, stats, "seg 128, 4 waves issue 18 each");
  }
  return 0;
}

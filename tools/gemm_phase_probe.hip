// gemm_phase_probe — where do the cycles of the ping-pong plane GEMM's main loop go?  Diagnostic build of the product
// kernel (k_gemm_planes.hip compiled with WT_PP_STAMPS: s_memtime sums per phase in scalar registers), launched on
// synthetic operands without the engine:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -Iinclude -Iwhisper.tflite_amd/csrc tools/gemm_phase_probe.hip -o tools/bin/gemm_phase_probe
// Prints, per shape and wave group, the average cycles per k-tile spent in: C(s0) issue, barrier wait, L(s1) reads,
// wait, C(s1), wait, L(s0) + DMA issue, wait — and the loop total (stamps cost ~10 %: compare shares, not wall time).
#define WT_PP_STAMPS 1
#include "../whisper.tflite_amd/csrc/k_gemm_planes.hip"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

namespace wt {
thread_local LaunchTimer g_launch_timer;
}

int main() {
  struct Shape { const char* name; int M, N, K, epi; bool planes; };
  const Shape shapes[] = {{"fc2", 48000, 384, 1536, 5, false}, {"qkv", 48000, 1152, 384, 1, true}, {"fc1", 48000, 1536, 384, 3, true}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  for (int mode = 1; mode <= 2; ++mode)
  for (const Shape& sh : shapes) {
    wt::set_plane_gemm_mode(mode);
    const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
    std::vector<unsigned short> hA(2 * na + 256);
    for (size_t i = 0; i < na; ++i) {
      const float v = nd(rng) * 1024.0f;
      const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
      std::memcpy(&hA[i], &h, 2);
      std::memcpy(&hA[na + 128 + i], &l, 2);
    }
    std::vector<float> hW(nw);
    for (auto& v : hW) v = nd(rng) * 0.05f;
    const std::vector<unsigned short> hWp = wt::split_weight_planes(hW.data(), sh.N, sh.K, sh.K, 4096.0f);
    unsigned short *dA, *dW, *dP;
    float *dC, *dB;
    hipMalloc(&dA, hA.size() * 2);
    hipMalloc(&dW, hWp.size() * 2 + 256);
    hipMalloc(&dP, (2 * nc + 256) * 2);
    hipMalloc(&dC, nc * 4);
    hipMalloc(&dB, sh.N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, nc * 4);
    hipMemset(dB, 0, sh.N * 4);
    wt::PlaneGemmArgs g;
    g.A = dA; g.a_plane = (long)na + 128; g.lda = sh.K; g.W = dW; g.bias = dB; g.C = dC; g.R = dC; g.ldc = sh.N;
    g.M = sh.M; g.N = sh.N; g.K = sh.K; g.a_scale = 1024.0f; g.w_scale = 4096.0f;
    if (sh.planes) { g.P = dP; g.p_plane = (long)nc + 128; g.out_scale[0] = 64.0f; }
    for (int it = 0; it < 5; ++it) wt::launch_gemm_planes(g, sh.epi, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 10; ++it) wt::launch_gemm_planes(g, sh.epi, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(1024 * 8 * 9);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(wt::g_pp_stamps), st.size() * 8);
    const int nkt = sh.K / 32, blocks = std::min(1024, ((sh.M + 191) / 192) * (sh.N / 384));
    printf("mode %d %s %dx%dx%d: %.1f us per launch (with stamps)\n", mode, sh.name, sh.M, sh.N, sh.K, 1e3 * ms / 10);
    for (int grp = 0; grp < 2; ++grp) {
      double a[9] = {0};
      for (int b = 0; b < blocks; ++b)
        for (int w = 4 * grp; w < 4 * grp + 4; ++w)
          for (int i = 0; i < 9; ++i) a[i] += (double)st[(b * 8 + w) * 9 + i];
      for (double& v : a) v /= 4.0 * blocks;
      const double ghz = a[7] / a[8] * 0.1;  // s_memrealtime ticks at 100 MHz
      printf("  G%d phases per k-tile: %5.0f %5.0f %5.0f %5.0f = %6.0f cycles (ideal 3456) | per tile: prologue %6.0f loop %7.0f epilogue %6.0f "
             "kernel %7.0f cycles = %6.1f us at %.2f GHz\n",
             grp, a[0] / nkt, a[1] / nkt, a[2] / nkt, a[3] / nkt, a[4] / nkt, a[5], a[4], a[6], a[7], a[7] / ghz * 1e-3, ghz);
    }
    hipFree(dA); hipFree(dW); hipFree(dP); hipFree(dC); hipFree(dB);
  }
  return 0;
}

// gemm_bf16_probe — where does a tile of the bf16-storage encoder GEMM spend its life?  Diagnostic build of the product
// kernel (k_gemm_bf16.hip with WT_BF16_STAMPS: s_memtime at kernel start / first stage landed / loop end / epilogue end
// of wave 0), on synthetic operands without the engine; -DWT_BF16_ABL=1..3 additionally ablates the LDS-DMA, the MFMAs
// or the epilogue (results are then garbage by construction; only the times matter):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -Iinclude -Iwhisper.tflite_amd/csrc [-DWT_BF16_ABL=n] tools/gemm_bf16_probe.hip -o tools/bin/gemm_bf16_probe[n]
#define WT_BF16_STAMPS 1
#include "../whisper.tflite_amd/csrc/k_gemm_bf16.hip"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

namespace wt {
thread_local LaunchTimer g_launch_timer;
}

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 64;
  struct Shape { const char* name; int M, N, K, epi; bool bf; };
  const int M = batch * 1500;
  const Shape shapes[] = {{"qkv", M, 1536, 512, 1, true}, {"out", M, 512, 512, 5, false}, {"fc1", M, 2048, 512, 3, true},
                          {"fc2", M, 512, 2048, 5, false}, {"cross-kv", M, 6144, 512, 1, true}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  for (const Shape& sh : shapes) {
    const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
    std::vector<unsigned short> hA(na + 256), hW(nw + 256);
    auto bf = [](float v) { unsigned u; std::memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); };
    for (size_t i = 0; i < na; ++i) hA[i] = bf(nd(rng));
    for (size_t i = 0; i < nw; ++i) hW[i] = bf(nd(rng) * 0.05f);
    unsigned short *dA, *dW, *dP;
    float *dC, *dB;
    hipMalloc(&dA, hA.size() * 2);
    hipMalloc(&dW, hW.size() * 2);
    hipMalloc(&dP, (nc + 256) * 2);
    hipMalloc(&dC, nc * 4);
    hipMalloc(&dB, sh.N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, nc * 4);
    hipMemset(dB, 0, sh.N * 4);
    wt::PlaneGemmArgs g;
    g.A = dA; g.lda = sh.K; g.W = dW; g.bias = dB; g.C = dC; g.R = dC; g.ldc = sh.N;
    g.M = sh.M; g.N = sh.N; g.K = sh.K;
    if (sh.bf) g.P = dP;
    for (int it = 0; it < 3; ++it) wt::launch_gemm_bf16_planes(g, sh.epi, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 10; ++it) wt::launch_gemm_bf16_planes(g, sh.epi, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(4096 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(wt::g_bf16_stamps), st.size() * 8);
    const int bn = sh.N % 384 == 0 ? 384 : 256, bm = 192;
    const int blocks = ((sh.M + bm - 1) / bm) * (sh.N / bn), nb = std::min(4096, blocks);
    double fill = 0, loop = 0, epi = 0, tot = 0, rt = 0;
    for (int b = 0; b < nb; ++b) {
      const long long* s = &st[b * 8];
      fill += s[1] - s[0]; loop += s[2] - s[1]; epi += s[3] - s[2]; tot += s[3] - s[0]; rt += s[5] - s[4];
    }
    fill /= nb; loop /= nb; epi /= nb; tot /= nb; rt /= nb;
    const double ghz = tot / rt * 0.1;
    // MFMA issue per wavefront: its share of the tile's 32 x 32 x 16 products, 8 passes of 4 cycles each (two wavefronts
    // share a SIMD); WT_BF16_RING=0 runs round 3's two-stage kernel
    const double mfma_cycles = (double)(sh.K / 16) * (bm / 32) * (bn / 32) / 8 * 32;
    printf("ABL %d %-8s %dx%dx%d tile %dx%d: %7.1f us per launch = %6.1f TF/s | per tile (%d blocks, %.2f per CU): fill %6.0f loop %7.0f "
           "(own MFMA issue %6.0f) epilogue %6.0f total %7.0f cycles = %5.1f us at %.2f GHz\n",
           WT_BF16_ABL, sh.name, sh.M, sh.N, sh.K, bm, bn, 1e3 * ms / 10, 2.0 * sh.M * sh.N * sh.K / (ms / 10) / 1e9, blocks, blocks / 256.0,
           fill, loop, mfma_cycles, epi, tot, tot / ghz * 1e-3, ghz);
    hipFree(dA); hipFree(dW); hipFree(dP); hipFree(dC); hipFree(dB);
  }
  return 0;
}

// gemm_two_streams_probe — would TWO encoder streams on half the CUs each beat one stream on all of them?  A round of the
// plane GEMM ends in a chip-wide store burst (every CU stores its tile at the same moment: HBM-write-bound);
// two independent streams drift apart, and one's stores fall into the other's loops.  The product kernels on synthetic
// operands, without the engine:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -Iinclude -Iwhisper.tflite_amd/csrc tools/gemm_two_streams_probe.hip -o tools/bin/gemm_two_streams_probe
// For qkv / fc1 / fc2 (whisper-tiny, 32 clips): 2 x 20 launches on ONE stream with all 256 CUs against 20 launches on each of
// TWO CU-masked streams (128 CUs each, every XCD split in half) running concurrently; wall time per launch pair.
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
  const Shape shapes[] = {{"qkv", 48000, 1152, 384, 1, true}, {"fc1", 48000, 1536, 384, 3, true}, {"fc2", 48000, 384, 1536, 5, false}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  hipStream_t full, half[2];
  (void)hipStreamCreateWithFlags(&full, hipStreamNonBlocking);
  for (int h = 0; h < 2; ++h) {
    std::vector<uint32_t> mask(8, 0u);
    for (int i = 0; i < 256; ++i)
      if (((i / 8) < 16) == (h == 0)) mask[i / 32] |= 1u << (i % 32);  // bit i = CU i / 8 of XCD i % 8: CUs 0-15 / 16-31 of every XCD
    if (hipExtStreamCreateWithCUMask(&half[h], 8, mask.data()) != hipSuccess) return 1;
  }
  for (const Shape& sh : shapes) {
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
    unsigned short *dA[2], *dW, *dP[2];
    float *dC[2], *dB;
    (void)hipMalloc(&dW, hWp.size() * 2 + 256);
    (void)hipMalloc(&dB, sh.N * 4);
    (void)hipMemcpy(dW, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dB, 0, sh.N * 4);
    wt::PlaneGemmArgs g[2];
    for (int h = 0; h < 2; ++h) {
      (void)hipMalloc(&dA[h], hA.size() * 2);
      (void)hipMalloc(&dP[h], (2 * nc + 256) * 2);
      (void)hipMalloc(&dC[h], nc * 4);
      (void)hipMemcpy(dA[h], hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
      (void)hipMemset(dC[h], 0, nc * 4);
      g[h].A = dA[h]; g[h].a_plane = (long)na + 128; g[h].lda = sh.K; g[h].W = dW; g[h].bias = dB; g[h].C = dC[h]; g[h].R = dC[h]; g[h].ldc = sh.N;
      g[h].M = sh.M; g[h].N = sh.N; g[h].K = sh.K; g[h].a_scale = 1024.0f; g[h].w_scale = 4096.0f;
      if (sh.planes) { g[h].P = dP[h]; g[h].p_plane = (long)nc + 128; g[h].out_scale[0] = 64.0f; }
    }
    auto wall = [&](bool two) {
      for (int rep = 0; rep < 2; ++rep) {  // the first repetition warms up
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1, eh;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&eh);
        (void)hipEventRecord(e0, two ? half[0] : full);
        if (two) {
          (void)hipStreamWaitEvent(half[1], e0, 0);
          for (int it = 0; it < 20; ++it)
            for (int h = 0; h < 2; ++h) { g[h].n_cu = 128; wt::launch_gemm_planes(g[h], sh.epi, half[h]); }
          (void)hipEventRecord(eh, half[1]);
          (void)hipStreamWaitEvent(half[0], eh, 0);
          (void)hipEventRecord(e1, half[0]);
        } else {
          for (int it = 0; it < 20; ++it)
            for (int h = 0; h < 2; ++h) { g[h].n_cu = 256; wt::launch_gemm_planes(g[h], sh.epi, full); }
          (void)hipEventRecord(e1, full);
        }
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) return 1e3 * ms / 20;
      }
      return 0.0;
    };
    for (int round = 0; round < 3; ++round) {
      const double one = wall(false), two = wall(true);
      printf("%-4s %dx%dx%d: one stream, 256 CUs %7.1f us per pair of launches | two streams, 128 CUs each %7.1f us (%+.1f %%)\n", sh.name, sh.M,
             sh.N, sh.K, one, two, 100.0 * (one / two - 1.0));
      fflush(stdout);
    }
    for (int h = 0; h < 2; ++h) { (void)hipFree(dA[h]); (void)hipFree(dP[h]); (void)hipFree(dC[h]); }
    (void)hipFree(dW); (void)hipFree(dB);
  }
  return 0;
}

// cross_abs_phase_probe — where a block of the absorbed cross-attention spends its cycles (diagnostic build of
// k_cross_absorbed.hip with WT_ABS_STAMPS), tiny's shape: 32 clips x 1500 keys x 384, 6 heads, one position.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -std=c++17 -Iinclude -Iwhisper.tflite_amd/csrc tools/cross_abs_phase_probe.hip -o tools/bin/cross_abs_phase_probe
#define WT_ABS_STAMPS 1
#include "../whisper.tflite_amd/csrc/k_cross_absorbed.hip"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

int main() {
  const int B = 32, H = 6, T = 1500, DM = 384;
  std::mt19937 rng(3);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  const size_t ne = (size_t)B * T * DM;
  std::vector<unsigned short> hE(2 * ne + 256);
  for (size_t i = 0; i < ne; ++i) {
    const float v = nd(rng) * 2048.0f;
    const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
    std::memcpy(&hE[i], &h, 2);
    std::memcpy(&hE[ne + 128 + i], &l, 2);
  }
  std::vector<float> hq((size_t)B * H * DM);
  for (auto& v : hq) v = nd(rng) * 0.15f;
  unsigned short* dE;
  float *dq, *dws;
  hipMalloc(&dE, hE.size() * 2);
  hipMalloc(&dq, hq.size() * 4);
  hipMalloc(&dws, (size_t)B * H * 16 * (DM + 4) * 4);
  hipMemcpy(dE, hE.data(), hE.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
  for (int chunks : {1, 4, 8}) {
    wt::CrossAbsorbedArgs a;
    a.qp = dq; a.e = dE; a.e_plane = (long)ne + 128; a.e_scale = 2048.0f; a.ws = dws;
    a.batch = B; a.heads = H; a.d_model = DM; a.T = T; a.chunks = chunks; a.nq = 1; a.p0 = 0;
    for (int it = 0; it < 5; ++it) wt::launch_cross_absorbed(a, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) wt::launch_cross_absorbed(a, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(4096 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(wt::g_abs_stamps), st.size() * 8);
    const int blocks = B * chunks;
    double s[16] = {0};
    for (int b = 0; b < blocks; ++b)
      for (int i = 0; i < 16; ++i) s[i] += (double)st[b * 16 + i];
    for (double& v : s) v /= blocks;
    const double ghz = s[4] / s[5] * 0.1;
    printf("chunks %d (%d blocks x %.0f tiles): %.1f us per launch | block: queries %5.0f, first tile wait %5.0f, tiles %6.0f (%.0f per tile), "
           "records %5.0f, total %6.0f cycles = %.1f us at %.2f GHz\n",
           chunks, blocks, s[6], 1e3 * ms / 20, s[0], s[1], s[2], s[2] / s[6], s[3], s[4], s[4] / ghz * 1e-3, ghz);
    printf("    per tile: wait+barrier %4.0f | DMA issue %4.0f | scores %4.0f | exchange+barrier %4.0f | softmax %4.0f | context %4.0f cycles\n",
           s[8] / s[6], s[9] / s[6], s[10] / s[6], s[11] / s[6], s[12] / s[6], s[13] / s[6]);
  }
  return 0;
}

// fp32 GEMM kernels for gfx950 on v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered
// fmaf chain, 64 FLOP/clk/SIMD).  These replace the arithmetic the reference runs
// inside tflite::Interpreter::Invoke() (whisper.tflite/whisper.cpp:295, :375) for
// every Conv1D / Linear of the encoder and decoder graphs.
//
// gemm_f32_128x128: 256 threads = 4 wavefronts (2x2), each wavefront owns a 64x64
// output sub-tile as 2x2 MFMA tiles (64 accumulator VGPRs).  A and W tiles are staged
// global -> registers -> LDS (issue-early / write-late), LDS rows padded to 36 floats
// so the ds_read_b128 fragment reads are bank-conflict-free.  blockIdx is remapped so
// that the tiles of one XCD (blockIdx % 8) are consecutive in (m, n) order and share
// their A panel through that XCD's L2.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 128, BN = 128;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

struct GemmDev {
  const float* A;
  const float* W;
  float* C;
  const float* bias;
  const float* R;
  const float* pos;
  int M, N, K;
  int a_rpb;
  long a_bs;
  int lda;
  int c_rpb;
  long c_bs;
  int ldc;
  int pos_period;
  int kv_batch, kv_heads, kv_dmodel;
};

// BK = 64 (69.6 KB of LDS, 2 blocks per CU) halves the barriers per k and leaves half of each
// SIMD's registers and 90 KB of LDS free, so the decoder's small kernels of the previous batch
// can co-reside with the encoder (two-stream pipeline); BK = 32 serves K % 64 != 0.
template <int EPI, int BK>
__global__ __launch_bounds__(256) void gemm_f32_128x128(GemmDev g) {
  constexpr int LDS_LD = BK + 4;  // odd multiple of 16 B: conflict-free ds_read_b128
  constexpr int NLD = BK / 8;     // float4 per thread per operand per k-tile
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

  // XCD-aware bijective remap: blocks with equal blockIdx % 8 share an XCD (speed only).
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_tiles = g.N / BN;
  const int m0 = (logical / n_tiles) * BM;
  const int n0 = (logical % n_tiles) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // staging map: 128 rows x BK/4 float4 per operand tile, NLD float4 per thread per operand;
  // BK/4 consecutive threads cover one row's k-tile (128 or 256 contiguous bytes)
  constexpr int TPR = BK / 4;       // threads per row
  constexpr int RPP = 256 / TPR;    // rows per pass
  const int srow = tid / TPR, scol = (tid % TPR) * 4;
  const float* a_ptr[NLD];
  const float* w_ptr[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int m = m0 + srow + RPP * i;
    m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
    a_ptr[i] = g.A + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + scol;
    w_ptr[i] = g.W + (long)(n0 + srow + RPP * i) * g.K + scol;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 ra[NLD], rb[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i]);
    rb[i] = *reinterpret_cast<const f32x4*>(w_ptr[i]);
  }

  const int nkt = g.K / BK;
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<f32x4*>(&As[(srow + RPP * i) * LDS_LD + scol]) = ra[i];
      *reinterpret_cast<f32x4*>(&Bs[(srow + RPP * i) * LDS_LD + scol]) = rb[i];
    }
    __syncthreads();
    if (kt + 1 < nkt) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (kt + 1) * BK);
        rb[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + (kt + 1) * BK);
      }
    }
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      // lane (row l31, half lh) takes k = 8*kq + 4*lh + j for MFMA step j: A and B use
      // the same k permutation, so the contraction is complete and exact.
      const int kof = kq * 8 + 4 * lh;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + l31) * LDS_LD + kof]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + 32 + l31) * LDS_LD + kof]);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bs[(wn * 64 + l31) * LDS_LD + kof]);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bs[(wn * 64 + 32 + l31) * LDS_LD + kof]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= g.M) continue;
      const int mb = m / g.c_rpb, mt = m % g.c_rpb;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + l31;
        float v = acc[mi][ni][r];
        if (EPI & kEpiBias) v += g.bias[n];
        if (EPI & kEpiGelu) v = gelu_erf(v);
        if (EPI & kEpiPos) v += g.pos[(long)(m % g.pos_period) * g.N + n];
        if (EPI & kEpiKvLayout) {
          const int slab = n / g.kv_dmodel, rem = n % g.kv_dmodel;
          const int head = rem >> 6, dd = rem & 63;
          const long o = (((long)slab * g.kv_batch + mb) * g.kv_heads + head) * (long)g.c_rpb * 64 +
                         (long)mt * 64 + dd;
          g.C[o] = v;
        } else {
          const long o = (long)mb * g.c_bs + (long)mt * g.ldc + n;
          if (EPI & kEpiResidual) v += g.R[o];
          g.C[o] = v;
        }
      }
    }
  }
}

template <int EPI>
void launch_gemm_t(const GemmDev& g, int blocks, int bk, hipStream_t s) {
  if (bk == 64 && g.K % 64 == 0) {
    hipLaunchKernelGGL((gemm_f32_128x128<EPI, 64>), dim3(blocks), dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_f32_128x128<EPI, 32>), dim3(blocks), dim3(256), 0, s, g);
  }
}

}  // namespace

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
  GemmDev g{a.A,   a.W,   a.C,    a.bias, a.R,   a.pos,        a.M,        a.N,        a.K,
            a.a_rpb, a.a_bs, a.lda, a.c_rpb, a.c_bs, a.ldc, a.pos_period, a.kv_batch, a.kv_heads,
            a.kv_dmodel};
  const int blocks = ((a.M + BM - 1) / BM) * (a.N / BN);
  switch (epi) {
    case 0: launch_gemm_t<0>(g, blocks, a.bk, s); break;
    case kEpiBias: launch_gemm_t<kEpiBias>(g, blocks, a.bk, s); break;
    case kEpiBias | kEpiGelu: launch_gemm_t<kEpiBias | kEpiGelu>(g, blocks, a.bk, s); break;
    case kEpiBias | kEpiResidual: launch_gemm_t<kEpiBias | kEpiResidual>(g, blocks, a.bk, s); break;
    case kEpiBias | kEpiGelu | kEpiPos: launch_gemm_t<kEpiBias | kEpiGelu | kEpiPos>(g, blocks, a.bk, s); break;
    case kEpiBias | kEpiKvLayout: launch_gemm_t<kEpiBias | kEpiKvLayout>(g, blocks, a.bk, s); break;
    default: abort();
  }
}

}  // namespace wt

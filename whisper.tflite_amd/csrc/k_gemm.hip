// fp32 GEMM kernels for gfx950 on v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered
// fmaf chain, 64 FLOP/clk/SIMD).  These replace the arithmetic the reference runs
// inside tflite::Interpreter::Invoke() (whisper.tflite/whisper.cpp:295, :375) for
// every Conv1D / Linear of the encoder and decoder graphs.
//
// gemm_f32_tile: 256 threads = 4 wavefronts (2x2), each wavefront owns MI x NI MFMA tiles
// of 32x32 (64 accumulator registers for the 128x128 tile).  A and W tiles are staged
// global -> registers -> LDS (issue-early / write-late), LDS rows padded to BK + 4 floats so
// the ds_read_b128 fragment reads are bank-conflict-free.  blockIdx is remapped so that the
// tiles of one XCD (blockIdx % 8) are consecutive in (m, n) order and share their A panel
// through that XCD's L2.  The epilogue goes through a per-wavefront LDS transpose (16-byte
// global accesses).  tools/mfma_probe.hip measures what bounds this loop.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;


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

// Tile template: BM x BN output tile (64 or 128 each), 4 wavefronts as 2 x 2, each owning
// (BM/2) x (BN/2) = MI x NI MFMA tiles of 32 x 32; k-tile BK (32 or 64); DBUF = two LDS
// buffers and one barrier per k-tile instead of two.
template <int EPI, int BM, int BN, int BK, bool DBUF, bool PF2 = false>
__global__ __launch_bounds__(256) void gemm_f32_tile(GemmDev g) {
  constexpr int LDS_LD = BK + 4;  // odd multiple of 16 B: conflict-free ds_read_b128
  constexpr int MI = BM / 64, NI = BN / 64;
  constexpr int TPR = BK / 4;     // threads covering one row's k-tile (128 or 256 contiguous B)
  constexpr int RPP = 256 / TPR;  // rows staged per pass
  constexpr int NA = BM / RPP, NB = BN / RPP;  // float4 per thread per k-tile (A, W)
  constexpr int NBUF = DBUF ? 2 : 1;
  constexpr int SLD = NI * 32 + 4;  // epilogue staging row stride (floats)
  constexpr int kTileFloats = NBUF * (BM + BN) * LDS_LD, kStageFloats = 4 * 32 * SLD;
  __shared__ __attribute__((aligned(16))) float smem[kTileFloats > kStageFloats ? kTileFloats : kStageFloats];
  float* const As = smem;
  float* const Bs = smem + NBUF * BM * LDS_LD;

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

  const int srow = tid / TPR, scol = (tid % TPR) * 4;
  const float* a_ptr[NA];
  const float* w_ptr[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + srow + RPP * i;
    m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
    a_ptr[i] = g.A + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + scol;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) w_ptr[i] = g.W + (long)(n0 + srow + RPP * i) * g.K + scol;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 ra[NA], rb[NB];
  f32x4 ra2[PF2 ? NA : 1], rb2[PF2 ? NB : 1];  // second register stage (prefetch depth 2)
  auto load_into = [&](f32x4* xa, f32x4* xb, int kt) {
#pragma unroll
    for (int i = 0; i < NA; ++i) xa[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < NB; ++i) xb[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + kt * BK);
  };
  auto store_from = [&](const f32x4* xa, const f32x4* xb, int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<f32x4*>(&As[buf * BM * LDS_LD + (srow + RPP * i) * LDS_LD + scol]) = xa[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<f32x4*>(&Bs[buf * BN * LDS_LD + (srow + RPP * i) * LDS_LD + scol]) = xb[i];
  };
  auto load_tile = [&](int kt) { load_into(ra, rb, kt); };
  auto store_tile = [&](int buf) { store_from(ra, rb, buf); };
  auto compute = [&](int buf) {
    const float* Ab = As + buf * BM * LDS_LD + (wm * (BM / 2) + l31) * LDS_LD + 4 * lh;
    const float* Bb = Bs + buf * BN * LDS_LD + (wn * (BN / 2) + l31) * LDS_LD + 4 * lh;
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      // lane (row l31, half lh) takes k = 8*kq + 4*lh + j for MFMA step j: A and B use the
      // same k permutation, so the contraction is complete and exact.
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_LD + kq * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_LD + kq * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
  };

  const int nkt = g.K / BK;
  load_tile(0);
  if (PF2) {
    // Two k-tiles of global loads in flight: HBM latency (~2-3 us under load) exceeds one
    // k-tile of MFMA work (4096 cycles), so a tile is requested two iterations before its
    // LDS write.  Register stages alternate (static names: the loop is unrolled by two).
    static_assert(!PF2 || DBUF, "prefetch depth 2 uses both LDS buffers");
    store_tile(0);
    if (nkt > 1) load_into(ra, rb, 1);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
      if (kt + 2 < nkt) load_into(ra2, rb2, kt + 2);
      compute(0);
      if (kt + 1 < nkt) store_from(ra, rb, 1);
      __syncthreads();
      if (kt + 1 < nkt) {
        if (kt + 3 < nkt) load_into(ra, rb, kt + 3);
        compute(1);
        if (kt + 2 < nkt) store_from(ra2, rb2, 0);
        __syncthreads();
      }
    }
  } else if (DBUF) {
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nkt) load_tile(kt + 1);
      compute(cur);
      if (kt + 1 < nkt) store_tile(cur ^ 1);
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      store_tile(0);
      __syncthreads();
      if (kt + 1 < nkt) load_tile(kt + 1);
      compute(0);
      __syncthreads();
    }
  }

  // Epilogue.  The C/D layout (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
  // would make every lane issue 16 * MI * NI single-dword stores (and as many residual loads):
  // measured on a probe of this loop, that costs +64 % on a K = 384 GEMM.  Each wavefront
  // instead transposes 32-row slabs of its tile through a private LDS stage (the operand
  // tiles are dead by now) and moves 16 bytes per lane: 4x fewer memory instructions, whole
  // 128/256-byte row segments per 8/16 lanes.
  __syncthreads();  // every wavefront is done reading the operand tiles
  float* const stage = smem + wid * (32 * SLD);
  constexpr int LPR = NI * 8;        // lanes per staged row (one float4 each)
  constexpr int RPS = 64 / LPR;      // rows per pass
  const int prow = lane / LPR, c4 = (lane % LPR) * 4;
  const int n = n0 + wn * (BN / 2) + c4;
  f32x4 bias4 = {0, 0, 0, 0};
  if (EPI & kEpiBias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + ni * 32 + l31] = acc[mi][ni][r];
    // the stage is private to this wavefront and LDS executes a wave's operations in order
#pragma unroll
    for (int p = 0; p < 32 / RPS; ++p) {
      const int row = p * RPS + prow;
      const int m = m0 + wm * (BM / 2) + mi * 32 + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * SLD + c4]);
      if (m < g.M) {
        const int mb = m / g.c_rpb, mt = m % g.c_rpb;
        v += bias4;
        if (EPI & kEpiGelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (EPI & kEpiPos) v += *reinterpret_cast<const f32x4*>(g.pos + (long)(m % g.pos_period) * g.N + n);
        if (EPI & kEpiKvLayout) {
          const int slab = n / g.kv_dmodel, rem = n % g.kv_dmodel;
          const int head = rem >> 6, dd = rem & 63;  // 4 consecutive dd: c4 is a multiple of 4
          const long o = (((long)slab * g.kv_batch + mb) * g.kv_heads + head) * (long)g.c_rpb * 64 +
                         (long)mt * 64 + dd;
          *reinterpret_cast<f32x4*>(g.C + o) = v;
        } else {
          const long o = (long)mb * g.c_bs + (long)mt * g.ldc + n;
          if (EPI & kEpiResidual) v += *reinterpret_cast<const f32x4*>(g.R + o);
          *reinterpret_cast<f32x4*>(g.C + o) = v;
        }
      }
    }
  }
}

template <int EPI, int BM, int BN, int BK, bool DBUF, bool PF2 = false>
void launch_tile(const GemmDev& g, hipStream_t s) {
  const int blocks = ((g.M + BM - 1) / BM) * (g.N / BN);
  hipLaunchKernelGGL((gemm_f32_tile<EPI, BM, BN, BK, DBUF, PF2>), dim3(blocks), dim3(256), 0, s, g);
}

// variant: 0 = 128x128x32 (3 blocks/CU), 1 = 128x128x64, 2 = 128x128x32 double-buffered,
//          3 = 128x64x32, 4 = 64x128x32, 5 = 128x64x32 double-buffered, 6 = 64x64x32,
//          7 = 128x128x32 double-buffered + global prefetch depth 2, 8 = 128x64x32 likewise,
//          9 = 192x128x32 (wave tile 96x64)
template <int EPI>
void launch_gemm_t(const GemmDev& g, int variant, hipStream_t s) {
  if (variant == 1 && g.K % 64 != 0) variant = 0;
  switch (variant) {
    case 0: launch_tile<EPI, 128, 128, 32, false>(g, s); break;
    case 1: launch_tile<EPI, 128, 128, 64, false>(g, s); break;
    case 2: launch_tile<EPI, 128, 128, 32, true>(g, s); break;
    case 3: launch_tile<EPI, 128, 64, 32, false>(g, s); break;
    case 4: launch_tile<EPI, 64, 128, 32, false>(g, s); break;
    case 5: launch_tile<EPI, 128, 64, 32, true>(g, s); break;
    case 6: launch_tile<EPI, 64, 64, 32, false>(g, s); break;
    case 7: launch_tile<EPI, 128, 128, 32, true, true>(g, s); break;
    case 8: launch_tile<EPI, 128, 64, 32, true, true>(g, s); break;
    case 9: launch_tile<EPI, 192, 128, 32, false>(g, s); break;
    default: abort();
  }
}

}  // namespace

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
  GemmDev g{a.A,   a.W,   a.C,    a.bias, a.R,   a.pos,        a.M,        a.N,        a.K,
            a.a_rpb, a.a_bs, a.lda, a.c_rpb, a.c_bs, a.ldc, a.pos_period, a.kv_batch, a.kv_heads,
            a.kv_dmodel};
  if (a.N % 128 != 0 || a.K % 32 != 0 || a.M < 1) abort();  // shape contract of the kernels
  int v = a.variant;
  if (v < 0) {
    // auto: 128x128 tiles unless they make fewer than three rounds of 3 blocks x 256 CUs, where
    // the ragged last round costs 15-20 % (measured); 64x128 tiles halve the quantum
    const long blocks128 = (long)((a.M + 127) / 128) * (a.N / 128);
    v = blocks128 < 3 * 768 ? 4 : 0;
  }
  switch (epi) {
    case 0: launch_gemm_t<0>(g, v, s); break;
    case kEpiBias: launch_gemm_t<kEpiBias>(g, v, s); break;
    case kEpiBias | kEpiGelu: launch_gemm_t<kEpiBias | kEpiGelu>(g, v, s); break;
    case kEpiBias | kEpiResidual: launch_gemm_t<kEpiBias | kEpiResidual>(g, v, s); break;
    case kEpiBias | kEpiGelu | kEpiPos: launch_gemm_t<kEpiBias | kEpiGelu | kEpiPos>(g, v, s); break;
    case kEpiBias | kEpiKvLayout: launch_gemm_t<kEpiBias | kEpiKvLayout>(g, v, s); break;
    default: abort();
  }
}

}  // namespace wt

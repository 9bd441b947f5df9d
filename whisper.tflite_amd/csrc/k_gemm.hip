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

#include <cmath>
#include <cstdlib>

#include "bf16_split.h"
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
  const unsigned short* Wp;  // W pre-split into bf16 planes [3][N][K] (split_planes), or nullptr
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
  float a_scale, w_scale, descale;  // two-plane fp16 kernels
};

// Epilogue shared by the GEMM kernels.  The C/D layout (col = lane & 31, row = (r & 3) +
// 8 * (r >> 2) + 4 * (lane >> 5)) would make every lane issue 16 * MI * NI single-dword stores
// (and as many residual loads): measured on a probe of the fp32 loop, that costs +64 % on a
// K = 384 GEMM.  Each wavefront instead transposes 32-row slabs of its tile through a private
// LDS stage (the operand tiles are dead by now) and moves 16 bytes per lane: 4x fewer memory
// instructions, whole 128/256-byte row segments per 8/16 lanes.
template <int EPI, int BM, int BN, int MI, int NI, bool FULL>
__device__ __forceinline__ void tile_epilogue_rows(const GemmDev& g, f32x16 (&acc)[MI][NI], float* smem, int m0,
                                                   int n0) {
  constexpr int SLD = NI * 32 + 4;  // staging row stride (floats)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  float* const stage = smem + wid * (32 * SLD);
  constexpr int LPR = NI * 8;        // lanes per staged row (one float4 each)
  constexpr int RPS = 64 / LPR;      // rows per pass
  const int prow = lane / LPR, c4 = (lane % LPR) * 4;
  const int n = n0 + wn * (BN / 2) + c4;
  f32x4 bias4 = {0, 0, 0, 0};
  if (EPI & kEpiBias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
  // kEpiKvLayout: the column decomposition does not depend on the row
  const int slab = (EPI & kEpiKvLayout) ? n / g.kv_dmodel : 0, rem = (EPI & kEpiKvLayout) ? n % g.kv_dmodel : 0;
  const int head = rem >> 6, dd = rem & 63;  // 4 consecutive dd: c4 is a multiple of 4
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + ni * 32 + l31] = acc[mi][ni][r];
    // the stage is private to this wavefront and LDS executes a wave's operations in order.
    // One division per 32-row slab: rows advance by at most 31 < c_rpb, pos_period (host-checked).
    const int mbase = m0 + wm * (BM / 2) + mi * 32;
    const int mb0 = mbase / g.c_rpb, mt0 = mbase % g.c_rpb;
    const int mp0 = (EPI & kEpiPos) ? mbase % g.pos_period : 0;
#pragma unroll
    for (int p = 0; p < 32 / RPS; ++p) {
      const int row = p * RPS + prow;
      f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * SLD + c4]);
      if (FULL || mbase + row < g.M) {
        int mb = mb0, mt = mt0 + row;
        if (mt >= g.c_rpb) mt -= g.c_rpb, mb += 1;
        v += bias4;
        if (EPI & kEpiGelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (EPI & kEpiPos) {
          int mp = mp0 + row;
          if (mp >= g.pos_period) mp -= g.pos_period;
          v += *reinterpret_cast<const f32x4*>(g.pos + (long)mp * g.N + n);
        }
        if (EPI & kEpiKvLayout) {
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

template <int EPI, int BM, int BN, int MI, int NI>
__device__ __forceinline__ void tile_epilogue(const GemmDev& g, f32x16 (&acc)[MI][NI], float* smem, int m0,
                                              int n0) {
  __syncthreads();  // every wavefront is done reading the operand tiles
  // tiles that lie entirely inside M (all of them when M % BM == 0) take a branch-free path: per-row
  // conditions put every store in its own basic block behind a full s_waitcnt vmcnt(0)
  if (m0 + BM <= g.M) {
    tile_epilogue_rows<EPI, BM, BN, MI, NI, true>(g, acc, smem, m0, n0);
  } else {
    tile_epilogue_rows<EPI, BM, BN, MI, NI, false>(g, acc, smem, m0, n0);
  }
}

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

  tile_epilogue<EPI, BM, BN, MI, NI>(g, acc, smem, m0, n0);
}

// gemm_split_tile: the same GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the
// fp32 MFMA rate per k).  NS = 3: every fp32 operand element is split EXACTLY into three bf16
// planes (bf16_split.h: x = h1 + h2 + h3, 8 significant bits each, the subtractions are exact),
// and the six plane products with weight >= 2^-16 are accumulated in fp32:
//   a.b = a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1) + dropped, |dropped| < 2^-22 |a||b| (bf16_split.h)
// Each bf16 x bf16 product is exact in fp32, so the result carries fp32-level error (measured
// against fp64 next to the fp32-MFMA kernel in tests/test_gpu_kernels.py) at 6/16 of the MFMA
// cycles.  NS = 1 rounds the operands to bf16 (RNE) and is the bf16 compute mode of
// BASELINE configs[3].  Operands stay fp32 in HBM; the split happens between the global load
// and the LDS write (4 VALU ops + 1.5 v_perm per element, hidden behind the partner
// wavefront's MFMAs at 2 blocks per CU).  128 x 128 x 32 tiles, 4 wavefronts as 2 x 2, two
// k-tiles of global loads in flight (a k-tile is ~1500 MFMA cycles, shorter than HBM latency).
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int NS>
__device__ __forceinline__ void split_store8(const f32x4& lo, const f32x4& hi, unsigned short* dst, int plane_stride) {
  float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  if (NS == 1) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // round to nearest even on the upper 16 bits
      unsigned a = __float_as_uint(x[2 * j]), b = __float_as_uint(x[2 * j + 1]);
      a += 0x7FFFu + ((a >> 16) & 1u);
      b += 0x7FFFu + ((b >> 16) & 1u);
      o[j] = __builtin_amdgcn_perm(b, a, 0x07060302u);
    }
    *reinterpret_cast<u32x4*>(dst) = o;
    return;
  }
  u32x4_t o[3];
  split8_planes(x, o);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * plane_stride) = o[p];
}

template <int EPI, int NS, bool WPRE = false>
__global__ __launch_bounds__(256, 2) void gemm_split_tile(GemmDev g) {
  constexpr int BM = 128, BN = 128, BK = 32, MI = 2, NI = 2;
  constexpr int LD = BK + 8;            // bf16 per LDS row: 80 B, an odd multiple of 16 B
  constexpr int PLANE = BM * LD;        // bf16 per operand plane
  constexpr int kTileBytes = 2 * NS * PLANE * 2, kStageBytes = 4 * 32 * (NI * 32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[kTileBytes > kStageBytes ? kTileBytes : kStageBytes];
  unsigned short* const As = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* const Bs = As + NS * PLANE;

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

  // staging: 4 threads per row, 8 consecutive k each (two float4), 64 rows per pass
  const int srow = tid >> 2, scol = (tid & 3) * 8;
  const float* a_ptr[2];
  const float* w_ptr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + srow + 64 * i;
    m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
    a_ptr[i] = g.A + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + scol;
    w_ptr[i] = g.W + (long)(n0 + srow + 64 * i) * g.K + scol;
  }
  // pre-split W: plane p of row n lives at Wp + (p * N + n) * K, 8 bf16 (16 B) per thread
  const unsigned short* wp_ptr[2];
  const long wp_plane = (long)g.N * g.K;
#pragma unroll
  for (int i = 0; i < 2; ++i) wp_ptr[i] = g.Wp + (long)(n0 + srow + 64 * i) * g.K + scol;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // two register stages: {A pass 0, A pass 1} x 2 float4, then W: {pass 0, pass 1} x 2 float4, or
  // (WPRE) {pass 0, pass 1} x NS planes of 8 bf16
  constexpr int NST = 4 + (WPRE ? 2 * NS : 4);
  f32x4 st0[NST], st1[NST];
  auto load_into = [&](f32x4* st, int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      st[2 * i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BK);
      st[2 * i + 1] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BK + 4);
      if (WPRE) {
#pragma unroll
        for (int p = 0; p < NS; ++p)
          st[4 + NS * i + p] = *reinterpret_cast<const f32x4*>(wp_ptr[i] + p * wp_plane + kt * BK);
      } else {
        st[4 + 2 * i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + kt * BK);
        st[4 + 2 * i + 1] = *reinterpret_cast<const f32x4*>(w_ptr[i] + kt * BK + 4);
      }
    }
  };
  auto store_from = [&](const f32x4* st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      split_store8<NS>(st[2 * i], st[2 * i + 1], As + (srow + 64 * i) * LD + scol, PLANE);
      if (WPRE) {
#pragma unroll
        for (int p = 0; p < NS; ++p)
          *reinterpret_cast<f32x4*>(Bs + p * PLANE + (srow + 64 * i) * LD + scol) = st[4 + NS * i + p];
      } else {
        split_store8<NS>(st[4 + 2 * i], st[4 + 2 * i + 1], Bs + (srow + 64 * i) * LD + scol, PLANE);
      }
    }
  };
  auto compute = [&]() {
    const unsigned short* Ab = As + (wm * 64 + l31) * LD + 8 * lh;
    const unsigned short* Bb = Bs + (wn * 64 + l31) * LD + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      // lane (row l31, half lh) holds k = 16*ks + 8*lh + 0..7 of its row, for A and W alike
      bf16x8 af[MI][NS], bf[NI][NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          af[i][p] = *reinterpret_cast<const bf16x8*>(Ab + p * PLANE + i * 32 * LD + ks * 16);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + p * PLANE + j * 32 * LD + ks * 16);
      }
      // smallest products first
#pragma unroll
      for (int w = 2 * (NS - 1) > 2 ? 2 : 2 * (NS - 1); w >= 0; --w)
#pragma unroll
        for (int pa = 0; pa < NS; ++pa) {
          const int pb = w - pa;
          if (pb < 0 || pb >= NS) continue;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nkt = g.K / BK;
  load_into(st0, 0);
  if (nkt > 1) load_into(st1, 1);
  for (int kt = 0; kt < nkt; kt += 2) {
    store_from(st0);
    __syncthreads();
    if (kt + 2 < nkt) load_into(st0, kt + 2);
    compute();
    __syncthreads();
    if (kt + 1 < nkt) {
      store_from(st1);
      __syncthreads();
      if (kt + 3 < nkt) load_into(st1, kt + 3);
      compute();
      __syncthreads();
    }
  }
  tile_epilogue<EPI, BM, BN, MI, NI>(g, acc, reinterpret_cast<float*>(smem_raw), m0, n0);
}

template <int EPI, int NS>
void launch_split(const GemmDev& g, hipStream_t s) {
  const int blocks = ((g.M + 127) / 128) * (g.N / 128);
  if (g.Wp && NS == 3) {
    hipLaunchKernelGGL((gemm_split_tile<EPI, NS, true>), dim3(blocks), dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_split_tile<EPI, NS, false>), dim3(blocks), dim3(256), 0, s, g);
  }
}

// gemm_split16_tile: the 3-plane split GEMM software-pipelined inside each wavefront.  k-tiles of
// 16 with two LDS buffers (73.7 KB, 2 blocks per CU): while the 24 MFMAs of k-tile t run, the
// same wavefront splits the registers of k-tile t+1 and writes them to the other buffer (the
// MFMA pipe is busy 32 cycles per instruction and holds vector issue for 8 of them), and the
// global loads of k-tile t+3 are in flight.  One barrier per k-tile.
template <int EPI, int SCHED, int ABL = 0, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_split16_tile(GemmDev g) {
  // F16: two fp16 planes and three products (bf16_split.h, split8_f16x2) instead of three bf16 planes and
  // six; A and W are scaled by powers of two (GemmArgs::a_scale, w_scale: from operand bounds) into fp16's
  // normal range and the accumulators are scaled back before the epilogue
  constexpr int BM = 128, BN = 128, BK = 16, MI = 2, NI = 2, NS = F16 ? 2 : 3;
  // bf16 per LDS row: 32 B, unpadded. The two 16-byte chunks of a row are stored swapped when bit 3
  // of the row is set: a 16-lane group of a ds_read_b128 (16 consecutive rows, same k half) then
  // covers all 64 banks once, and the staging writes (both chunks of 4 consecutive rows per 8
  // lanes) stay 128 contiguous bytes.
  constexpr int LD = BK;
  constexpr int PLANE = BM * LD;        // bf16 per operand plane
  constexpr int BUF = 2 * NS * PLANE;   // bf16 per buffer (A planes, then W planes)
  constexpr int kTileBytes = 2 * BUF * 2 + ((ABL & 8) ? (55700 - 2 * BUF * 2 > 0 ? 55700 - 2 * BUF * 2 : 0) : 0)  /* ABL 8: occupancy pad, results unchanged */, kStageBytes = 4 * 32 * (NI * 32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[kTileBytes > kStageBytes ? kTileBytes : kStageBytes];
  unsigned short* const lds = reinterpret_cast<unsigned short*>(smem_raw);

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

  // staging: 2 threads per row, 8 consecutive k each (two float4); 128 rows of A and of W per k-tile
  const int srow = tid >> 1, scol = (tid & 1) * 8;
  int m = m0 + srow;
  m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
  const float* const a_ptr = g.A + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + scol;
  const float* const w_ptr = g.W + (long)(n0 + srow) * g.K + scol;
  const int st_off = srow * LD + (((tid & 1) ^ ((srow >> 3) & 1)) * 8);  // this thread's slot inside a plane

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 st0[4], st1[4];  // {A lo, A hi, W lo, W hi}
  auto load_into = [&](f32x4* st, int kt) {
    if ((ABL & 1) && kt > 2) return;  // timing ablation: no global loads in the steady state
    st[0] = *reinterpret_cast<const f32x4*>(a_ptr + kt * BK);
    st[1] = *reinterpret_cast<const f32x4*>(a_ptr + kt * BK + 4);
    st[2] = *reinterpret_cast<const f32x4*>(w_ptr + kt * BK);
    st[3] = *reinterpret_cast<const f32x4*>(w_ptr + kt * BK + 4);
  };
  auto store_from = [&](const f32x4* st, int buf) {
    if (ABL & 2) {  // timing ablation: no split arithmetic, same LDS writes
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        *reinterpret_cast<f32x4*>(lds + buf * BUF + p * PLANE + st_off) = st[p & 1];
        *reinterpret_cast<f32x4*>(lds + buf * BUF + (NS + p) * PLANE + st_off) = st[2 + (p & 1)];
      }
      return;
    }
    if (F16) {
      const float xa[8] = {st[0][0], st[0][1], st[0][2], st[0][3], st[1][0], st[1][1], st[1][2], st[1][3]};
      const float xw[8] = {st[2][0], st[2][1], st[2][2], st[2][3], st[3][0], st[3][1], st[3][2], st[3][3]};
      u32x4_t oa[3], ow[3];
      split8_f16x2(xa, g.a_scale, oa);
      split8_f16x2(xw, g.w_scale, ow);
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        *reinterpret_cast<u32x4_t*>(lds + buf * BUF + p * PLANE + st_off) = oa[p];
        *reinterpret_cast<u32x4_t*>(lds + buf * BUF + (NS + p) * PLANE + st_off) = ow[p];
      }
      return;
    }
    split_store8<3>(st[0], st[1], lds + buf * BUF + st_off, PLANE);
    split_store8<3>(st[2], st[3], lds + buf * BUF + NS * PLANE + st_off, PLANE);
  };
  const int swz = 8 * (lh ^ ((l31 >> 3) & 1));
  const int a_off = (wm * 64 + l31) * LD + swz, b_off = NS * PLANE + (wn * 64 + l31) * LD + swz;
  bf16x8 af[MI][NS], bf[NI][NS];
  auto read_frags = [&](int buf) {
    const unsigned short* base = lds + buf * BUF;
#pragma unroll
    for (int p = 0; p < NS; ++p) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i][p] = *reinterpret_cast<const bf16x8*>(base + a_off + p * PLANE + i * 32 * LD);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j][p] = *reinterpret_cast<const bf16x8*>(base + b_off + p * PLANE + j * 32 * LD);
    }
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int w = (ABL & 4) ? 0 : NS - 1; w >= 0; --w)  // smallest products first (ablation 4: one product)
#pragma unroll
      for (int pa = 0; pa < NS; ++pa) {
        const int pb = w - pa;
        if (pb < 0 || pb >= NS) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            if (F16) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af[i][pa]),
                                                                 __builtin_bit_cast(half8, bf[j][pb]), acc[i][j], 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0);
            }
      }
  };
  auto interleave = [&]() {
    if (SCHED) {
      // per MFMA: 4 split VALU ops ride in its shadow; a DS write every fourth
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
        if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
      }
    }
  };

  const int nkt = g.K / BK;
  load_into(st0, 0);
  if (nkt > 1) load_into(st1, 1);
  store_from(st0, 0);
  if (nkt > 2) load_into(st0, 2);
  __syncthreads();
  int kt = 0;
  // steady state, branch-free so that the scheduler can interleave the split with the MFMAs:
  // even k-tile computes buffer 0 while k-tile kt+1 (st1) is staged into buffer 1, and so on
  for (; kt + 4 < nkt; kt += 2) {
    if (SCHED == 2) {
      // stage first: the registers of k-tile kt+1 go to the idle buffer and are refilled with
      // k-tile kt+3 a whole MFMA phase earlier than in the compute-first order
      store_from(st1, 1);
      load_into(st1, kt + 3);
      read_frags(0);
      mfmas();
      __syncthreads();
      store_from(st0, 0);
      load_into(st0, kt + 4);
      read_frags(1);
      mfmas();
      __syncthreads();
      continue;
    }
    read_frags(0);
    mfmas();
    store_from(st1, 1);
    interleave();
    load_into(st1, kt + 3);
    __syncthreads();
    read_frags(1);
    mfmas();
    store_from(st0, 0);
    interleave();
    load_into(st0, kt + 4);
    __syncthreads();
  }
  for (; kt < nkt; kt += 2) {  // the last k-tiles: nothing left to load / stage
    read_frags(0);
    mfmas();
    if (kt + 1 < nkt) store_from(st1, 1);
    if (kt + 3 < nkt) load_into(st1, kt + 3);
    __syncthreads();
    if (kt + 1 < nkt) {
      read_frags(1);
      mfmas();
      if (kt + 2 < nkt) store_from(st0, 0);
      if (kt + 4 < nkt) load_into(st0, kt + 4);
      __syncthreads();
    }
  }
  if (F16) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] *= g.descale;
  }
  tile_epilogue<EPI, BM, BN, MI, NI>(g, acc, reinterpret_cast<float*>(smem_raw), m0, n0);
}

template <int EPI, int SCHED, int ABL = 0, bool F16 = false>
void launch_split16(const GemmDev& g, hipStream_t s) {
  const int blocks = ((g.M + 127) / 128) * (g.N / 128);
  hipLaunchKernelGGL((gemm_split16_tile<EPI, SCHED, ABL, F16>), dim3(blocks), dim3(256), 0, s, g);
}

// x[n] -> three bf16 planes out[p * n + i] with x = h1 + h2 + h3 exactly (see gemm_split_tile)
__global__ void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const unsigned u = __float_as_uint(v) + 0x8000u;
  const float r1 = v - __uint_as_float(u & 0xFFFF0000u);
  const unsigned u1 = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
  out[i] = (unsigned short)(u >> 16);
  out[n + i] = (unsigned short)(u1 >> 16);
  out[2 * n + i] = (unsigned short)(__float_as_uint(r2) >> 16);
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
//          10 = 128x128x32 on the bf16 matrix cores, exact 3-plane operand split (fp32 result)
//          11 = 128x128x32 on the bf16 matrix cores, operands rounded to bf16
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
    case 10: launch_split<EPI, 3>(g, s); break;  // fp32 result from six bf16 plane products
    case 11: launch_split<EPI, 1>(g, s); break;  // bf16-rounded operands (configs[3] compute mode)
    case 13: launch_split16<EPI, 0>(g, s); break;  // split-3, k-tiles of 16, double-buffered LDS
    case 14: launch_split16<EPI, 1>(g, s); break;  // same with an explicit MFMA/VALU interleave
    case 15: launch_split16<EPI, 2>(g, s); break;  // same, staging before the MFMAs of a k-tile
    case 16: launch_split16<EPI, 0, 8>(g, s); break;  // variant 13 with LDS padded to 54 KB: 2 blocks per CU, which
                                                      // leaves registers and LDS for co-resident decoder blocks
    case 17: launch_split16<EPI, 0, 0, true>(g, s); break;  // two fp16 planes, three products (22-bit operands)
    case 18: launch_split16<EPI, 0, 8, true>(g, s); break;  // same at 2 blocks per CU
    case 21: launch_split16<EPI, 0, 1>(g, s); break;  // timing ablations of 13 (wrong results)
    case 22: launch_split16<EPI, 0, 2>(g, s); break;
    case 23: launch_split16<EPI, 0, 3>(g, s); break;
    case 24: launch_split16<EPI, 0, 4>(g, s); break;
    case 27: launch_split16<EPI, 0, 7>(g, s); break;
    default: abort();
  }
}

}  // namespace

int gemm_occupancy(int variant) {
  int n = -1;
  switch (variant) {
    case 0: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_f32_tile<kEpiBias, 128, 128, 32, false>, 256, 0); break;
    case 10: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split_tile<kEpiBias, 3, false>, 256, 0); break;
    case 13: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split16_tile<kEpiBias, 0, 0>, 256, 0); break;
    case 14: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split16_tile<kEpiBias, 1, 0>, 256, 0); break;
    case 15: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split16_tile<kEpiBias, 2, 0>, 256, 0); break;
    default: break;
  }
  return n;
}

float f16_scale_for(float bound) {
  if (!(bound > 0.0f)) return 1.0f;
  int e = 0;
  (void)std::frexp(16384.0f / bound, &e);  // 16384 / bound = m * 2^e, m in [0.5, 1)
  e -= 1;
  e = e > 24 ? 24 : (e < -24 ? -24 : e);
  return std::ldexp(1.0f, e);
}

void launch_split_planes(const float* x, unsigned short* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, n);
}

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
  GemmDev g{a.A,   a.W, a.Wp,  a.C,    a.bias, a.R,   a.pos,        a.M,        a.N,        a.K,
            a.a_rpb, a.a_bs, a.lda, a.c_rpb, a.c_bs, a.ldc, a.pos_period, a.kv_batch, a.kv_heads,
            a.kv_dmodel, a.a_scale, a.w_scale, 1.0f / (a.a_scale * a.w_scale)};
  // shape contract of the kernels (the epilogue wraps batch / position rows at most once per 32 rows)
  if (a.N % 128 != 0 || a.K % 32 != 0 || a.M < 1 || a.c_rpb < 32 || a.pos_period < (epi & kEpiPos ? 32 : 1)) abort();
  int v = a.variant;
  if (v < 0) v = 13;  // auto for callers that do not choose: the full-range bf16 three-plane split kernel
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

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
//
// This translation unit holds the fp32-in / fp32-out GEMMs: variant 0 = fp32 MFMA (bench.py's exact-fp32 leg, the
// reference form of the accuracy tests) and 13 / 16 = fp32 operands split into three bf16 planes inside the loop
// (full fp32 operand range: the log-mel front end's two GEMMs and the per-contraction fall-back of the encoder,
// engine.cpp).  The encoder's default GEMM is k_gemm_planes.hip.
#include <hip/hip_runtime.h>

#include <cmath>

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

// gemm_f32_tile: BM x BN output tile, 4 wavefronts as 2 x 2, each owning (BM/2) x (BN/2) = MI x NI MFMA
// tiles of 32 x 32; k-tile BK; one LDS buffer, global loads of k-tile t+1 in flight during the MFMAs of t.
template <int EPI, int BM, int BN, int BK>
__global__ __launch_bounds__(256) void gemm_f32_tile(GemmDev g) {
  constexpr int LDS_LD = BK + 4;  // odd multiple of 16 B: conflict-free ds_read_b128
  constexpr int MI = BM / 64, NI = BN / 64;
  constexpr int TPR = BK / 4;     // threads covering one row's k-tile (128 contiguous B)
  constexpr int RPP = 256 / TPR;  // rows staged per pass
  constexpr int NA = BM / RPP, NB = BN / RPP;  // float4 per thread per k-tile (A, W)
  constexpr int SLD = NI * 32 + 4;  // epilogue staging row stride (floats)
  constexpr int kTileFloats = (BM + BN) * LDS_LD, kStageFloats = 4 * 32 * SLD;
  __shared__ __attribute__((aligned(16))) float smem[kTileFloats > kStageFloats ? kTileFloats : kStageFloats];
  float* const As = smem;
  float* const Bs = smem + BM * LDS_LD;

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
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + kt * BK);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<f32x4*>(&As[(srow + RPP * i) * LDS_LD + scol]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<f32x4*>(&Bs[(srow + RPP * i) * LDS_LD + scol]) = rb[i];
  };
  auto compute = [&]() {
    const float* Ab = As + (wm * (BM / 2) + l31) * LDS_LD + 4 * lh;
    const float* Bb = Bs + (wn * (BN / 2) + l31) * LDS_LD + 4 * lh;
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
  for (int kt = 0; kt < nkt; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nkt) load_tile(kt + 1);
    compute();
    __syncthreads();
  }

  tile_epilogue<EPI, BM, BN, MI, NI>(g, acc, smem, m0, n0);
}

// gemm_bf16_tile: the bf16 compute mode of BASELINE configs[3] on fp32 storage: operands rounded to bf16
// (RNE) between the global load and the LDS write, one v_mfma_f32_32x32x16_bf16 product, fp32
// accumulation.  128 x 128 x 32 tiles, 4 wavefronts as 2 x 2, two k-tiles of global loads in flight.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int NS>
__device__ __forceinline__ void split_store8(const f32x4& lo, const f32x4& hi, unsigned short* dst, int plane_stride) {
  static_assert(NS == 3, "three bf16 planes");
  const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  u32x4_t o[3];
  split8_planes(x, o);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * plane_stride) = o[p];
}

// gemm_split16_tile: the 3-plane split GEMM software-pipelined inside each wavefront.  k-tiles of
// 16 with two LDS buffers (73.7 KB, 2 blocks per CU): while the 24 MFMAs of k-tile t run, the
// same wavefront splits the registers of k-tile t+1 and writes them to the other buffer (the
// MFMA pipe is busy 32 cycles per instruction and holds vector issue for 8 of them), and the
// global loads of k-tile t+3 are in flight.  One barrier per k-tile.
// PAD2: LDS padded to 54 KB so that exactly two blocks share a CU (pipelined mode: leaves registers and LDS
// to co-resident decoder blocks; results unchanged).
template <int EPI, bool PAD2 = false>
__global__ __launch_bounds__(256, 2) void gemm_split16_tile(GemmDev g) {
  constexpr int BM = 128, BN = 128, BK = 16, MI = 2, NI = 2, NS = 3;
  // bf16 per LDS row: 32 B, unpadded. The two 16-byte chunks of a row are stored swapped when bit 3
  // of the row is set: a 16-lane group of a ds_read_b128 (16 consecutive rows, same k half) then
  // covers all 64 banks once, and the staging writes (both chunks of 4 consecutive rows per 8
  // lanes) stay 128 contiguous bytes.
  constexpr int LD = BK;
  constexpr int PLANE = BM * LD;        // bf16 per operand plane
  constexpr int BUF = 2 * NS * PLANE;   // bf16 per buffer (A planes, then W planes)
  constexpr int kTileBytes = 2 * BUF * 2 + (PAD2 ? (55700 - 2 * BUF * 2 > 0 ? 55700 - 2 * BUF * 2 : 0) : 0), kStageBytes = 4 * 32 * (NI * 32 + 4) * 4;
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
    st[0] = *reinterpret_cast<const f32x4*>(a_ptr + kt * BK);
    st[1] = *reinterpret_cast<const f32x4*>(a_ptr + kt * BK + 4);
    st[2] = *reinterpret_cast<const f32x4*>(w_ptr + kt * BK);
    st[3] = *reinterpret_cast<const f32x4*>(w_ptr + kt * BK + 4);
  };
  auto store_from = [&](const f32x4* st, int buf) {
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
    for (int w = NS - 1; w >= 0; --w)  // smallest products first
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
    read_frags(0);
    mfmas();
    store_from(st1, 1);
    load_into(st1, kt + 3);
    __syncthreads();
    read_frags(1);
    mfmas();
    store_from(st0, 0);
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
  tile_epilogue<EPI, BM, BN, MI, NI>(g, acc, reinterpret_cast<float*>(smem_raw), m0, n0);
}

template <int EPI, bool PAD2>
void launch_split16(const GemmDev& g, hipStream_t s) {
  const int blocks = ((g.M + 127) / 128) * (g.N / 128);
  hipLaunchKernelGGL((gemm_split16_tile<EPI, PAD2>), dim3(blocks), dim3(256), 0, s, g);
}

template <int EPI>
void launch_gemm_t(const GemmDev& g, int variant, hipStream_t s) {
  switch (variant) {
    case 0: {  // fp32 MFMA, 128 x 128 x 32 (3 blocks per CU)
      const int blocks = ((g.M + 127) / 128) * (g.N / 128);
      hipLaunchKernelGGL((gemm_f32_tile<EPI, 128, 128, 32>), dim3(blocks), dim3(256), 0, s, g);
      break;
    }
    case 13: launch_split16<EPI, false>(g, s); break;  // three bf16 planes, six products: full fp32 range
    case 16: launch_split16<EPI, true>(g, s); break;   // same at 2 blocks per CU
    default: throw Error(kErrInvalidArg, "gemm_variant must be one of 0, 13, 16");
  }
}

}  // namespace

bool gemm_variant_supported(int variant) {
  return variant == 0 || variant == 13 || variant == 16;
}

int gemm_occupancy(int variant) {
  int n = -1;
  switch (variant) {
    case 0: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_f32_tile<kEpiBias, 128, 128, 32>, 256, 0); break;
    case 13: (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split16_tile<kEpiBias, false>, 256, 0); break;
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

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
  GemmDev g{a.A,   a.W,  a.C,    a.bias, a.R,   a.pos,        a.M,        a.N,        a.K,
            a.a_rpb, a.a_bs, a.lda, a.c_rpb, a.c_bs, a.ldc, a.pos_period, a.kv_batch, a.kv_heads,
            a.kv_dmodel, a.a_scale, a.w_scale, 1.0f / (a.a_scale * a.w_scale)};
  // shape contract of the kernels (the epilogue wraps batch / position rows at most once per 32 rows)
  if (a.N % 128 != 0 || a.K % 32 != 0 || a.M < 1 || a.c_rpb < 32 || a.pos_period < (epi & kEpiPos ? 32 : 1)) {
    throw Error(kErrInvalidArg, "GEMM shape outside the kernel contract (N % 128, K % 32, M >= 1, rows per clip >= 32)");
  }
  int v = a.variant;
  if (v < 0) v = 13;  // auto for callers that do not choose: the full-range bf16 three-plane split kernel
  switch (epi) {
    case 0: launch_gemm_t<0>(g, v, s); break;
    case kEpiBias: launch_gemm_t<kEpiBias>(g, v, s); break;
    case kEpiBias | kEpiGelu: launch_gemm_t<kEpiBias | kEpiGelu>(g, v, s); break;
    case kEpiBias | kEpiResidual: launch_gemm_t<kEpiBias | kEpiResidual>(g, v, s); break;
    case kEpiBias | kEpiGelu | kEpiPos: launch_gemm_t<kEpiBias | kEpiGelu | kEpiPos>(g, v, s); break;
    case kEpiBias | kEpiKvLayout: launch_gemm_t<kEpiBias | kEpiKvLayout>(g, v, s); break;
    default: throw Error(kErrInvalidArg, "unsupported GEMM epilogue combination");
  }
}

}  // namespace wt

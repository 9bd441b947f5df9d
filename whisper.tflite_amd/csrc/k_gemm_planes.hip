// Encoder GEMM on pre-split operands (gfx950).  C = epilogue(A . W^T) like k_gemm.hip, but both operands arrive
// as TWO fp16 PLANES (hi = fp16(x * scale), lo = fp16(x * scale - hi): 22 significand bits, csrc/bf16_split.h) —
// weights split once at load time, activations split by the kernel that PRODUCES them (LayerNorm, the GELU / plain
// epilogues of this kernel, encoder attention), 4 bytes per element exactly like fp32.  The main loop therefore has
// no VALU work at all: planes go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no registers), fragments come
// out with ds_read_b128, and the contraction is three v_mfma_f32_32x32x16_f16 products per 16-deep k-step
// (hi.lo + lo.hi + hi.hi, fp32 accumulation: fp32-level error at 3/16 of the fp32-MFMA cycles).  Round 1's
// gemm_split16_tile re-split every A element on the VALU once per column block (N/128 times) inside the loop.
//
// Tile 192 x 128 x 32, 4 wavefronts as 2 x 2, wave tile 96 x 64 (3 x 2 MFMA tiles: 20 ds_read_b128 per 36 MFMAs).
// M = 48000 rows of a 32-clip batch are 250 row tiles, so every encoder shape fills whole rounds of the 256 CUs
// (750 / 2250 / 3000 / 6000 blocks = 2.93 / 8.8 / 11.7 / 23.4 rounds; 128-row tiles gave 4.39).  Two LDS stages of
// 40 KB: the LDS-DMA of k-tile t + 1 is in flight during the MFMAs of k-tile t, one barrier per k-tile.
// LDS rows are 64 B (32 halfs); the four 16-byte chunks of a row are stored XOR-swizzled by (row >> 2) & 3 —
// applied to the per-lane SOURCE address of the LDS-DMA, whose destination is lane-linear — so that every 16-lane
// group of a ds_read_b128 covers all 64 banks once.
#include <hip/hip_runtime.h>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BM = 192, BN = 128, BK = 32;
constexpr int MI = 3, NI = 2;                       // MFMA tiles per wavefront (rows, columns)
constexpr int kAPlane = BM * BK * 2;                // bytes of one A plane of a stage
constexpr int kWPlane = BN * BK * 2;
constexpr int kStage = 2 * kAPlane + 2 * kWPlane;   // 40960
constexpr int kInstr = kStage / 1024;               // LDS-DMA wave-instructions per stage (40)
constexpr int SLD = NI * 32 + 4;                    // epilogue staging row stride (floats)

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

struct PlaneGemmDev {
  const _Float16* A;   // hi plane; lo plane at A + a_plane
  long a_plane;
  const _Float16* W;   // hi plane [N][K]; lo plane at W + w_plane
  long w_plane;
  float* C;            // fp32 output (kOutF32)
  _Float16* P;         // plane output (kOutPlanes): hi at P, lo at P + p_plane
  long p_plane;
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
  float descale;       // 1 / (a_scale * w_scale)
  float out_scale[3];  // plane output: column n is multiplied by out_scale[n / seg] before the split
  int seg;
};

template <int EPI, bool PLANES_OUT>
__global__ __launch_bounds__(256, 2) void gemm_planes_tile(PlaneGemmDev g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware bijective remap: blocks with equal blockIdx % 8 share an XCD (speed only).
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_tiles = g.N / BN;
  const int m0 = (logical / n_tiles) * BM;
  const int n0 = (logical % n_tiles) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- LDS-DMA source addresses.  Wave w issues instructions q = w, w + 4, ...: q in [0,12) A hi rows 16q..,
  // [12,24) A lo, [24,32) W hi, [32,40) W lo.  Lane i writes LDS slot (row i >> 2, chunk i & 3) of its instruction
  // and reads the global chunk (i & 3) ^ ((row >> 2) & 3) of that row.
  const int srow = lane >> 2;
  const _Float16* a_src[3];
  const _Float16* w_src[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = 16 * (wid + 4 * j) + srow;                      // row inside the A tile
    int m = m0 + row;
    m = m < g.M ? m : g.M - 1;                                      // clamp: rows past M are computed and discarded
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    a_src[j] = g.A + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + chunk * 8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * (wid + 4 * j) + srow;                      // row inside the W tile
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    w_src[j] = g.W + (long)(n0 + row) * g.K + chunk * 8;
  }
  auto issue_stage = [&](int kt, int buf) {
    unsigned char* base = smem + buf * kStage;
    const int ko = kt * BK;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = wid + 4 * j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + ko),
                                       (__attribute__((address_space(3))) void*)(base + q * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + g.a_plane + ko),
                                       (__attribute__((address_space(3))) void*)(base + kAPlane + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wid + 4 * j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + ko),
                                       (__attribute__((address_space(3))) void*)(base + 2 * kAPlane + q * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + g.w_plane + ko),
                                       (__attribute__((address_space(3))) void*)(base + 2 * kAPlane + kWPlane + q * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment addresses: row = tile base (a multiple of 32) + l31, so the swizzle term depends on the lane only
  const int swz = (l31 >> 2) & 3;
  const int a_off = (wm * 96 + l31) * 64, b_off = 2 * kAPlane + (wn * 64 + l31) * 64;
  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * kStage;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int slot = ((ks * 2 + lh) ^ swz) * 16;
      half8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(base + a_off + i * 32 * 64 + slot);
        al[i] = *reinterpret_cast<const half8*>(base + kAPlane + a_off + i * 32 * 64 + slot);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        bh[j] = *reinterpret_cast<const half8*>(base + b_off + j * 32 * 64 + slot);
        bl[j] = *reinterpret_cast<const half8*>(base + kWPlane + b_off + j * 32 * 64 + slot);
      }
      // smallest products first
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nkt = g.K / BK;
  issue_stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();  // k-tile kt has landed (every wave waited for its own LDS-DMA) and k-tile kt - 1 has been read
    if (kt + 1 < nkt) issue_stage(kt + 1, (kt + 1) & 1);
    compute(kt & 1);
  }
  __syncthreads();  // the operand stages are dead: the epilogue reuses them

  // ---- epilogue: each wavefront transposes 32-row slabs of its tile through a private LDS stage and moves 16 bytes
  // per lane (whole row segments per 8 / 16 lanes)
  float* const stage = reinterpret_cast<float*>(smem) + wid * (32 * SLD);
  constexpr int CPL = PLANES_OUT ? 8 : 4;    // columns per lane
  constexpr int LPR = NI * 32 / CPL;         // lanes per staged row
  constexpr int RPS = 64 / LPR;              // rows per pass
  const int prow = lane / LPR, c0 = (lane % LPR) * CPL;
  const int n = n0 + wn * (BN / 2) + c0;
  float bias_v[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) bias_v[e] = (EPI & kEpiBias) ? g.bias[n + e] : 0.0f;
  const float oscale = PLANES_OUT ? g.out_scale[n / g.seg] : 1.0f;  // CPL consecutive columns never straddle a segment
  // kEpiKvLayout: the column decomposition does not depend on the row
  const int slab = (EPI & kEpiKvLayout) ? n / g.kv_dmodel : 0, rem = (EPI & kEpiKvLayout) ? n % g.kv_dmodel : 0;
  const int head = rem >> 6, dd = rem & 63;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + ni * 32 + l31] = acc[mi][ni][r] * g.descale;
    // the stage is private to this wavefront and LDS executes a wave's operations in order.
    // One division per 32-row slab: rows advance by at most 31 < c_rpb, pos_period (host-checked).
    const int mbase = m0 + wm * 96 + mi * 32;
    const int mb0 = mbase / g.c_rpb, mt0 = mbase % g.c_rpb;
    const int mp0 = (EPI & kEpiPos) ? mbase % g.pos_period : 0;
#pragma unroll
    for (int p = 0; p < 32 / RPS; ++p) {
      const int row = p * RPS + prow;
      float v[CPL];
#pragma unroll
      for (int e = 0; e < CPL; e += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(&stage[row * SLD + c0 + e]);
        v[e] = t[0], v[e + 1] = t[1], v[e + 2] = t[2], v[e + 3] = t[3];
      }
      if (mbase + row < g.M) {
        int mb = mb0, mt = mt0 + row;
        if (mt >= g.c_rpb) mt -= g.c_rpb, mb += 1;
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
          v[e] += bias_v[e];
          if (EPI & kEpiGelu) v[e] = gelu_erf(v[e]);
        }
        if (EPI & kEpiPos) {
          int mp = mp0 + row;
          if (mp >= g.pos_period) mp -= g.pos_period;
#pragma unroll
          for (int e = 0; e < CPL; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(g.pos + (long)mp * g.N + n + e);
            v[e] += t[0], v[e + 1] += t[1], v[e + 2] += t[2], v[e + 3] += t[3];
          }
        }
        if (PLANES_OUT) {
          const long o = (long)mb * g.c_bs + (long)mt * g.ldc + n;
          half8 hi, lo;
#pragma unroll
          for (int e = 0; e < CPL; ++e) {
            _Float16 h, l;
            split_f16(v[e] * oscale, &h, &l);
            hi[e] = h;
            lo[e] = l;
          }
          *reinterpret_cast<half8*>(g.P + o) = hi;
          *reinterpret_cast<half8*>(g.P + g.p_plane + o) = lo;
        } else if (EPI & kEpiKvLayout) {
          const long o = (((long)slab * g.kv_batch + mb) * g.kv_heads + head) * (long)g.c_rpb * 64 + (long)mt * 64 + dd;
          *reinterpret_cast<f32x4*>(g.C + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          const long o = (long)mb * g.c_bs + (long)mt * g.ldc + n;
          f32x4 out = {v[0], v[1], v[2], v[3]};
          if (EPI & kEpiResidual) out += *reinterpret_cast<const f32x4*>(g.R + o);
          *reinterpret_cast<f32x4*>(g.C + o) = out;
        }
      }
    }
  }
}

template <int EPI, bool PLANES_OUT>
void launch_planes(const PlaneGemmDev& g, hipStream_t s) {
  const int blocks = ((g.M + BM - 1) / BM) * (g.N / BN);
  constexpr size_t smem = 2 * kStage;  // 80 KB: two blocks share a CU; the epilogue stage (4 x 32 x 68 x 4 B) fits inside
  static const bool raised = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_tile<EPI, PLANES_OUT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)raised;
  hipLaunchKernelGGL((gemm_planes_tile<EPI, PLANES_OUT>), dim3(blocks), dim3(256), smem, s, g);
}

}  // namespace

void launch_gemm_planes(const PlaneGemmArgs& a, int epi, hipStream_t s) {
  PlaneGemmDev g{};
  g.A = reinterpret_cast<const _Float16*>(a.A); g.a_plane = a.a_plane;
  g.W = reinterpret_cast<const _Float16*>(a.W); g.w_plane = a.w_plane;
  g.C = a.C; g.P = reinterpret_cast<_Float16*>(a.P); g.p_plane = a.p_plane;
  g.bias = a.bias; g.R = a.R; g.pos = a.pos;
  g.M = a.M; g.N = a.N; g.K = a.K;
  g.a_rpb = a.a_rpb; g.a_bs = a.a_bs; g.lda = a.lda;
  g.c_rpb = a.c_rpb; g.c_bs = a.c_bs; g.ldc = a.ldc;
  g.pos_period = a.pos_period;
  g.kv_batch = a.kv_batch; g.kv_heads = a.kv_heads; g.kv_dmodel = a.kv_dmodel;
  g.descale = 1.0f / (a.a_scale * a.w_scale);
  g.out_scale[0] = a.out_scale[0]; g.out_scale[1] = a.out_scale[1]; g.out_scale[2] = a.out_scale[2];
  g.seg = a.seg > 0 ? a.seg : a.N;
  const bool planes = a.P != nullptr;
  // shape contract of the kernel (16-byte chunks, whole k-tiles; the epilogue wraps clip / position rows at most once
  // per 32 rows; 8 output columns never straddle a scale segment)
  if (a.N % BN != 0 || a.K % BK != 0 || a.M < 1 || a.c_rpb < 32 || a.pos_period < (epi & kEpiPos ? 32 : 1) || a.lda % 8 != 0 ||
      a.a_bs % 8 != 0 || a.ldc % 8 != 0 || a.c_bs % 8 != 0 || (planes && (g.seg % 8 != 0 || (a.N + g.seg - 1) / g.seg > 3)) ||
      (!planes && !a.C) || !(a.a_scale > 0.0f) || !(a.w_scale > 0.0f)) {
    throw Error(kErrInvalidArg, "plane GEMM shape outside the kernel contract");
  }
  switch (epi | (planes ? 256 : 0)) {
    case kEpiBias: launch_planes<kEpiBias, false>(g, s); break;
    case kEpiBias | kEpiResidual: launch_planes<kEpiBias | kEpiResidual, false>(g, s); break;
    case kEpiBias | kEpiGelu | kEpiPos: launch_planes<kEpiBias | kEpiGelu | kEpiPos, false>(g, s); break;
    case kEpiBias | kEpiKvLayout: launch_planes<kEpiBias | kEpiKvLayout, false>(g, s); break;
    case kEpiBias | 256: launch_planes<kEpiBias, true>(g, s); break;
    case kEpiBias | kEpiGelu | 256: launch_planes<kEpiBias | kEpiGelu, true>(g, s); break;
    default: throw Error(kErrInvalidArg, "unsupported plane GEMM epilogue combination");
  }
}

}  // namespace wt

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

#include <algorithm>
#include <cstdlib>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BK = 32;

struct PlaneGemmDev {
  const _Float16* A;   // hi plane; lo plane at A + a_plane
  long a_plane;
  const _Float16* W;   // hi plane [N][K]; lo plane at W + w_plane
  long w_plane;
  float* C;            // fp32 output
  _Float16* P;         // plane output: hi at P, lo at P + p_plane
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
  // LN: LayerNorm of the finished rows (N == BN: a block owns whole rows), written as planes for the next GEMM
  const float* ln_g;
  const float* ln_b;
  _Float16* ln_P;      // hi plane [M][N]; lo plane at ln_P + ln_plane
  long ln_plane;
  float ln_scale;
  float* ln_y32;       // optional fp32 copy of the LayerNorm output (the encoder's enc_out)
  int* nonfinite;      // optional flag word: a row with a non-finite mean or variance sets it
};

// Block tile 192 x BN with BN = WN * NI * 32: 2 x WN wavefronts, each owning 3 x NI MFMA tiles.
//   WN 2, NI 2: 192 x 128, 4 wavefronts, 40 KB per stage, two blocks per CU (round 2's first shape);
//   WN 4, NI 3: 192 x 384, 8 wavefronts, 72 KB per stage, one block per CU.  What bounds this kernel is the LDS-DMA
//   round trip (one k-tile of prefetch distance, ~2 us under load) against the bytes a CU can hold in flight (LDS):
//   the wide tile contracts 197 FLOP per staged byte instead of 118 and needs 24 ds_read_b128 per 54 MFMAs instead of
//   20 per 36.  Every N of the encoder (384, 1152, 1536, 3072) is a multiple of 384, and 250 row tiles x {1, 3, 4, 8}
//   column tiles fill 0.98 / 2.93 / 3.9 / 7.8 rounds of the 256 CUs.
//   MI = 32-row MFMA tiles per wavefront (two wavefront rows): 3 -> 192 block rows; 4 -> 256 rows x 384 columns, the
//   whole register file (255 VGPRs) and all 160 KB of LDS, 230 FLOP per staged byte — used where 188 row tiles fill the
//   available CUs in fewer rounds than 250 (the CU-masked stream of the pipeline on the N = d_model shapes).
template <int EPI, bool PLANES_OUT, int WN, int NI, int MI, bool LN = false>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 2) void gemm_planes_tile(PlaneGemmDev g) {
  static_assert(!LN || (!PLANES_OUT && WN == 4 && NI == 3), "LayerNorm fusion: fp32 output, 384-column tile");
  constexpr int BN = WN * NI * 32, NW = 2 * WN, BM = 64 * MI;
  constexpr int kAPlane = BM * BK * 2, kWPlane = BN * BK * 2;  // bytes of one plane of a stage
  constexpr int kStage = 2 * kAPlane + 2 * kWPlane;
  constexpr int QA = BM / 16, QW = BN / 16;                   // LDS-DMA instructions per A / W plane (16 rows each)
  constexpr int QT = 2 * QA + 2 * QW, QPW = QT / NW;          // per stage, per wavefront
  static_assert(QT % NW == 0, "whole instructions per wavefront");
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
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- LDS-DMA source addresses.  Instruction q of a stage copies 16 rows x 64 B: q in [0, QA) A hi, [QA, 2 QA) A lo,
  // [2 QA, 2 QA + QW) W hi, then W lo; its LDS destination is the stage base + q * 1024 (lane-linear).  Wave w issues
  // q = w, w + NW, ...  Lane i fills LDS slot (row i >> 2, chunk i & 3) of its instruction with the global chunk
  // (i & 3) ^ ((row >> 2) & 3) of that row: the XOR swizzle lives on the SOURCE side.
  const int srow = lane >> 2;
  // Addresses as (uniform base) + (32-bit per-lane byte offset): the LDS-DMA then takes its base from SGPRs that SALU
  // instructions advance per k-tile, and its issue needs NO vector instruction.  With per-lane 64-bit pointers every
  // issue was preceded by a v_lshl_add_u64 — and on this part a wavefront's VALU instruction does not slip into the
  // other wavefront's MFMA burst (tools/mfma_valu_overlap.hip), so the loads of the next k-tile left a whole MFMA
  // phase late: the main loop took MFMA time + DMA time (profiles/r03_gemm_mainloop_ablation.txt).
  unsigned voff[QPW];
  const unsigned char* ubase[QPW];
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    const int q = wid + NW * j;
    const bool is_a = q < 2 * QA;
    const int qq = is_a ? (q < QA ? q : q - QA) : (q - 2 * QA < QW ? q - 2 * QA : q - 2 * QA - QW);
    const bool lo = is_a ? q >= QA : q - 2 * QA >= QW;
    const int row = 16 * qq + srow;
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    if (is_a) {
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
      voff[j] = (unsigned)(2 * ((lo ? g.a_plane : 0) + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + chunk * 8));
      ubase[j] = reinterpret_cast<const unsigned char*>(g.A);
    } else {
      voff[j] = (unsigned)(2 * ((lo ? g.w_plane : 0) + (long)(n0 + row) * g.K + chunk * 8));
      ubase[j] = reinterpret_cast<const unsigned char*>(g.W);
    }
  }
  auto issue_stage = [&](int kt, int buf) {
    unsigned char* base = smem + buf * kStage;
    const size_t ko = (size_t)kt * (BK * 2);  // bytes
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
      // global_load_lds_dwordx4 in its SGPR-base form, written out: the builtin only selects the 64-bit per-lane address
      // form (a v_lshl_add_u64 in front of every issue, even with the base held in SGPRs)
      const unsigned long long sb = reinterpret_cast<unsigned long long>(ubase[j]) + ko;
      const unsigned dst = lds_base + (unsigned)(buf * kStage + (wid + NW * j) * 1024);
      lds_dma16_sgpr(voff[j], sb, dst);
    }
    (void)base;
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
  const int a_off = (wm * (32 * MI) + l31) * 64, b_off = 2 * kAPlane + (wn * NI * 32 + l31) * 64;
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
    // k-tile kt has landed: every wave waits for its own LDS-DMA, the barrier for everybody else's; k-tile kt - 1 has
    // been read by everyone, so its stage may be refilled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nkt) issue_stage(kt + 1, (kt + 1) & 1);
    compute(kt & 1);
  }
  __syncthreads();  // the operand stages are dead: the epilogue reuses them

  // ---- epilogue: each wavefront transposes one 32 x 32 MFMA tile at a time through a private LDS stage and moves
  // 16 bytes per lane (128-byte row segments per 8 lanes, or two 64-byte plane segments per 4)
  // Staging image: 32 rows x 32 floats, unpadded, columns XORed by 4 on rows with bit 2 set.  With that the
  // ds_write_b32 of the accumulator registers (32 consecutive lanes = one row) and the ds_read_b128 of the row
  // segments (served in the non-contiguous 16-lane groups of MI355X_MICROARCH.md, LDS table) are both conflict-free
  // for 4 and for 8 columns per lane; 36-float rows made every read 2-way.
  constexpr int SLD = 32;
  float* const stage = reinterpret_cast<float*>(smem) + wid * (32 * SLD);
  constexpr int CPL = PLANES_OUT ? 8 : 4;    // columns per lane
  constexpr int LPR = 32 / CPL;              // lanes per staged row
  constexpr int RPS = 64 / LPR;              // rows per pass
  const int prow = lane / LPR, c0 = (lane % LPR) * CPL;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + (wn * NI + ni) * 32 + c0;
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
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + (l31 ^ (lh << 2))] = acc[mi][ni][r] * g.descale;
      // the stage is private to this wavefront and LDS executes a wave's operations in order.
      // One division per 32-row slab: rows advance by at most 31 < c_rpb, pos_period (host-checked).
      const int mbase = m0 + wm * (32 * MI) + mi * 32;
      const int mb0 = mbase / g.c_rpb, mt0 = mbase % g.c_rpb;
      const int mp0 = (EPI & kEpiPos) ? mbase % g.pos_period : 0;
#pragma unroll
      for (int p = 0; p < 32 / RPS; ++p) {
        const int row = p * RPS + prow;
        float v[CPL];
#pragma unroll
        for (int e = 0; e < CPL; e += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&stage[row * SLD + ((c0 + e) ^ (((row >> 2) & 1) << 2))]);
          v[e] = t[0], v[e + 1] = t[1], v[e + 2] = t[2], v[e + 3] = t[3];
        }
        if (mbase + row < g.M) {
          int mb = mb0, mt = mt0 + row;
          if (mt >= g.c_rpb) mt -= g.c_rpb, mb += 1;
#pragma unroll
          for (int e = 0; e < CPL; e += 2) {
            v[e] += bias_v[e];
            v[e + 1] += bias_v[e + 1];
            if (EPI & kEpiGelu) {
              const f32x2_t gl = gelu_erf2(f32x2_t{v[e], v[e + 1]});
              v[e] = gl[0];
              v[e + 1] = gl[1];
            }
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
            u32x4 hi, lo;
#pragma unroll
            for (int e = 0; e < CPL; e += 2) {
              unsigned h, l;
              split_f16x2(v[e] * oscale, v[e + 1] * oscale, &h, &l);
              hi[e / 2] = h;
              lo[e / 2] = l;
            }
            *reinterpret_cast<u32x4*>(g.P + o) = hi;
            *reinterpret_cast<u32x4*>(g.P + g.p_plane + o) = lo;
          } else if (EPI & kEpiKvLayout) {
            const long o = (((long)slab * g.kv_batch + mb) * g.kv_heads + head) * (long)g.c_rpb * 64 + (long)mt * 64 + dd;
            // the cache is next read by the decoder, long after L2 / Infinity Cache have turned over: streaming store
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(g.C + o));
          } else {
            const long o = (long)mb * g.c_bs + (long)mt * g.ldc + n;
            f32x4 out = {v[0], v[1], v[2], v[3]};
            if (EPI & kEpiResidual) out += *reinterpret_cast<const f32x4*>(g.R + o);
            *reinterpret_cast<f32x4*>(g.C + o) = out;
            if constexpr (LN) {  // the accumulator registers of this tile are dead: they keep the finished row-major values
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[mi][ni][4 * p + e] = out[e];
            }
          }
        } else if constexpr (LN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][4 * p + e] = 0.0f;  // rows past M: no part in anything
        }
      }
    }
  }
  if constexpr (LN) {
    // LayerNorm of the rows this block has just finished (N == BN = 384: whole rows), written as the planes the
    // next GEMM reads: the separate LayerNorm launch re-read and re-wrote the residual stream (147 MB per launch at 32
    // clips).  acc[mi][ni][4 p + e] now holds row (mi, p * 8 + prow), columns 32 (wn NI + ni) + c0 + e of the finished
    // values.  Two-pass statistics like layernorm_rows_planes: the four wavefront columns of a block row exchange
    // their partial sums through LDS (behind the epilogue's stage area), first of the values, then of the squared
    // deviations.
    float* const part = reinterpret_cast<float*>(smem) + NW * (32 * SLD);  // [BM][4] partial sums, twice
    const int rbase = wm * (32 * MI) + prow;
    float mean[MI][4], rstd[MI][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          float t = 0.0f;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = acc[mi][ni][4 * p + e] - (pass ? mean[mi][p] : 0.0f);
              t += pass ? x * x : x;
            }
          t += __shfl_xor(t, 1, 64);
          t += __shfl_xor(t, 2, 64);
          t += __shfl_xor(t, 4, 64);
          if ((lane & 7) == 0) part[pass * (BM * 4) + (rbase + mi * 32 + p * 8) * 4 + wn] = t;
        }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(&part[pass * (BM * 4) + (rbase + mi * 32 + p * 8) * 4]);
          const float tot = ((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / (float)BN);
          if (pass == 0) {
            mean[mi][p] = tot;
          } else {
            if (g.nonfinite != nullptr && (lane & 7) == 0 && wn == 0 && !(fabsf(mean[mi][p]) <= 3.0e38f && tot <= 3.0e38f)) atomicOr(g.nonfinite, 1);
            rstd[mi][p] = rsqrtf(tot + 1e-5f);
          }
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + (wn * NI + ni) * 32 + c0;
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g.ln_g + n), bb = *reinterpret_cast<const f32x4*>(g.ln_b + n);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int m = m0 + rbase + mi * 32 + p * 8;
          if (m >= g.M) continue;
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[mi][ni][4 * p + e] - mean[mi][p]) * rstd[mi][p] * gg[e] + bb[e];
          const long o = (long)m * BN + n;
          if (g.ln_y32 != nullptr) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(g.ln_y32 + o));
          unsigned h0, l0, h1, l1;
          split_f16x2(y[0] * g.ln_scale, y[1] * g.ln_scale, &h0, &l0);
          split_f16x2(y[2] * g.ln_scale, y[3] * g.ln_scale, &h1, &l1);
          using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
          *reinterpret_cast<u32x2*>(g.ln_P + o) = u32x2{h0, h1};
          *reinterpret_cast<u32x2*>(g.ln_P + g.ln_plane + o) = u32x2{l0, l1};
        }
    }
  }
}

template <int EPI, bool PLANES_OUT, int WN, int NI, int MI = 3, bool LN = false>
void launch_planes_shape(const PlaneGemmDev& g, hipStream_t s) {
  constexpr int BN = WN * NI * 32, BM = 64 * MI;
  const int blocks = ((g.M + BM - 1) / BM) * (g.N / BN);
  constexpr size_t smem = 2 * (2 * BM * BK * 2 + 2 * BN * BK * 2);  // two stages; the epilogue stages fit inside
  static const bool raised = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_tile<EPI, PLANES_OUT, WN, NI, MI, LN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)raised;
  WT_LAUNCH_TIMED((gemm_planes_tile<EPI, PLANES_OUT, WN, NI, MI, LN>), dim3(blocks), dim3(128 * WN), smem, s, g);
}

// returns true when the launch also wrote the LayerNorm planes (g.ln_P set and a 384-column tile chosen)
template <int EPI, bool PLANES_OUT>
bool launch_planes(const PlaneGemmDev& g, int n_cu, hipStream_t s) {
  static const int forced = [] {
    const char* v = getenv("WT_PLANE_TILE");  // measurement knob (tools/gemm_planes_bench.py): 128 / 384 = that tile only
    return v ? atoi(v) : 0;
  }();
  // Tile choice by how the blocks fill the CUs this stream may use (the pipelined encoder stream leaves some CUs
  // to the decoders): 250 wide row tiles are one round on 256 CUs but two on 224, 188 tall ones one round on either.
  // Cost of a candidate = rows x 128-column units x rounds of the CUs; the narrow kernel (two co-resident blocks per
  // CU, which share the CU's throughput) costs ~10 % more per unit, the 256-row tile ~5 % less (fewer staged bytes
  // per FLOP).
  const int cu = n_cu > 0 ? n_cu : 256;
  auto rounds = [&](long tiles) { return (tiles + cu - 1) / cu; };
  const long rt192 = (g.M + 191) / 192, rt256 = (g.M + 255) / 256;
  const bool can_wide = g.N % 384 == 0;
  const double c_wide = can_wide ? 192.0 * 3 * rounds(rt192 * (g.N / 384)) : 1e30;
  const double c_tall = can_wide ? 256.0 * 3 * 0.95 * rounds(rt256 * (g.N / 384)) : 1e30;
  const double c_narrow = 192.0 * 1.1 * rounds(rt192 * (g.N / 128));
  int pick = c_wide <= c_narrow ? (c_tall < c_wide ? 2 : 1) : (c_tall < c_narrow ? 2 : 0);
  if (forced == 128) pick = 0;
  if (forced == 384 && can_wide) pick = 1;
  if (forced == 256 && can_wide) pick = 2;
  if constexpr (!PLANES_OUT && (EPI == (kEpiBias | kEpiResidual) || EPI == (kEpiBias | kEpiGelu | kEpiPos))) {
    if (g.ln_P != nullptr && g.N == 384 && pick != 0) {  // whole rows in one block: LayerNorm fused into the epilogue
      if (pick == 2) {
        launch_planes_shape<EPI, false, 4, 3, 4, true>(g, s);
      } else {
        launch_planes_shape<EPI, false, 4, 3, 3, true>(g, s);
      }
      return true;
    }
  }
  if (pick == 2) {
    launch_planes_shape<EPI, PLANES_OUT, 4, 3, 4>(g, s);
  } else if (pick == 1) {
    launch_planes_shape<EPI, PLANES_OUT, 4, 3, 3>(g, s);
  } else {
    launch_planes_shape<EPI, PLANES_OUT, 2, 2, 3>(g, s);
  }
  return false;
}

}  // namespace

bool launch_gemm_planes(const PlaneGemmArgs& a, int epi, hipStream_t s) {
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
  g.ln_g = a.ln_g; g.ln_b = a.ln_b; g.ln_P = reinterpret_cast<_Float16*>(a.ln_P); g.ln_plane = a.ln_plane;
  g.ln_scale = a.ln_scale; g.ln_y32 = a.ln_y32; g.nonfinite = a.nonfinite;
  if (a.ln_P && (!a.ln_g || !a.ln_b || a.P || a.c_rpb < a.M || a.ldc != a.N || !(a.ln_scale > 0.0f))) {
    throw Error(kErrInvalidArg, "plane GEMM LayerNorm fusion: needs gain and shift, fp32 output, contiguous [M][N] rows");
  }
  const bool planes = a.P != nullptr;
  // shape contract of the kernel (16-byte chunks, whole k-tiles; the epilogue wraps clip / position rows at most once
  // per 32 rows; 8 output columns never straddle a scale segment)
  if (a.N % 128 != 0 || a.K % BK != 0 || a.M < 1 || a.c_rpb < 32 || a.pos_period < (epi & kEpiPos ? 32 : 1) || a.lda % 8 != 0 ||
      a.a_bs % 8 != 0 || a.ldc % 8 != 0 || a.c_bs % 8 != 0 || (planes && (g.seg % 8 != 0 || (a.N + g.seg - 1) / g.seg > 3)) ||
      (!planes && !a.C) || !(a.a_scale > 0.0f) || !(a.w_scale > 0.0f)) {
    throw Error(kErrInvalidArg, "plane GEMM shape outside the kernel contract");
  }
  // the kernel addresses both operands as a uniform base + 32-bit per-lane byte offset (see gemm_planes_tile)
  {
    const long a_span = a.a_plane + (long)((a.M - 1) / a.a_rpb) * a.a_bs + (long)std::min(a.a_rpb, a.M) * a.lda + a.K + 64;
    const long w_span = a.w_plane + (long)a.N * a.K + 64;
    if (a.a_plane < 0 || a.w_plane < 0 || a.a_bs < 0 || 2 * a_span >= (1L << 32) || 2 * w_span >= (1L << 32)) {
      throw Error(kErrInvalidArg, "plane GEMM operand spans more than the 4 GiB its 32-bit offsets reach");
    }
  }
  switch (epi | (planes ? 256 : 0)) {
    case kEpiBias: return launch_planes<kEpiBias, false>(g, a.n_cu, s);
    case kEpiBias | kEpiGelu: return launch_planes<kEpiBias | kEpiGelu, false>(g, a.n_cu, s);  // fp32 for a fall-back consumer
    case kEpiBias | kEpiResidual: return launch_planes<kEpiBias | kEpiResidual, false>(g, a.n_cu, s);
    case kEpiBias | kEpiGelu | kEpiPos: return launch_planes<kEpiBias | kEpiGelu | kEpiPos, false>(g, a.n_cu, s);
    case kEpiBias | kEpiKvLayout: return launch_planes<kEpiBias | kEpiKvLayout, false>(g, a.n_cu, s);
    case kEpiBias | 256: return launch_planes<kEpiBias, true>(g, a.n_cu, s);
    case kEpiBias | kEpiGelu | 256: return launch_planes<kEpiBias | kEpiGelu, true>(g, a.n_cu, s);
    default: throw Error(kErrInvalidArg, "unsupported plane GEMM epilogue combination");
  }
}

}  // namespace wt

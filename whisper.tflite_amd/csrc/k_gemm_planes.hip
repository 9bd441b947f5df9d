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
#include <cstring>
#include <type_traits>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {

// schedule of the 384-column tiles: 0 = both wavefronts of a SIMD in step (gemm_planes_tile), 1 = ping-pong groups
// (gemm_planes_pp), 2 = ping-pong groups on 16 x 16 x 32 MFMAs (gemm_planes_pp16, persistent where a CU runs several
// plane-output tiles), 4 = the same, one tile per block everywhere.  WT_PLANE_GEMM_MODE / wt_dbg_set_plane_gemm_mode are measurement knobs (tools/gemm_planes_bench.py).
static int g_plane_gemm_mode = -1;
int plane_gemm_mode() {
  if (g_plane_gemm_mode < 0) {
    const char* v = getenv("WT_PLANE_GEMM_MODE");
    g_plane_gemm_mode = v ? atoi(v) : 2;
  }
  return g_plane_gemm_mode;
}
void set_plane_gemm_mode(int m) { g_plane_gemm_mode = m; }

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
  const _Float16* W;   // both planes in the blocked LDS-image layout of split_weight_planes()
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

// The four 16-byte chunks of a staged 64-byte row are XOR-swizzled by a function of the row: chunk c of row r sits in
// slot c ^ chunk_swizzle(r).  (-(r >> 2)) & 3 = 0, 3, 2, 1 for rows 0-3, 4-7, 8-11, 12-15 (period 16) makes the
// ds_read_b128 fragment reads conflict-free for the 32 x 32 x 16 pattern (lane = row l & 31, chunk pair l >> 5) AND for the
// 16 x 16 x 32 pattern (lane = row l & 15, chunk l >> 4) under the 16-lane service groups of MI355X_MICROARCH.md; round
// 3's (r >> 2) & 3 served the first pattern only.
__host__ __device__ __forceinline__ constexpr int chunk_swizzle(int row) { return (0 - (row >> 2)) & 3; }

// ---- LDS-DMA source addresses of a stage, shared by the tile kernels.  Instruction q of a stage copies 16 rows x 64 B
// (one plane of a 32-deep k-tile): q in [0, QA) A hi, [QA, 2 QA) A lo, [2 QA, 2 QA + QW) W hi, then W lo; its LDS
// destination is the stage base + q * 1024 (lane-linear).  Wave w issues q = w, w + NW, ...  Lane i fills LDS slot
// (row i >> 2, chunk i & 3) of its instruction with the chunk (i & 3) ^ chunk_swizzle(row) of that row: the XOR swizzle that
// makes the fragment reads conflict-free lives on the SOURCE side.
//   A (activations, row-major planes): 16 segments of 64 B per instruction, the swizzle in the per-lane offset.
//   W (weights): stored by split_weight_planes() as the LDS image itself — [N / 16][K / 32][hi | lo][16 rows][32 k] with
//   the chunks already swizzled — so an instruction reads 1 KiB CONTIGUOUS bytes (lane i its i-th 16).  Round 4, from
//   tools/dma_probe.hip: the staging of a k-tile alone takes 2.05 us with 64-byte segments, 1.48 us with contiguous
//   KiBs (36 against 50 GB/s per CU), and the 9 issues of a wave 1070 against 480 cycles; W is two thirds of a stage.
// Addresses are (uniform base) + (32-bit per-lane byte offset): the LDS-DMA takes its base from SGPRs that SALU
// instructions advance per k-tile (by kstep[j] bytes), and its issue needs NO vector instruction.
template <int QA, int QW, int NW, int QPW>
__device__ __forceinline__ void setup_stage_dma(const PlaneGemmDev& g, int m0, int n0, int wid, int lane, unsigned (&voff)[QPW],
                                                const unsigned char* (&ubase)[QPW], unsigned (&kstep)[QPW]) {
  const int srow = lane >> 2;
  const int nkt = g.K / BK;
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    const int q = wid + NW * j;
    if (q < 2 * QA) {
      const bool lo = q >= QA;
      const int row = 16 * (lo ? q - QA : q) + srow;
      const int chunk = (lane & 3) ^ chunk_swizzle(row);
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
      voff[j] = (unsigned)(2 * ((lo ? g.a_plane : 0) + (long)(m / g.a_rpb) * g.a_bs + (long)(m % g.a_rpb) * g.lda + chunk * 8));
      ubase[j] = reinterpret_cast<const unsigned char*>(g.A);
      kstep[j] = BK * 2;
    } else {
      const int wq = q - 2 * QA;
      const int rg = wq % QW, plane = wq / QW;
      voff[j] = (unsigned)(lane * 16);
      ubase[j] = reinterpret_cast<const unsigned char*>(g.W) + ((size_t)(n0 / 16 + rg) * nkt * 2 + plane) * 1024;
      kstep[j] = 2048;
    }
  }
}

// ---- accumulators of a wave tile as the epilogue sees them: MI x NI blocks of 32 x 32 outputs, 16 floats per lane and
// block.  stage_write() puts a block into the wave's private LDS stage (32 rows x 32 floats, columns XORed by 20 — bits 2
// and 4 — on rows with bit 2 set: conflict-free for the ds_write_b32 of BOTH accumulator layouts and for the row reads of 4
// and 8 columns per lane; round 3's XOR by 4 served the 32 x 32 layout only, the 16 x 16 one wrote 2-way); get / set address the 16 floats as plain storage (the LayerNorm fusion parks finished values there).
template <int MI_, int NI_>
struct Acc32 {  // v_mfma_f32_32x32x16: lane (l31, lh) holds column l31, rows (r & 3) + 8 (r >> 2) + 4 lh
  static constexpr int MI = MI_, NI = NI_;
  f32x16 t[MI_][NI_];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < MI_; ++i)
#pragma unroll
      for (int j = 0; j < NI_; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[i][j][r] = 0.0f;
  }
  __device__ __forceinline__ float get(int mi, int ni, int i) const { return t[mi][ni][i]; }
  __device__ __forceinline__ void set(int mi, int ni, int i, float v) { t[mi][ni][i] = v; }
  __device__ __forceinline__ void stage_write(int mi, int ni, float* stage, int lane, float scale) const {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + (l31 ^ (lh * 20))] = t[mi][ni][r] * scale;  // row bit 2 = lh
  }
};
template <int MI_, int NI_>
struct Acc16 {  // v_mfma_f32_16x16x32: 2 x 2 tiles per block; lane (c = l & 15, q = l >> 4) holds column c, rows 4 q + r
  static constexpr int MI = MI_, NI = NI_;
  f32x4 t[2 * MI_][2 * NI_];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 2 * MI_; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NI_; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[i][j][r] = 0.0f;
  }
  __device__ __forceinline__ float get(int mi, int ni, int i) const { return t[2 * mi + (i >> 3)][2 * ni + ((i >> 2) & 1)][i & 3]; }
  __device__ __forceinline__ void set(int mi, int ni, int i, float v) { t[2 * mi + (i >> 3)][2 * ni + ((i >> 2) & 1)][i & 3] = v; }
  __device__ __forceinline__ void stage_write(int mi, int ni, float* stage, int lane, float scale) const {
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)  // row 16 a + 4 q + r: its bit 2 is q & 1
          stage[(16 * a + 4 * q + r) * 32 + ((16 * b + c) ^ ((q & 1) * 20))] = t[2 * mi + a][2 * ni + b][r] * scale;
  }
};

// ---- epilogue shared by the tile kernels: each wavefront transposes one 32 x 32 MFMA tile at a time through a
// private LDS stage and moves 16 bytes per lane (128-byte row segments per 8 lanes, or two 64-byte plane segments
// per 4).  `smem` is the block's dynamic LDS, dead as operand staging by the time this runs.
struct NoEpilogueHook {
  static constexpr bool active = false;
  __device__ __forceinline__ void operator()() const {}
};
template <class F>
struct ActiveEpilogueHook {
  static constexpr bool active = true;
  F& f;
  __device__ __forceinline__ void operator()() const { f(); }
};
// `hook` (persistent kernel): called once after the tile's bias values are IN registers and before anything else touches
// memory — the point where the next tile's first LDS-DMA may be issued without an epilogue load queueing behind it.
template <int EPI, bool PLANES_OUT, int WN, bool LN, class Acc, class Hook = NoEpilogueHook>
__device__ __forceinline__ void planes_epilogue(const PlaneGemmDev& g, Acc& acc, unsigned char* smem, int m0, int n0, int wid, int wm,
                                                int wn, int lane, Hook hook = Hook{}) {
  constexpr int MI = Acc::MI, NI = Acc::NI;
  constexpr int BN = WN * NI * 32, NW = 2 * WN, BM = 64 * MI;
  __syncthreads();  // the operand stages are dead: the epilogue reuses them

  // ---- epilogue: each wavefront transposes one 32 x 32 MFMA tile at a time through a private LDS stage and moves
  // 16 bytes per lane (128-byte row segments per 8 lanes, or two 64-byte plane segments per 4)
  // Staging image: 32 rows x 32 floats, unpadded, columns XORed by 20 on rows with bit 2 set.  With that the
  // ds_write_b32 of the accumulator registers (32 consecutive lanes = one row) and the ds_read_b128 of the row
  // segments (served in the non-contiguous 16-lane groups of MI355X_MICROARCH.md, LDS table) are both conflict-free
  // for 4 and for 8 columns per lane; 36-float rows made every read 2-way.
  constexpr int SLD = 32;
  float* const stage = reinterpret_cast<float*>(smem) + wid * (32 * SLD);
  constexpr int CPL = PLANES_OUT ? 8 : 4;    // columns per lane
  constexpr int LPR = 32 / CPL;              // lanes per staged row
  constexpr int RPS = 64 / LPR;              // rows per pass
  const int prow = lane / LPR, c0 = (lane % LPR) * CPL;
  float bias_all[NI][CPL];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int e = 0; e < CPL; ++e) bias_all[ni][e] = (EPI & kEpiBias) ? g.bias[n0 + (wn * NI + ni) * 32 + c0 + e] : 0.0f;
  if constexpr (Hook::active) {
    // the values are "used" here, so hipcc's wait for the loads sits here too — in front of the hook's LDS-DMA, which
    // the same wait would otherwise have to sit out (vmcnt retires in issue order, and the DMA is inline asm hipcc
    // does not count)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < CPL; e += 4)
        asm volatile("" : "+v"(bias_all[ni][e]), "+v"(bias_all[ni][e + 1]), "+v"(bias_all[ni][e + 2]), "+v"(bias_all[ni][e + 3]));
    hook();
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + (wn * NI + ni) * 32 + c0;
    float bias_v[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) bias_v[e] = bias_all[ni][e];
    const float oscale = PLANES_OUT ? g.out_scale[n / g.seg] : 1.0f;  // CPL consecutive columns never straddle a segment
    // kEpiKvLayout: the column decomposition does not depend on the row
    const int slab = (EPI & kEpiKvLayout) ? n / g.kv_dmodel : 0, rem = (EPI & kEpiKvLayout) ? n % g.kv_dmodel : 0;
    const int head = rem >> 6, dd = rem & 63;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      acc.stage_write(mi, ni, stage, lane, g.descale);
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
          const f32x4 t = *reinterpret_cast<const f32x4*>(&stage[row * SLD + ((c0 + e) ^ (((row >> 2) & 1) * 20))]);
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
          } else if constexpr ((EPI & kEpiPower) != 0) {
            // columns (2 k, 2 k + 1) = (re, im) of bin k: the power spectrum, two bins per lane, row stride ldc
            using f32x2 = __attribute__((ext_vector_type(2))) float;
            const long o = (long)mb * g.c_bs + (long)mt * g.ldc + (n >> 1);
            *reinterpret_cast<f32x2*>(g.C + o) = f32x2{v[0] * v[0] + v[1] * v[1], v[2] * v[2] + v[3] * v[3]};
          } else {
            const long o = (long)mb * g.c_bs + (long)mt * g.ldc + n;
            f32x4 out = {v[0], v[1], v[2], v[3]};
            if (EPI & kEpiResidual) out += *reinterpret_cast<const f32x4*>(g.R + o);
            *reinterpret_cast<f32x4*>(g.C + o) = out;
            if constexpr (LN) {  // the accumulator registers of this tile are dead: they keep the finished row-major values
#pragma unroll
              for (int e = 0; e < 4; ++e) acc.set(mi, ni, 4 * p + e, out[e]);
            }
          }
        } else if constexpr (LN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc.set(mi, ni, 4 * p + e, 0.0f);  // rows past M: no part in anything
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
              const float x = acc.get(mi, ni, 4 * p + e) - (pass ? mean[mi][p] : 0.0f);
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
          for (int e = 0; e < 4; ++e) y[e] = (acc.get(mi, ni, 4 * p + e) - mean[mi][p]) * rstd[mi][p] * gg[e] + bb[e];
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

  unsigned voff[QPW], kstep[QPW];
  const unsigned char* ubase[QPW];
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  setup_stage_dma<QA, QW, NW, QPW>(g, m0, n0, wid, lane, voff, ubase, kstep);
  auto issue_stage = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
      // global_load_lds_dwordx4 in its SGPR-base form, written out: the builtin only selects the 64-bit per-lane address
      // form (a v_lshl_add_u64 in front of every issue, even with the base held in SGPRs)
      const unsigned long long sb = reinterpret_cast<unsigned long long>(ubase[j]) + (size_t)kt * kstep[j];
      const unsigned dst = lds_base + (unsigned)(buf * kStage + (wid + NW * j) * 1024);
      lds_dma16_sgpr(voff[j], sb, dst);
    }
  };

  Acc32<MI, NI> accs;
  accs.zero();
  auto& acc = accs.t;

  // fragment addresses: row = tile base (a multiple of 32) + l31, so the swizzle term depends on the lane only
  const int swz = chunk_swizzle(l31);
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
  planes_epilogue<EPI, PLANES_OUT, WN, LN>(g, accs, smem, m0, n0, wid, wm, wn, lane);
}

// Ping-pong form of the 384-column tiles (round 4).  Same tile, same staging image, same k order and product order
// (bit-identical results), another schedule: the block's eight wavefronts are two GROUPS of four — waves w and w + 4
// share a SIMD — that run the same program ONE PHASE apart.  A phase is either "read the fragments of a 16-deep k-step
// (12 ds_read_b128) and issue the wave's share of the next-but-one k-tile's LDS-DMA" or "27 MFMAs"; a raw s_barrier
// closes every phase, so while one wavefront of a SIMD streams its MFMAs the other one does its LDS reads and DMA
// issues.  In gemm_planes_tile both wavefronts of a SIMD did the same thing at the same time: the matrix pipe idled
// while both issued DMA (60-180 cycles per instruction, 9-10 per wave and k-tile) and fragment reads, then both
// contended for it (profiles/r03_gemm_mainloop_ablation.txt: DMA + MFMA cost nearly their sum).
//   phase p:     G0  L(0) C(0) L(1) C(1) L(2) ...        L(s) = fragment reads of k-step s (+ DMA issue on even s)
//                G1       L(0) C(0) L(1) C(1) ...        C(s) = the 27 MFMAs of k-step s
// Stage kt & 1 is read in L(2 kt) and L(2 kt + 1); its last reader is G1's L(2 kt + 1) in phase 4 kt + 3, so k-tile
// kt + 2 is issued into it from phase 4 kt + 4 on: by every wave in its own L(2 kt + 2).  It is first read by G0's
// L(2 kt + 4) in phase 4 kt + 8: G0 waves certify their DMA (vmcnt(0)) at the end of C(2 kt + 3), G1 waves at the end of
// L(2 kt + 3), both phase 4 kt + 7.
#ifdef WT_PP_STAMPS  // diagnostic build (tools/gemm_phase_probe.hip): where a wavefront's cycles of the main loop go
__device__ long long g_pp_stamps[1024 * 8 * 9];  // per wave: 4 phase sums, loop, prologue, epilogue, kernel, realtime ticks
#define PP_STAMP(i)                                  \
  do {                                               \
    const long long t_ = __builtin_readcyclecounter(); \
    st_[i] += t_ - tl_;                              \
    tl_ = t_;                                        \
  } while (0)
#else
#define PP_STAMP(i) \
  do {              \
  } while (0)
#endif

template <int EPI, bool PLANES_OUT, int MI, bool LN = false>
__global__ __launch_bounds__(512, 2) void gemm_planes_pp(PlaneGemmDev g) {
  constexpr int WN = 4, NI = 3;
  static_assert(!LN || !PLANES_OUT, "LayerNorm fusion: fp32 output");
  constexpr int BN = WN * NI * 32, NW = 2 * WN, BM = 64 * MI;
  constexpr int kAPlane = BM * BK * 2, kWPlane = BN * BK * 2;
  constexpr int kStage = 2 * kAPlane + 2 * kWPlane;
  constexpr int QA = BM / 16, QW = BN / 16;
  constexpr int QT = 2 * QA + 2 * QW, QPW = QT / NW;
  static_assert(QT % NW == 0, "whole instructions per wavefront");
#ifdef WT_PP_STAMPS
  const long long t_kernel_ = __builtin_readcyclecounter();
  const long long r_kernel_ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_tiles = g.N / BN;
  const int m0 = (logical / n_tiles) * BM;
  const int n0 = (logical % n_tiles) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;  // wm = the wave's group
  const int l31 = lane & 31, lh = lane >> 5;

  unsigned voff[QPW], kstep[QPW];
  const unsigned char* ubase[QPW];
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  setup_stage_dma<QA, QW, NW, QPW>(g, m0, n0, wid, lane, voff, ubase, kstep);
  auto issue_stage = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
      const unsigned long long sb = reinterpret_cast<unsigned long long>(ubase[j]) + (size_t)kt * kstep[j];
      const unsigned dst = lds_base + (unsigned)(buf * kStage + (wid + NW * j) * 1024);
      lds_dma16_sgpr(voff[j], sb, dst);
    }
  };

  Acc32<MI, NI> accs;
  accs.zero();
  auto& acc = accs.t;

  const int swz = chunk_swizzle(l31);
  const int a_off = (wm * (32 * MI) + l31) * 64, b_off = 2 * kAPlane + (wn * NI * 32 + l31) * 64;
  half8 ah[MI], al[MI], bh[NI], bl[NI];
  auto load_frags = [&](int buf, int ks) {
    const unsigned char* base = smem + buf * kStage;
    const int slot = ((ks * 2 + lh) ^ swz) * 16;
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
  };
  auto compute = [&]() {  // smallest products first, as gemm_planes_tile
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
  };
  // a phase boundary: nothing (the register-only MFMAs included) is scheduled across it
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto reads_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto dma_done = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  const int nkt = g.K / BK;
  issue_stage(0, 0);
  if (nkt > 1) {
    issue_stage(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPW) : "memory");  // k-tile 0 has landed, k-tile 1 stays in flight
  } else {
    dma_done();
  }
  phase_end();
  if (wm == 1) phase_end();  // G1 runs one phase behind
  load_frags(0, 0);
  reads_done();
  phase_end();
#ifdef WT_PP_STAMPS
  long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tl_ = __builtin_readcyclecounter();
  const long long t_begin_ = tl_;
  st_[5] = tl_ - t_kernel_;
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    compute();  // C(2 kt)
    phase_end();
    PP_STAMP(0);
    load_frags(kt & 1, 1);  // L(2 kt + 1)
    reads_done();
    if (wm == 1) dma_done();
    phase_end();
    PP_STAMP(1);
    compute();  // C(2 kt + 1)
    if (wm == 0) dma_done();
    phase_end();
    PP_STAMP(2);
    if (kt + 1 < nkt) {  // L(2 kt + 2)
      if (kt + 2 < nkt) issue_stage(kt + 2, kt & 1);
      load_frags((kt + 1) & 1, 0);
      reads_done();
    }
    phase_end();
    PP_STAMP(3);
  }
  if (wm == 0) phase_end();
#ifdef WT_PP_STAMPS
  const long long t_loop_end_ = __builtin_readcyclecounter();
  st_[4] = t_loop_end_ - t_begin_;
#endif
  planes_epilogue<EPI, PLANES_OUT, WN, LN>(g, accs, smem, m0, n0, wid, wm, wn, lane);
#ifdef WT_PP_STAMPS
  {
    const long long t_end_ = __builtin_readcyclecounter();
    st_[6] = t_end_ - t_loop_end_;
    st_[7] = t_end_ - t_kernel_;
    st_[8] = (long long)__builtin_amdgcn_s_memrealtime() - r_kernel_;
    if (lane == 0 && blockIdx.x < 1024)
      for (int i = 0; i < 9; ++i) g_pp_stamps[(blockIdx.x * 8 + wid) * 9 + i] = st_[i];
  }
#endif
}

// The same ping-pong schedule on v_mfma_f32_16x16x32_f16 (round 4; tile 192 x 384, wave tile 96 x 96 = 6 x 6 MFMA tiles).
// Why another MFMA shape: this kernel is POWER-bound — the chip holds 1.25-1.4 GHz under it (s_memtime over s_memrealtime,
// tools/gemm_phase_probe.hip) while the loop is within 15 % of its MFMA cycle count — and the 16 x 16 x 32 form does the
// same arithmetic on less energy: a register-only loop of the three plane products on random planes sustains 1855 TF/s
// at 1.79 GHz against 1590 TF/s at 1.53 GHz for 32 x 32 x 16 (tools/mfma_shape_probe.hip, profiles/r04_mfma_shape.txt).
// A k-tile is two half-tiles of 54 MFMAs (864 cycles, as before): rows 0-47 of the wave tile, then rows 48-95.
//   L_A(kt): 12 reads of the W fragments (kept for both halves) + 6 of the first three A tiles (+ the DMA issue of k-tile
//   kt + 1's successor);  C_A: 54 MFMAs;  L_B: 6 reads of the other three A tiles;  C_B: 54 MFMAs.  Stage lifetimes and
//   the vmcnt certification are those of gemm_planes_pp with L_A / L_B in the place of L(2 kt) / L(2 kt + 1).
// A 16 x 16 x 32 fragment is (row l & 15, chunk l >> 4): one ds_read_b128 covers a whole 16-row x 64-byte block, and
// chunk_swizzle() keeps its 16-lane service groups on distinct banks.  A 32-deep sum per MFMA instead of 16: results
// differ from the 32 x 32 x 16 kernels in the last bits (both within the fp32 error budget of tests/test_gpu_kernels.py).
template <int EPI, bool PLANES_OUT, bool LN = false>
__global__ __launch_bounds__(512, 2) void gemm_planes_pp16(PlaneGemmDev g) {
  constexpr int WN = 4, NI = 3, MI = 3;
  static_assert(!LN || !PLANES_OUT, "LayerNorm fusion: fp32 output");
  constexpr int BN = WN * NI * 32, NW = 2 * WN, BM = 64 * MI;
  constexpr int kAPlane = BM * BK * 2, kWPlane = BN * BK * 2;
  constexpr int kStage = 2 * kAPlane + 2 * kWPlane;
  constexpr int QA = BM / 16, QW = BN / 16;
  constexpr int QT = 2 * QA + 2 * QW, QPW = QT / NW;
  static_assert(QT % NW == 0, "whole instructions per wavefront");
#ifdef WT_PP_STAMPS
  const long long t_kernel_ = __builtin_readcyclecounter();
  const long long r_kernel_ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_tiles = g.N / BN;
  const int m0 = (logical / n_tiles) * BM;
  const int n0 = (logical % n_tiles) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;  // wm = the wave's group

  unsigned voff[QPW], kstep[QPW];
  const unsigned char* ubase[QPW];
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  setup_stage_dma<QA, QW, NW, QPW>(g, m0, n0, wid, lane, voff, ubase, kstep);
  auto issue_stage = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
      const unsigned long long sb = reinterpret_cast<unsigned long long>(ubase[j]) + (size_t)kt * kstep[j];
      const unsigned dst = lds_base + (unsigned)(buf * kStage + (wid + NW * j) * 1024);
      lds_dma16_sgpr(voff[j], sb, dst);
    }
  };

  Acc16<MI, NI> accs;
  accs.zero();
  auto& acc = accs.t;

  const int fc = lane & 15, fq = lane >> 4;
  const int frag = fc * 64 + ((fq ^ chunk_swizzle(fc)) * 16);  // byte offset of the lane's 16 bytes inside a 16-row block
  const int a_off = wm * (32 * MI) * 64 + frag, b_off = 2 * kAPlane + wn * (NI * 32) * 64 + frag;
  half8 ah[3], al[3], bh[6], bl[6];
  auto load_b = [&](int buf) {
    const unsigned char* base = smem + buf * kStage + b_off;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      bh[j] = *reinterpret_cast<const half8*>(base + j * 1024);
      bl[j] = *reinterpret_cast<const half8*>(base + kWPlane + j * 1024);
    }
  };
  auto load_a = [&](int buf, int half) {
    const unsigned char* base = smem + buf * kStage + a_off + half * 3 * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ah[i] = *reinterpret_cast<const half8*>(base + i * 1024);
      al[i] = *reinterpret_cast<const half8*>(base + kAPlane + i * 1024);
    }
  };
  auto compute = [&](auto half_c) {  // smallest products first
    constexpr int H = decltype(half_c)::value;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[3 * H + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[3 * H + i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[3 * H + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[3 * H + i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[3 * H + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[3 * H + i][j], 0, 0, 0);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto reads_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto dma_done = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  const int nkt = g.K / BK;
  issue_stage(0, 0);
  if (nkt > 1) {
    issue_stage(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPW) : "memory");
  } else {
    dma_done();
  }
  phase_end();
  if (wm == 1) phase_end();  // G1 runs one phase behind
  load_b(0);
  load_a(0, 0);
  reads_done();
  phase_end();
#ifdef WT_PP_STAMPS
  long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tl_ = __builtin_readcyclecounter();
  const long long t_begin_ = tl_;
  st_[5] = tl_ - t_kernel_;
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    compute(H0{});  // C_A(kt)
    phase_end();
    PP_STAMP(0);
    load_a(kt & 1, 1);  // L_B(kt)
    reads_done();
    if (wm == 1) dma_done();
    phase_end();
    PP_STAMP(1);
    compute(H1{});  // C_B(kt)
    if (wm == 0) dma_done();
    phase_end();
    PP_STAMP(2);
    if (kt + 1 < nkt) {  // L_A(kt + 1)
      if (kt + 2 < nkt) issue_stage(kt + 2, kt & 1);
      load_b((kt + 1) & 1);
      load_a((kt + 1) & 1, 0);
      reads_done();
    }
    phase_end();
    PP_STAMP(3);
  }
  if (wm == 0) phase_end();
#ifdef WT_PP_STAMPS
  const long long t_loop_end_ = __builtin_readcyclecounter();
  st_[4] = t_loop_end_ - t_begin_;
#endif
  planes_epilogue<EPI, PLANES_OUT, WN, LN>(g, accs, smem, m0, n0, wid, wm, wn, lane);
#ifdef WT_PP_STAMPS
  {
    const long long t_end_ = __builtin_readcyclecounter();
    st_[6] = t_end_ - t_loop_end_;
    st_[7] = t_end_ - t_kernel_;
    st_[8] = (long long)__builtin_amdgcn_s_memrealtime() - r_kernel_;
    if (lane == 0 && blockIdx.x < 1024)
      for (int i = 0; i < 9; ++i) g_pp_stamps[(blockIdx.x * 8 + wid) * 9 + i] = st_[i];
  }
#endif
}

// gemm_planes_pp16 as a PERSISTENT launch for the shapes a CU runs several tiles of (qkv, fc1: plane output, bias / GELU):
// one block per CU walks tiles bid, bid + grid, ... and issues the NEXT tile's first k-tile from the head of the current
// tile's epilogue (into the stage the epilogue's LDS staging does not touch), the second right behind the epilogue.  A
// tile of these shapes spent 9 k of its 75 k cycles waiting for its first k-tile (tools/gemm_phase_probe.hip).
// Stages: k-tile kt lives in stage (kt + 1) & 1, so that a tile's first k-tile is in stage 1 — the epilogue stages its
// 32 x 32 blocks at the bottom of stage 0.  The next tile's first k-tile is certified behind the epilogue with a counted
// vmcnt: the epilogue's 36 plane stores per wave are younger (a tile with rows past M may skip stores: it waits for all).
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_planes_pp16_persist(PlaneGemmDev g, int total_tiles) {
  constexpr int WN = 4, NI = 3, MI = 3;
  constexpr bool PLANES_OUT = true;
  static_assert((EPI & ~(kEpiBias | kEpiGelu)) == 0, "bias / GELU epilogues with plane output");
  constexpr int BN = WN * NI * 32, NW = 2 * WN, BM = 64 * MI;
  constexpr int kAPlane = BM * BK * 2, kWPlane = BN * BK * 2;
  constexpr int kStage = 2 * kAPlane + 2 * kWPlane;
  constexpr int QA = BM / 16, QW = BN / 16;
  constexpr int QT = 2 * QA + 2 * QW, QPW = QT / NW;
  constexpr int kEpiStores = MI * NI * 4;  // per wave: 9 blocks x 2 passes x (hi, lo)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int grid = gridDim.x, bid = blockIdx.x;
  const int q8 = total_tiles >> 3, r8 = total_tiles & 7;
  const int n_tiles = g.N / BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  unsigned voff[QPW], kstep[QPW];
  const unsigned char* ubase[QPW];
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  int m0 = 0, n0 = 0;
  // virtual block id -> tile: the XCD-aware bijective remap of the one-tile-per-block kernels over all tiles (grid is a
  // multiple of 8, so a block's tiles keep its XCD)
  auto set_tile = [&](int vid) {
    const int xcd = vid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vid >> 3);
    m0 = (logical / n_tiles) * BM;
    n0 = (logical % n_tiles) * BN;
    setup_stage_dma<QA, QW, NW, QPW>(g, m0, n0, wid, lane, voff, ubase, kstep);
  };
  auto issue_stage = [&](int kt) {
    const int buf = (kt + 1) & 1;
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
      const unsigned long long sb = reinterpret_cast<unsigned long long>(ubase[j]) + (size_t)kt * kstep[j];
      lds_dma16_sgpr(voff[j], sb, lds_base + (unsigned)(buf * kStage + (wid + NW * j) * 1024));
    }
  };

  Acc16<MI, NI> accs;
  auto& acc = accs.t;
  const int fc = lane & 15, fq = lane >> 4;
  const int frag = fc * 64 + ((fq ^ chunk_swizzle(fc)) * 16);
  const int a_off = wm * (32 * MI) * 64 + frag, b_off = 2 * kAPlane + wn * (NI * 32) * 64 + frag;
  half8 ah[3], al[3], bh[6], bl[6];
  auto load_b = [&](int kt) {
    const unsigned char* base = smem + ((kt + 1) & 1) * kStage + b_off;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      bh[j] = *reinterpret_cast<const half8*>(base + j * 1024);
      bl[j] = *reinterpret_cast<const half8*>(base + kWPlane + j * 1024);
    }
  };
  auto load_a = [&](int kt, int half) {
    const unsigned char* base = smem + ((kt + 1) & 1) * kStage + a_off + half * 3 * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ah[i] = *reinterpret_cast<const half8*>(base + i * 1024);
      al[i] = *reinterpret_cast<const half8*>(base + kAPlane + i * 1024);
    }
  };
  auto compute = [&](auto half_c) {
    constexpr int H = decltype(half_c)::value;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[3 * H + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[3 * H + i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[3 * H + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[3 * H + i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[3 * H + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[3 * H + i][j], 0, 0, 0);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto reads_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto dma_done = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  const int nkt = g.K / BK;  // even and >= 4 (launcher)
  int vid = bid;
  set_tile(vid);
  issue_stage(0);
  issue_stage(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPW) : "memory");
  phase_end();
  for (;;) {
    if (wm == 1) phase_end();  // G1 runs one phase behind
    accs.zero();
    load_b(0);
    load_a(0, 0);
    reads_done();
    phase_end();
    for (int kt = 0; kt < nkt; ++kt) {
      compute(H0{});
      phase_end();
      load_a(kt, 1);
      reads_done();
      if (wm == 1) dma_done();
      phase_end();
      compute(H1{});
      if (wm == 0) dma_done();
      phase_end();
      if (kt + 1 < nkt) {
        if (kt + 2 < nkt) issue_stage(kt + 2);
        load_b(kt + 1);
        load_a(kt + 1, 0);
        reads_done();
      }
      phase_end();
    }
    if (wm == 0) phase_end();
    const int m_cur = m0, n_cur = n0;
    const int next = vid + grid;
    const bool has_next = next < total_tiles;
    const bool whole = m_cur + BM <= g.M;
    auto prefetch = [&]() {
      if (has_next) {
        set_tile(next);
        issue_stage(0);
      }
    };
    planes_epilogue<EPI, PLANES_OUT, WN, false>(g, accs, smem, m_cur, n_cur, wid, wm, wn, lane, ActiveEpilogueHook<decltype(prefetch)>{prefetch});
    if (!has_next) break;
    if (whole) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEpiStores) : "memory");
    } else {
      dma_done();
    }
    phase_end();  // every wave is done with the epilogue's LDS; the next tile's first k-tile is visible
    issue_stage(1);
    vid = next;
  }
}

template <int EPI, bool PLANES_OUT, int WN, int NI, int MI = 3, bool LN = false>
void launch_planes_shape(const PlaneGemmDev& g, hipStream_t s, int n_cu = 0) {
  constexpr int BN = WN * NI * 32, BM = 64 * MI;
  const int blocks = ((g.M + BM - 1) / BM) * (g.N / BN);
  constexpr size_t smem = 2 * (2 * BM * BK * 2 + 2 * BN * BK * 2);  // two stages; the epilogue stages fit inside
  static const bool raised = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_tile<EPI, PLANES_OUT, WN, NI, MI, LN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)raised;
  if constexpr (WN == 4 && NI == 3 && MI == 3) {  // (the 256-row tile's fragments do not fit beside its accumulators)
    if constexpr (PLANES_OUT && !LN && (EPI & ~(kEpiBias | kEpiGelu)) == 0) {
      // several tiles per CU: one persistent block per CU that prefetches across tiles (WT_PLANE_GEMM_MODE=4 switches it off)
      const int cu = n_cu > 0 ? n_cu : 256;
      if (plane_gemm_mode() == 2 && cu % 8 == 0 && blocks >= 2 * cu && (g.K / BK) % 2 == 0 && g.K / BK >= 4) {
        static const bool raised_p = [] {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_pp16_persist<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024);
          return true;
        }();
        (void)raised_p;
        WT_LAUNCH_TIMED((gemm_planes_pp16_persist<EPI>), dim3(cu), dim3(512), smem, s, g, blocks);
        return;
      }
    }
    if (plane_gemm_mode() == 2 || plane_gemm_mode() == 4) {
      static const bool raised_pp16 = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_pp16<EPI, PLANES_OUT, LN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)raised_pp16;
      WT_LAUNCH_TIMED((gemm_planes_pp16<EPI, PLANES_OUT, LN>), dim3(blocks), dim3(512), smem, s, g);
      return;
    }
    if (plane_gemm_mode() == 1) {
      static const bool raised_pp = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_pp<EPI, PLANES_OUT, MI, LN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)raised_pp;
      WT_LAUNCH_TIMED((gemm_planes_pp<EPI, PLANES_OUT, MI, LN>), dim3(blocks), dim3(512), smem, s, g);
      return;
    }
  }
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
  // CU, which share the CU's throughput) costs ~10 % more per unit.
  const int cu = n_cu > 0 ? n_cu : 256;
  auto rounds = [&](long tiles) { return (tiles + cu - 1) / cu; };
  const long rt192 = (g.M + 191) / 192, rt256 = (g.M + 255) / 256;
  const bool can_wide = g.N % 384 == 0;
  const double c_wide = can_wide ? 192.0 * 3 * rounds(rt192 * (g.N / 384)) : 1e30;
  // (round 4: the 192-row tile runs the ping-pong 16 x 16 x 32 kernel, the 256-row one the older in-step kernel — 8 % more per
  // unit of work instead of 5 % less: fc1 236 us on 4 rounds of 192-row tiles against 260 us on 3 rounds of 256-row tiles)
  const double c_tall = can_wide ? 256.0 * 3 * 1.08 * rounds(rt256 * (g.N / 384)) : 1e30;
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
    launch_planes_shape<EPI, PLANES_OUT, 4, 3, 3>(g, s, n_cu);
  } else {
    launch_planes_shape<EPI, PLANES_OUT, 2, 2, 3>(g, s);
  }
  return false;
}

}  // namespace

// W [N][K] fp32 -> two fp16 planes (hi = fp16(w * scale), lo = fp16(w * scale - hi)) in the layout the plane GEMM's
// LDS-DMA stages without rearranging (k_gemm_planes.hip, setup_stage_dma): [N / 16][Kpad / 32][hi | lo][16 rows][32 k],
// 1 KiB per (row group, k-tile, plane) — exactly the LDS image of one global_load_lds_dwordx4, the four 16-byte chunks of
// a row XOR-swizzled by (row >> 2) & 3 as the fragment reads expect.  N a multiple of 16, Kpad of 32, zero filled.
std::vector<unsigned short> split_weight_planes(const float* W, int N, int K, int Kpad, float scale) {
  if (N % 16 != 0 || Kpad % 32 != 0 || K > Kpad) throw Error(kErrInvalidArg, "split_weight_planes: N % 16, Kpad % 32");
  std::vector<unsigned short> out(size_t(2) * N * Kpad, 0);
  const int nkt = Kpad / 32;
  for (int n = 0; n < N; ++n) {
    const int rg = n / 16, r = n % 16;
    for (int k = 0; k < K; ++k) {
      const float v = W[size_t(n) * K + k] * scale;
      const _Float16 h = static_cast<_Float16>(v);
      const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
      const int kt = k / 32, c = (k % 32) / 8, e = k % 8;
      const size_t blk = (size_t(rg) * nkt + kt) * 2 * 512;  // halfs
      const size_t at = size_t(r * 4 + (c ^ chunk_swizzle(r))) * 8 + e;
      std::memcpy(out.data() + blk + at, &h, 2);
      std::memcpy(out.data() + blk + 512 + at, &l, 2);
    }
  }
  return out;
}

bool launch_gemm_planes(const PlaneGemmArgs& a, int epi, hipStream_t s) {
  PlaneGemmDev g{};
  g.A = reinterpret_cast<const _Float16*>(a.A); g.a_plane = a.a_plane;
  g.W = reinterpret_cast<const _Float16*>(a.W);
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
    if (a.a_plane < 0 || a.a_bs < 0 || 2 * a_span >= (1L << 32)) {
      throw Error(kErrInvalidArg, "plane GEMM operand spans more than the 4 GiB its 32-bit offsets reach");
    }
  }
  switch (epi | (planes ? 256 : 0)) {
    case kEpiBias: return launch_planes<kEpiBias, false>(g, a.n_cu, s);
    case kEpiBias | kEpiGelu: return launch_planes<kEpiBias | kEpiGelu, false>(g, a.n_cu, s);  // fp32 for a fall-back consumer
    case kEpiBias | kEpiResidual: return launch_planes<kEpiBias | kEpiResidual, false>(g, a.n_cu, s);
    case kEpiBias | kEpiGelu | kEpiPos: return launch_planes<kEpiBias | kEpiGelu | kEpiPos, false>(g, a.n_cu, s);
    case kEpiBias | kEpiKvLayout: return launch_planes<kEpiBias | kEpiKvLayout, false>(g, a.n_cu, s);
    case kEpiBias | kEpiPower: return launch_planes<kEpiBias | kEpiPower, false>(g, a.n_cu, s);
    case kEpiBias | 256: return launch_planes<kEpiBias, true>(g, a.n_cu, s);
    case kEpiBias | kEpiGelu | 256: return launch_planes<kEpiBias | kEpiGelu, true>(g, a.n_cu, s);
    default: throw Error(kErrInvalidArg, "unsupported plane GEMM epilogue combination");
  }
}

}  // namespace wt

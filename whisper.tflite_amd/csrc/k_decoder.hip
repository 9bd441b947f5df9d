// Decoder-step GEMMs for gfx950: out[M][N] = x[M][K] . W[N][K]^T with M = clips x positions of one decoder
// pass (32 rows for a generated position of a 32-clip batch, 128 rows for the four prompt positions in one
// pass), i.e. one to four 32-row MFMA tiles.  These replace the Linear ops the reference runs inside every
// decoder Invoke() (whisper.tflite/whisper.cpp:375).
//
// A decoder pass is a chain of ~31 tiny dependent launches, each costing a kernel boundary plus one
// memory-latency chain, so the design minimises the number of launches and the length of each:
//   * weights are split at load time into two fp16 planes (hi, lo: 22 significand bits, csrc/bf16_split.h) and
//     pre-tiled into MFMA-fragment order ([tile][k/16][plane][lane][8]), so one wave-instruction reads 1 KiB
//     contiguous, a wave issues ALL its weight loads before anything else, and the contraction runs on
//     v_mfma_f32_32x32x16_f16 (three plane products per k-step, fp32 accumulation): 18 MFMAs of 32 cycles for a
//     96-deep k-slice where the fp32 instruction needed 48 of 64 cycles;
//   * activations are split in registers with a DYNAMIC power-of-two scale per (row, k-slice) taken from the
//     values themselves (no bounds, no range assumptions: the slice's largest element lands in [2^14, 2^15)),
//     and the accumulators are scaled back per row before the k-slices are summed;
//   * every block = 4 or 8 wavefronts splitting K, combined through LDS in a fixed order (deterministic, no
//     atomics);
//   * prologues/epilogues fuse what used to be separate launches: LayerNorm (+ token/positional embedding at
//     layer 0) and the cross-attention chunk combine in front of the GEMM, bias / GELU / residual add / argmax
//     behind it;
//   * every trip count that guards a load is a template constant or a clamped index: a runtime-predicated load
//     makes hipcc branch around it and wait vmcnt(0) per element (measured: 6x slower).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ unsigned ordered_bits(float v) {
  v = v + 0.0f;  // -0.0 -> +0.0 so that equal values compare equal
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct DecGemmDev {
  const unsigned short* Wt;
  float w_descale;
  int N, K, M, B;  // M rows = positions x B clips (row = p * B + b)
  const float* X;
  int ldx;
  const float* xin;
  float* xout;
  const float* ln_g;
  const float* ln_b;
  const long long* ids;
  int ids_stride, pos;
  const float* tok_emb;
  const float* pos_emb;
  int n_vocab;
  const float* cross_ws;
  int heads, chunks;
  const float* bias;
  int gelu;
  const float* R;
  float* Y;
  int ldy;
  unsigned long long* best;
  int best_stride;
  int ksplit;
  float* part;
  const float* xpart;
  int lnp_off;  // float offset of the LayerNorm gain / shift staging area in LDS
};

// Row sources of the residual stream.  LNMODE 0: x = xin;  2: x = tok_emb[id] + pos_emb (row = p * B + b reads
// ids[b][pos + p] and positional row pos + p);  3: x = xin + xpart (the second K-half of the previous residual
// GEMM is still pending in xpart).
struct RowSrc {
  const float* xin;
  const float* xpart;
  const long long* ids;
  int ids_stride, pos, B;
  const float* tok_emb;
  const float* pos_emb;
  int n_vocab;
};

// 8 lanes own one row (16-byte columns sub, sub + 8, ...); the NF4 loads of the row are
// independent and issued before the first use.
template <int NF4, int LNMODE>
__device__ __forceinline__ void load_row(f32x4 (&v)[NF4], const RowSrc& r, int row, int sub, int K) {
  const float* src = r.xin + (long)row * K;
  int p = 0;
  if (LNMODE == 2) {
    p = row / r.B;
    const int b = row - p * r.B;
    long long id = r.ids[(long)b * r.ids_stride + r.pos + p];
    id = id < 0 ? 0 : (id >= r.n_vocab ? r.n_vocab - 1 : id);  // never index outside the table
    src = r.tok_emb + id * K;
  }
#pragma unroll
  for (int j = 0; j < NF4; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (sub + 8 * j) * 4);
  if (LNMODE == 2) {
    const float* pe = r.pos_emb + (long)(r.pos + p) * K;
#pragma unroll
    for (int j = 0; j < NF4; ++j) v[j] += *reinterpret_cast<const f32x4*>(pe + (sub + 8 * j) * 4);
  }
  if (LNMODE == 3) {
    const float* pp = r.xpart + (long)row * K;
#pragma unroll
    for (int j = 0; j < NF4; ++j) v[j] += *reinterpret_cast<const f32x4*>(pp + (sub + 8 * j) * 4);
  }
}

// LayerNorm statistics of a row spread over 8 lanes (two-pass, eps 1e-5).
template <int NF4>
__device__ __forceinline__ void row_stats(const f32x4 (&v)[NF4], int K, float* mean, float* rstd) {
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < NF4; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  const float m = s / (float)K;
  float q = 0.0f;
#pragma unroll
  for (int j = 0; j < NF4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = v[j][e] - m;
      q += t * t;
    }
  q += __shfl_xor(q, 1, 64);
  q += __shfl_xor(q, 2, 64);
  q += __shfl_xor(q, 4, 64);
  *mean = m;
  *rstd = rsqrtf(q / (float)K + 1e-5f);
}

// PRO: kProNone / kProLn / kProCombine; LNMODE as above (kProLn only); NF4 = K / 32 (kProLn only); WAVES =
// wavefronts per block, all splitting K (8 for the narrow N = d_model GEMMs, which have only N/32 = 12 column
// tiles: parallelism has to come from K); CH = compile-time key-chunk count of the combine prologue; SMAX =
// 16-deep k-steps a wavefront keeps in flight (>= its k-slice / 16; weights: 2 planes x 16 B per lane per step).
// Rows are processed in groups of GT 32-row tiles: the LayerNorm rows of one group live in LDS at a time.
// BF: the bf16 storage mode (BASELINE configs[3]): weights are ONE bf16 plane in the same fragment order
// ([tile][k/16][lane][8]), activations are rounded to bf16 in registers, one v_mfma_f32_32x32x16_bf16 per k-step —
// no scales anywhere (bf16 has fp32's exponent range).
template <int PRO, int EPI, int NF4, int LNMODE, int WAVES, int CH, int GT, int SMAX, bool BF>
__global__ __launch_bounds__(WAVES * 64) void dec_gemm(DecGemmDev g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [GT*32][K + 4] LayerNorm rows (kProLn) and, once they are dead, [WAVES][GT][16][64] split-K partials; behind
  // them (kProLn) the LayerNorm gain and shift, [2][K]
  float* const xs = smem;
  float* const red = smem;
  float* const lnp = smem + g.lnp_off;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // ksplit = 2 (kProNone + kDecResid only): blocks [0, n_tiles) take the first half of K and finish the
  // residual, blocks [n_tiles, 2 n_tiles) take the second half and leave their partial in g.part
  const int n_tiles = (g.N + 31) / 32;
  const int tile = blockIdx.x % n_tiles, khalf = blockIdx.x / n_tiles;
  const int K = g.K, M = g.M, xld = K + 4;
  const int m_tiles = (M + 31) >> 5;

  const int kwave = (K / g.ksplit) / WAVES;  // a multiple of 16 (host-checked)
  const int k0 = khalf * (K / g.ksplit) + wid * kwave;
  const int nsteps = kwave >> 4;             // rounds of SMAX steps
  // weights of this (tile, k-slice): [step][plane][lane][8 halfs]; the stream does not depend on the
  // prologue, so it goes in flight now and its HBM/L2 latency overlaps the LayerNorm / combine work below.
  // Steps past nsteps re-read the last valid step (no branch around a load) and are never used.  (With more than
  // SMAX steps or several row groups the registers are refilled per round, below.)
  constexpr int WSTEP = BF ? 512 : 1024;  // 16-bit elements per (tile, k-step): one or two planes of 64 lanes x 8
  const unsigned short* wp = g.Wt + ((long)tile * (K >> 4) + (k0 >> 4)) * WSTEP + lane * 8;
  u32x4_t wh[SMAX], wl[BF ? 1 : SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    const int ss = s < nsteps ? s : nsteps - 1;
    wh[s] = *reinterpret_cast<const u32x4_t*>(wp + (long)ss * WSTEP);
    if (!BF) wl[s] = *reinterpret_cast<const u32x4_t*>(wp + (long)ss * WSTEP + 512);
  }
  // LayerNorm gain / shift are row-independent: requested now (one 16-byte load per thread), parked in LDS while
  // the rows are in flight, read back after the row statistics — not a second dependent trip to memory
  f32x4 lnreg = {0, 0, 0, 0};
  if (PRO == kProLn) {
    const int c4n = K >> 2;  // float4 per vector (<= 128)
    if (tid < 2 * c4n) lnreg = *reinterpret_cast<const f32x4*>((tid < c4n ? g.ln_g : g.ln_b - K) + tid * 4);
  }
  // the epilogue's column operand does not depend on the product either
  const int n_epi = tile * 32 + l31;
  const bool n_ok = n_epi < g.N;
  float bias = 0.0f;
  if ((EPI == kDecBias || EPI == kDecResid) && khalf == 0) bias = g.bias[n_ok ? n_epi : 0];

  for (int g0 = 0; g0 < m_tiles; g0 += GT) {
    if (PRO == kProLn) {
      // LayerNorm of the residual stream; a wavefront handles 8 rows at once, one memory round
      // trip per pass.  With the embedding / pending-partial source the rows are also materialised once (block 0).
      static_assert(PRO != kProLn || WAVES == 4, "the LayerNorm prologue maps 4 waves x 8 rows");
      const bool writer = (LNMODE == 2 || LNMODE == 3) && blockIdx.x == 0 && g.xout != nullptr;
      const int r8 = lane >> 3, sub = lane & 7;
      const RowSrc src{g.xin, g.xpart, g.ids, g.ids_stride, g.pos, g.B, g.tok_emb, g.pos_emb, g.n_vocab};
      constexpr int NV = NF4 > 0 ? NF4 : 1;
      if (g0 > 0) __syncthreads();  // the previous group's partials (aliasing xs) have been consumed
#pragma unroll 1
      for (int pass = 0; pass < GT; ++pass) {
        const int lrow = pass * 32 + wid * 8 + r8, row = g0 * 32 + lrow;
        f32x4 v[NV];
        if (row < M) load_row<NV, LNMODE>(v, src, row < M ? row : M - 1, sub, K);
        if (g0 == 0 && pass == 0) {
          if (tid < 2 * (K >> 2)) *reinterpret_cast<f32x4*>(&lnp[tid * 4]) = lnreg;
          __syncthreads();
        }
        if (row < M) {
          if (writer) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
              *reinterpret_cast<f32x4*>(g.xout + (long)row * K + (sub + 8 * j) * 4) = v[j];
          }
          float mean, rstd;
          row_stats<NV>(v, K, &mean, &rstd);
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const int c = (sub + 8 * j) * 4;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(&lnp[c]);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&lnp[K + c]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gg[e] + bb[e];
            *reinterpret_cast<f32x4*>(&xs[lrow * xld + c]) = o;
          }
        } else {
#pragma unroll
          for (int j = 0; j < NV; ++j)
            *reinterpret_cast<f32x4*>(&xs[lrow * xld + (sub + 8 * j) * 4]) = f32x4{0, 0, 0, 0};
        }
      }
      __syncthreads();
    }

    // the residual rows of the tile this wave will finish do not depend on the product: requested before the
    // contraction instead of after the split-K combine (one dependent round trip less per launch)
    // The GT x 16 accumulator registers of the group are finished by ALL wavefronts, PER of them each (item i = tile
    // i >> 4, register i & 15; wavefront w takes items w PER .. w PER + PER - 1) — one wavefront finishing a whole tile
    // while the others idle was the serial tail of every launch.
    constexpr int PER = GT * 16 / WAVES;
    float r_pre[EPI == kDecResid ? PER : 1];
    if (EPI == kDecResid && khalf == 0) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int i = wid * PER + j, t = i >> 4, r = i & 15;
        const int m = (g0 + t) * 32 + crow(r, lh);
        r_pre[j] = g.R[(long)(m < M ? m : M - 1) * g.ldy + (n_ok ? n_epi : 0)];
      }
    }
    f32x16 acc[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // k-slices deeper than SMAX steps (K = 4 d_model without the K split) take several rounds of SMAX steps; each
    // round has its own activation scale and is scaled back before it is added
    for (int c0 = 0; c0 < nsteps; c0 += SMAX) {
      if (c0 > 0 || g0 > 0) {
        if (nsteps > SMAX) {  // the weight registers hold one round at a time
#pragma unroll
          for (int s = 0; s < SMAX; ++s) {
            const int ss = c0 + s < nsteps ? c0 + s : nsteps - 1;
            wh[s] = *reinterpret_cast<const u32x4_t*>(wp + (long)ss * WSTEP);
            if (!BF) wl[s] = *reinterpret_cast<const u32x4_t*>(wp + (long)ss * WSTEP + 512);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < GT; ++t) {
        if (g0 + t >= m_tiles) continue;  // block-uniform
        const int rowl = t * 32 + l31;
        int row = (g0 + t) * 32 + l31;
        row = row < M ? row : M - 1;  // clamped: rows past M are computed and discarded
        // A fragments of this wave's k-slice: lane (row l31, half lh) holds x[row][k0 + 16 s + 8 lh .. + 7]
        float xa[SMAX][8];
        if (PRO == kProCombine) {
          // the A fragment is the combine of the cross-attention key-chunk partials (o[64], m, l, pad; 68-float
          // records), computed by the lane that consumes it — no LDS, every load independent
#pragma unroll
          for (int s = 0; s < SMAX; ++s) {
            const int ss = c0 + s < nsteps ? c0 + s : nsteps - 1;
            const int col = k0 + 16 * ss + 8 * lh;
            const int hh = col >> 6, dd = col & 63;
            const float* p = g.cross_ws + ((long)(row * g.heads + hh) * CH) * 68;
            float mc[CH], lc[CH];
            f32x4 pv0[CH], pv1[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {  // all loads first: CH is a compile-time constant
              mc[c] = p[c * 68 + 64];
              lc[c] = p[c * 68 + 65];
              pv0[c] = *reinterpret_cast<const f32x4*>(p + c * 68 + dd);
              pv1[c] = *reinterpret_cast<const f32x4*>(p + c * 68 + dd + 4);
            }
            float mxc = mc[0];
#pragma unroll
            for (int c = 1; c < CH; ++c) mxc = fmaxf(mxc, mc[c]);
            float l = 0.0f;
            f32x4 o0 = {0, 0, 0, 0}, o1 = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const float wgt = __expf(mc[c] - mxc);
              o0 += wgt * pv0[c];
              o1 += wgt * pv1[c];
              l += wgt * lc[c];
            }
            const float linv = 1.0f / l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xa[s][e] = o0[e] * linv;
              xa[s][4 + e] = o1[e] * linv;
            }
          }
        } else {
          const float* xp = (PRO == kProLn ? xs + rowl * xld : g.X + (long)row * g.ldx) + k0 + 8 * lh;
#pragma unroll
          for (int s = 0; s < SMAX; ++s) {
            const int ss = c0 + s < nsteps ? c0 + s : nsteps - 1;
            const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 16 * ss);
            const f32x4 b = *reinterpret_cast<const f32x4*>(xp + 16 * ss + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xa[s][e] = a[e];
              xa[s][4 + e] = b[e];
            }
          }
        }
        if (BF) {
          using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
#pragma unroll
          for (int s = 0; s < SMAX; ++s) {
            if (c0 + s < nsteps) {
              u32x4_t pa;
#pragma unroll
              for (int e = 0; e < 4; ++e) pa[e] = pack_bf16x2(xa[s][2 * e], xa[s][2 * e + 1]);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, wh[s]),
                                                               acc[t], 0, 0, 0);
            }
          }
          continue;
        }
        // dynamic scale of this (row, k-round): the largest element goes to [2^14, 2^15), inside fp16's range
        float mx = 0.0f;
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(xa[s][e]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const unsigned ex = (__float_as_uint(mx) >> 23) & 0xFFu;
        const float sc = ex < 32u ? 1.0f : __uint_as_float((268u - ex) << 23);   // 2^(14 - (ex - 127))
        const float inv = ex < 32u ? 1.0f : __uint_as_float((ex - 14u) << 23);   // 1 / sc
        f32x16 part;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          if (c0 + s < nsteps) {
            u32x4_t pl[3];
            split8_f16x2(xa[s], sc, pl);
            const half8 ah = __builtin_bit_cast(half8, pl[0]), al = __builtin_bit_cast(half8, pl[1]);
            const half8 bh = __builtin_bit_cast(half8, wh[s]), bl = __builtin_bit_cast(half8, wl[BF ? 0 : s]);
            part = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, part, 0, 0, 0);
            part = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, part, 0, 0, 0);
            part = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, part, 0, 0, 0);
          }
        }
        // scale back per row: accumulator register r of this lane belongs to row crow(r, lh), whose scale is
        // held by lane crow(r, lh)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += part[r] * (__shfl(inv, crow(r, lh), 64) * g.w_descale);
      }
    }

    // split-K combine through LDS, fixed order
    if (PRO == kProLn) {
      __syncthreads();  // every wave is done reading the LayerNorm rows the partials alias
    } else if (g0 > 0) {
      __syncthreads();  // the previous group's partials have been consumed
    }
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wid * GT + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = wid * PER + j, t = i >> 4, r = i & 15;
      if (g0 + t >= m_tiles) continue;  // wave-uniform
      float sum = red[(t * 16 + r) * 64 + lane];
#pragma unroll
      for (int wv = 1; wv < WAVES; ++wv) sum += red[((wv * GT + t) * 16 + r) * 64 + lane];  // fixed order: wave 0, 1, 2, ...
      const int m = (g0 + t) * 32 + crow(r, lh);
      float v = sum + bias;
      if (EPI == kDecBias && g.gelu) v = gelu_erf(v);
      const bool ok = n_ok && m < M;
      if (EPI == kDecResid && khalf == 1) {  // second K-half: the raw partial, summed by the consumer
        if (ok) g.part[(long)m * g.ldy + n_epi] = sum;
        continue;
      }
      // R may alias Y: each element is read and written by the same thread
      if (EPI == kDecResid && ok) v += r_pre[EPI == kDecResid ? j : 0];
      if (ok && g.Y) g.Y[(long)m * g.ldy + n_epi] = v;
      if (EPI == kDecLogits) {
        // fold (value, column): larger value wins, then the larger column — the reference's
        // `>=` scan keeps the LAST maximal index (whisper.cpp:353)
        unsigned long long p = ok ? (((unsigned long long)ordered_bits(v) << 32) | (unsigned)n_epi) : 0ull;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          const unsigned long long o2 = __shfl_xor(p, off, 64);
          p = o2 > p ? o2 : p;
        }
        // one record per (row, tile); select_token reduces them (no same-address atomics)
        if (l31 == 0 && m < M) g.best[(long)m * g.best_stride + tile] = p;
      }
    }
  }
}

// Logits GEMM, one 32-row tile per blockIdx.y (a generated position of a <= 32-clip batch is one tile): logits[m][n] = x[m] . E[n] against the
// tied embedding, N = n_vocab (51865: 1621 column tiles), plus the per-tile argmax records.  dec_gemm gives every
// column tile its own short-lived block (load 49 KB, 18 MFMAs, reduce, leave): 1621 blocks x ~11 us on 2 blocks per
// CU = 37 us per launch for 80 MB of weights.  Here 512 PERSISTENT blocks walk the column tiles: the activation
// planes (and their per-row scales) are made once per block instead of once per tile, and the weights of the next
// tile are requested before the MFMAs of the current one, so the stream never waits for a block to start.  Same
// arithmetic in the same order as dec_gemm<kProNone, kDecLogits> (per-wavefront k-slice partials, scaled back per row,
// summed over wavefronts 0..3), hence bit-identical logits and records.
// LNMODE >= 0: the rows are LayerNorm(xin [+ xpart]) (the decoder's final LayerNorm, load_row's modes 0 / 3), made
// by the block itself once — what used to be a separate dec_finalize_ln launch in front of every logits GEMM;
// LNMODE < 0: the rows are g.X as they are.
template <int KS, bool BF, int LNMODE>  // KS = 16-deep k-steps per wavefront = K / 64
__global__ __launch_bounds__(256, 2) void dec_logits_persistent(DecGemmDev g) {
  constexpr int KC = KS * 64, XLD = KC + 4;
  constexpr int kRed = 2 * 4 * 16 * 64, kXs = LNMODE >= 0 ? 32 * XLD : 0;
  // [tile parity][wavefront][register][lane] split-K partials; before the first tile the same memory holds the 32
  // normalised rows (LNMODE >= 0)
  __shared__ __attribute__((aligned(16))) float smem_l[kRed > kXs ? kRed : kXs];
  float (*red)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(smem_l);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n_tiles = (g.N + 31) / 32, K = g.K, M = g.M;
  const int k0 = wid * (K >> 2);
  constexpr int WSTEP = BF ? 512 : 1024;
  const unsigned short* const wbase = g.Wt + (long)(k0 >> 4) * WSTEP + lane * 8;
  const long tile_stride = (long)(K >> 4) * WSTEP;
  int tile = blockIdx.x;
  u32x4_t wh[KS], wl[BF ? 1 : KS];
  {
    const unsigned short* wp = wbase + (long)(tile < n_tiles ? tile : n_tiles - 1) * tile_stride;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wh[s] = *reinterpret_cast<const u32x4_t*>(wp + (long)s * WSTEP);
      if (!BF) wl[s] = *reinterpret_cast<const u32x4_t*>(wp + (long)s * WSTEP + 512);
    }
  }
  // activation planes of this wavefront's k-slice: lane (row l31, half lh) holds x[row][k0 + 16 s + 8 lh .. + 7]
  u32x4_t ah[KS], al[BF ? 1 : KS];
  float inv = 1.0f;
  const int m0 = blockIdx.y * 32;  // row tile: each one streams the weights (blockIdx.y > 0 only for more than 32 rows)
  if (LNMODE >= 0) {
    // 8 lanes per row, a wavefront normalises 8 rows: the 32 rows of the tile in one pass (as dec_finalize_ln did)
    constexpr int NF4 = 2 * KS;
    const int r8 = lane >> 3, sub = lane & 7, lrow = wid * 8 + r8, row = m0 + lrow;
    const RowSrc src{g.xin, g.xpart, nullptr, 0, 0, g.B, nullptr, nullptr, 0};
    f32x4 v[NF4], gg[NF4], bb[NF4];
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      gg[j] = *reinterpret_cast<const f32x4*>(g.ln_g + (sub + 8 * j) * 4);
      bb[j] = *reinterpret_cast<const f32x4*>(g.ln_b + (sub + 8 * j) * 4);
    }
    load_row<NF4, LNMODE < 0 ? 0 : LNMODE>(v, src, row < M ? row : M - 1, sub, K);
    float mean, rstd;
    row_stats<NF4>(v, K, &mean, &rstd);
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int c = (sub + 8 * j) * 4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gg[j][e] + bb[j][e];
      *reinterpret_cast<f32x4*>(&smem_l[lrow * XLD + c]) = o;
    }
    __syncthreads();
  }
  {
    int row = m0 + l31;
    row = row < M ? row : M - 1;
    const float* xp = (LNMODE >= 0 ? smem_l + l31 * XLD : g.X + (long)row * g.ldx) + k0 + 8 * lh;
    float xa[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 16 * s);
      const f32x4 b = *reinterpret_cast<const f32x4*>(xp + 16 * s + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xa[s][e] = a[e];
        xa[s][4 + e] = b[e];
      }
    }
    if (BF) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) ah[s][e] = pack_bf16x2(xa[s][2 * e], xa[s][2 * e + 1]);
    } else {
      float mx = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(xa[s][e]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const unsigned ex = (__float_as_uint(mx) >> 23) & 0xFFu;
      const float sc = ex < 32u ? 1.0f : __uint_as_float((268u - ex) << 23);
      inv = ex < 32u ? 1.0f : __uint_as_float((ex - 14u) << 23);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4_t pl[3];
        split8_f16x2(xa[s], sc, pl);
        ah[s] = pl[0];
        al[s] = pl[1];
      }
    }
  }
  if (LNMODE >= 0) __syncthreads();  // every wavefront has its fragments: the rows' memory becomes the partials'
  float rscale[16];  // accumulator register r belongs to row crow(r, lh), whose scale lane crow(r, lh) holds
#pragma unroll
  for (int r = 0; r < 16; ++r) rscale[r] = BF ? 1.0f : __shfl(inv, crow(r, lh), 64) * g.w_descale;

  for (int par = 0; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    // the next tile's weights first (a tile past the end re-reads this one: no branch around a load)
    const int nt = tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile;
    const unsigned short* np = wbase + (long)nt * tile_stride;
    u32x4_t nwh[KS], nwl[BF ? 1 : KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      nwh[s] = *reinterpret_cast<const u32x4_t*>(np + (long)s * WSTEP);
      if (!BF) nwl[s] = *reinterpret_cast<const u32x4_t*>(np + (long)s * WSTEP + 512);
    }
    f32x16 part;
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (BF) {
        using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
        part = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[s]), __builtin_bit_cast(bf16x8, wh[s]), part, 0, 0, 0);
      } else {
        const half8 a_h = __builtin_bit_cast(half8, ah[s]), a_l = __builtin_bit_cast(half8, al[BF ? 0 : s]);
        const half8 b_h = __builtin_bit_cast(half8, wh[s]), b_l = __builtin_bit_cast(half8, wl[BF ? 0 : s]);
        part = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_l, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, b_h, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_h, part, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[par][wid][r][lane] = BF ? part[r] : 0.0f + part[r] * rscale[r];
    __syncthreads();  // one barrier per tile: the other parity's buffer is only rewritten after the NEXT barrier
    {
      // every wavefront finishes four of the sixteen accumulator registers (= eight rows)
      const int n_epi = tile * 32 + l31;
      const bool n_ok = n_epi < g.N;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = wid * 4 + rr;
        float v = red[par][0][r][lane];
        v += red[par][1][r][lane];
        v += red[par][2][r][lane];
        v += red[par][3][r][lane];
        v += 0.0f;
        const int m = m0 + crow(r, lh);
        const bool ok = n_ok && m < M;
        if (ok && g.Y) g.Y[(long)m * g.ldy + n_epi] = v;
        // (value, column) record of this row over the tile's 32 columns: the largest ordered key, then the LARGEST
        // column among equals (the reference's `>=` scan, whisper.cpp:353) = the highest lane of the ballot
        const unsigned key = ok ? ordered_bits(v) : 0u;
        unsigned kmax = key;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          const unsigned o2 = __shfl_xor(kmax, off, 64);
          kmax = o2 > kmax ? o2 : kmax;
        }
        const unsigned long long hit = __ballot(ok && key == kmax);
        const unsigned half_hits = (unsigned)(hit >> (32 * lh));
        if (l31 == 0 && m < M) {
          const unsigned col = half_hits ? (unsigned)(tile * 32 + 31 - __clz(half_hits)) : 0u;
          g.best[(long)m * g.best_stride + tile] = half_hits ? (((unsigned long long)kmax << 32) | col) : 0ull;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wh[s] = nwh[s];
      if (!BF) wl[s] = nwl[s];
    }
  }
}

template <bool BF, int LNMODE>
bool launch_logits_persistent(const DecGemmDev& g, hipStream_t s, int blocks = 0) {
  if (g.K % 64 != 0 || g.ksplit != 1) return false;
  const int n_tiles = (g.N + 31) / 32, m_tiles = (g.M + 31) / 32;
  static const int forced = [] {  // measurement knob: resident blocks in all
    const char* v = getenv("WT_LOGITS_BLOCKS");
    const int n = v ? atoi(v) : 0;
    return n >= 32 && n <= 1024 ? n : 0;
  }();
  // 512 = two per CU when the decoder has the chip; next to an encoder 256 measure 0.5 % more end to end
  const int total = forced ? forced : (blocks >= 32 && blocks <= 1024 ? blocks : 512);
  const int per = total / m_tiles;
  const dim3 grid(n_tiles < per ? n_tiles : per, m_tiles);
  switch (g.K / 64) {
    case 2: hipLaunchKernelGGL((dec_logits_persistent<2, BF, LNMODE>), grid, dim3(256), 0, s, g); return true;
    case 6: hipLaunchKernelGGL((dec_logits_persistent<6, BF, LNMODE>), grid, dim3(256), 0, s, g); return true;
    case 8: hipLaunchKernelGGL((dec_logits_persistent<8, BF, LNMODE>), grid, dim3(256), 0, s, g); return true;
    default: return false;
  }
}

// y = LayerNorm(x)   (input rows of the logits GEMM).
template <int NF4, int LNMODE>
__global__ __launch_bounds__(256) void dec_finalize_ln(RowSrc src, const float* __restrict__ g,
                                                       const float* __restrict__ b,
                                                       float* __restrict__ y, int B, int K) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r8 = lane >> 3, sub = lane & 7;
  const int row = blockIdx.x * 32 + wid * 8 + r8;
  if (row >= B) return;  // whole 8-lane groups leave together; the shuffles stay inside a group
  f32x4 v[NF4], gg[NF4], bb[NF4];
#pragma unroll
  for (int j = 0; j < NF4; ++j) {  // gain / shift requested with the row, not after its statistics
    gg[j] = *reinterpret_cast<const f32x4*>(g + (sub + 8 * j) * 4);
    bb[j] = *reinterpret_cast<const f32x4*>(b + (sub + 8 * j) * 4);
  }
  load_row<NF4, LNMODE>(v, src, row, sub, K);
  float mean, rstd;
  row_stats<NF4>(v, K, &mean, &rstd);
#pragma unroll
  for (int j = 0; j < NF4; ++j) {
    const int c = (sub + 8 * j) * 4;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gg[j][e] + bb[j][e];
    *reinterpret_cast<f32x4*>(y + (long)row * K + c) = o;
  }
}

template <int PRO, int EPI, int NF4, int LNMODE, int WAVES, int CH, int GT, int SMAX, bool BF>
void launch_one(const DecGemmDev& g, size_t smem, dim3 grid, hipStream_t s) {
  // dynamic LDS beyond the 64 KiB default needs an opt-in, once per kernel
  static const bool raised = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_gemm<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, SMAX, BF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)raised;
  hipLaunchKernelGGL((dec_gemm<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, SMAX, BF>), grid, dim3(WAVES * 64), smem, s, g);
}

template <int PRO, int EPI, int NF4, int LNMODE, int WAVES, int CH, int GT, bool BF>
void launch_steps(DecGemmDev g, hipStream_t s) {
  const int n_tiles = (g.N + 31) / 32;
  const dim3 grid(n_tiles * (PRO == kProNone && EPI == kDecResid ? g.ksplit : 1));
  const size_t red_bytes = (size_t)WAVES * GT * 16 * 64 * sizeof(float);
  const size_t xs_bytes = PRO == kProLn ? (size_t)GT * 32 * (g.K + 4) * sizeof(float) : 0;
  const size_t body = red_bytes > xs_bytes ? red_bytes : xs_bytes;
  g.lnp_off = (int)(body / sizeof(float));
  const size_t smem = body + (PRO == kProLn ? (size_t)2 * g.K * sizeof(float) : 0);
  const int nsteps = (g.K / g.ksplit) / WAVES / 16;  // k-steps per wavefront; deeper slices run in rounds of 8
  if (nsteps <= 2) launch_one<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, 2, BF>(g, smem, grid, s);
  else if (nsteps == 3) launch_one<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, 3, BF>(g, smem, grid, s);
  else if (nsteps == 4) launch_one<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, 4, BF>(g, smem, grid, s);
  else if (nsteps <= 6) launch_one<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, 6, BF>(g, smem, grid, s);
  else launch_one<PRO, EPI, NF4, LNMODE, WAVES, CH, GT, 8, BF>(g, smem, grid, s);
}

template <int PRO, int EPI, int NF4, int LNMODE, int WAVES, int CH, bool BF>
void launch_gt(const DecGemmDev& g, hipStream_t s) {
  // row groups of two 32-row tiles when there is more than one tile: the weights of a (tile, k-slice) are then loaded
  // once per 64 rows.  The LayerNorm rows of a group live in LDS: 64 x (K + 4) x 4 B = 99 KB at K = 384, 132 KB at
  // K = 512 (+ 4 KB of gain / shift: inside the 160 KB a block may use)
  const bool two = g.M > 32 && !(PRO == kProLn && g.K > 512);
  if (two) {
    launch_steps<PRO, EPI, NF4, LNMODE, WAVES, CH, 2, BF>(g, s);
  } else {
    launch_steps<PRO, EPI, NF4, LNMODE, WAVES, CH, 1, BF>(g, s);
  }
}

template <int LNMODE, bool BF>
void launch_ln(const DecGemmDev& g, hipStream_t s) {
  switch (g.K) {
    case 128: launch_gt<kProLn, kDecBias, 4, LNMODE, 4, 1, BF>(g, s); break;
    case 384: launch_gt<kProLn, kDecBias, 12, LNMODE, 4, 1, BF>(g, s); break;
    case 512: launch_gt<kProLn, kDecBias, 16, LNMODE, 4, 1, BF>(g, s); break;
    default: throw Error(kErrFormat, "decoder kernels support d_model 128, 384 or 512");
  }
}

}  // namespace

template <bool BF>
static void dispatch_dec_gemm(const DecGemmArgs& a, const DecGemmDev& g, int pro, int epi, hipStream_t s) {
  if (pro == kProLn && epi == kDecLogits) {  // final LayerNorm inside the persistent logits kernel
    if (a.ids || !a.xin || !a.ln_g || !a.ln_b) throw Error(kErrInvalidArg, "logits GEMM: LayerNorm rows come from xin (+ xpart)");
    const bool ok = a.xpart ? launch_logits_persistent<BF, 3>(g, s, a.logits_blocks) : launch_logits_persistent<BF, 0>(g, s, a.logits_blocks);
    if (!ok) throw Error(kErrInvalidArg, "logits GEMM with the LayerNorm prologue: K must be a multiple of 64");
    return;
  }
  if (pro == kProLn) {
    if (epi != kDecBias) throw Error(kErrInvalidArg, "decoder GEMM: the LayerNorm prologue pairs with the bias epilogue");
    if (a.ids) {
      if (a.xpart) throw Error(kErrInvalidArg, "decoder GEMM: embedding rows have no pending partial");
      launch_ln<2, BF>(g, s);
    } else if (a.xpart) {
      launch_ln<3, BF>(g, s);
    } else {
      launch_ln<0, BF>(g, s);
    }
    return;
  }
  const int key = pro * 8 + epi;
  switch (key) {
    case kProNone * 8 + kDecResid: launch_gt<kProNone, kDecResid, 0, 0, 8, 1, BF>(g, s); break;
    case kProNone * 8 + kDecBias: launch_gt<kProNone, kDecBias, 0, 0, 4, 1, BF>(g, s); break;
    case kProNone * 8 + kDecLogits:
      if (!launch_logits_persistent<BF, -1>(g, s, a.logits_blocks)) launch_gt<kProNone, kDecLogits, 0, 0, 4, 1, BF>(g, s);
      break;
    case kProCombine * 8 + kDecResid:
      switch (a.chunks) {  // compile-time chunk count keeps the partial loads independent
        case 1: launch_gt<kProCombine, kDecResid, 0, 0, 8, 1, BF>(g, s); break;
        case 2: launch_gt<kProCombine, kDecResid, 0, 0, 8, 2, BF>(g, s); break;
        case 4: launch_gt<kProCombine, kDecResid, 0, 0, 8, 4, BF>(g, s); break;
        case 8: launch_gt<kProCombine, kDecResid, 0, 0, 8, 8, BF>(g, s); break;
        default: throw Error(kErrInvalidArg, "cross_chunks must be 1, 2, 4 or 8");
      }
      break;
    default: throw Error(kErrInvalidArg, "unsupported decoder GEMM prologue / epilogue pair");
  }
}

void launch_dec_gemm(const DecGemmArgs& a, int pro, int epi, hipStream_t s) {
  const int gelu = epi == kDecBiasGelu ? 1 : 0;
  if (epi == kDecBiasGelu) epi = kDecBias;
  const int M = a.M > 0 ? a.M : a.B;
  const int ksplit = a.ksplit > 0 ? a.ksplit : 1;
  DecGemmDev g{a.Wt,      1.0f / a.w_scale, a.N,  a.K,       M,         a.B,     a.X,     a.ldx,   a.xin,  a.xout,
               a.ln_g,    a.ln_b,    a.ids,      a.ids_stride, a.pos,   a.tok_emb, a.pos_emb, a.n_vocab,
               a.cross_ws, a.heads,  a.chunks,   a.bias,    gelu,      a.R,     a.Y,     a.ldy,   a.best,
               a.best_stride > 0 ? a.best_stride : (a.N + 31) / 32, ksplit, a.part,  a.xpart, 0};
  // host-side shape contract: operands must match what the kernel indexes
  const bool resid = epi == kDecResid;  // N = d_model: 8 wavefronts split K
  const int waves = resid ? 8 : 4;
  const int kblock = a.K / ksplit;
  if (!a.Wt || a.B < 1 || M < a.B || M > 128 || M % a.B != 0 || a.K > 4096 || a.K % 16 != 0 || !(a.w_scale > 0.0f) ||
      ksplit > 2 || kblock % (16 * waves) != 0 ||
      (pro == kProCombine && a.K != a.heads * 64) || (resid && (!a.R || !a.Y))) {
    throw Error(kErrInvalidArg, "decoder GEMM shape outside the kernel contract");
  }
  if (ksplit != 1 && !(pro == kProNone && epi == kDecResid && a.part && a.R != a.Y)) {
    // split K: residual GEMM only, with a partial buffer, out of place (the consumer completes the rows into R)
    throw Error(kErrInvalidArg, "decoder GEMM: K split needs the out-of-place residual form");
  }
  if (a.bf16) {
    dispatch_dec_gemm<true>(a, g, pro, epi, s);
  } else {
    dispatch_dec_gemm<false>(a, g, pro, epi, s);
  }
}

template <int LNMODE>
static void launch_finalize_mode(const RowSrc& src, const float* g, const float* b, float* y, int B, int K,
                                 hipStream_t s) {
  const dim3 grid((B + 31) / 32);
  switch (K) {
    case 128: hipLaunchKernelGGL((dec_finalize_ln<4, LNMODE>), grid, dim3(256), 0, s, src, g, b, y, B, K); break;
    case 384: hipLaunchKernelGGL((dec_finalize_ln<12, LNMODE>), grid, dim3(256), 0, s, src, g, b, y, B, K); break;
    case 512: hipLaunchKernelGGL((dec_finalize_ln<16, LNMODE>), grid, dim3(256), 0, s, src, g, b, y, B, K); break;
    default: throw Error(kErrFormat, "decoder kernels support d_model 128, 384 or 512");
  }
}

void launch_dec_finalize_ln(const float* xin, const float* g, const float* b, float* y, int B, int K,
                            hipStream_t s, const float* xpart) {
  const RowSrc src{xin, xpart, nullptr, 0, 0, B, nullptr, nullptr, 0};
  if (xpart) {
    launch_finalize_mode<3>(src, g, b, y, B, K, s);
  } else {
    launch_finalize_mode<0>(src, g, b, y, B, K, s);
  }
}

}  // namespace wt

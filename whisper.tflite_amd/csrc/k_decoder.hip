// Decoder-step GEMMs for gfx950: out[B][N] = x[B][K] . W[N][K]^T with B = clips in the batch
// (<= 64), i.e. one or two 32-row MFMA tiles, fp32 v_mfma_f32_32x32x2_f32.  These replace the
// Linear ops the reference runs inside every decoder Invoke() (whisper.tflite/whisper.cpp:375).
//
// A decoder position is a chain of ~35 tiny dependent launches, each costing a kernel
// boundary plus one memory-latency chain, so the design minimises the number of launches and
// maximises loads in flight per launch:
//   * weights are pre-tiled at load time into MFMA-fragment order ([tile][k/8][lane][4]), so
//     one wave-instruction reads 1 KiB contiguous and a wave issues ALL its weight loads
//     (<= 12 x 16 B per lane) before anything else — operand streamed once, straight to VGPRs;
//   * every block = 4 wavefronts splitting K, combined through LDS in a fixed order
//     (deterministic, no atomics);
//   * prologues/epilogues fuse what used to be separate launches: LayerNorm (+ token/positional
//     embedding at layer 0) and the cross-attention chunk combine in front of the GEMM, bias /
//     GELU / residual add / argmax behind it.  A prologue reads ONE array (the residual
//     stream): a first version folded 4 split-K slabs + bias there and was latency-bound on
//     the 6x larger read (26 us per launch instead of ~8);
//   * every trip count that guards a load is a template constant: a runtime-predicated load
//     makes hipcc branch around it and wait vmcnt(0) per element (measured: 6x slower).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kGroupMax = 12;  // weight chunks (8 k each) a wave keeps in flight (4-wave blocks)

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ unsigned ordered_bits(float v) {
  v = v + 0.0f;  // -0.0 -> +0.0 so that equal values compare equal
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct DecGemmDev {
  const float* Wt;
  int N, K, B;
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
  int ksplit;
  float* part;
  const float* xpart;
};

// Row sources of the residual stream.  LNMODE 0: x = xin;  2: x = tok_emb[id] + pos_emb;  3: x = xin + xpart
// (the second K-half of the previous residual GEMM is still pending in xpart).
struct RowSrc {
  const float* xin;
  const float* xpart;
  const long long* ids;
  int ids_stride, pos;
  const float* tok_emb;
  const float* pos_emb;
  int n_vocab;
};

// 8 lanes own one row (16-byte columns sub, sub + 8, ...); the NF4 loads of the row are
// independent and issued before the first use.
template <int NF4, int LNMODE>
__device__ __forceinline__ void load_row(f32x4 (&v)[NF4], const RowSrc& r, int row, int sub, int B,
                                         int K) {
  const float* src = r.xin + (long)row * K;
  if (LNMODE == 2) {
    long long id = r.ids[(long)row * r.ids_stride + r.pos];
    id = id < 0 ? 0 : (id >= r.n_vocab ? r.n_vocab - 1 : id);  // never index outside the table
    src = r.tok_emb + id * K;
  }
#pragma unroll
  for (int j = 0; j < NF4; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (sub + 8 * j) * 4);
  if (LNMODE == 2) {
    const float* pe = r.pos_emb + (long)r.pos * K;
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

// PRO: kProNone / kProLn / kProCombine; LNMODE as above (kProLn only); NF4 = K / 32 (kProLn
// only); WAVES = wavefronts per block, all splitting K (16 for the narrow N = d_model GEMMs,
// which have only N/32 = 12 column tiles: parallelism has to come from K); CH = compile-time
// key-chunk count of the combine prologue.
template <int PRO, int EPI, int MT, int NF4, int LNMODE, int WAVES, int CH>
__global__ __launch_bounds__(WAVES * 64) void dec_gemm(DecGemmDev g) {
  // 1024-thread blocks are capped at 128 VGPRs: keep 6 chunks (not 12) in flight there
  constexpr int kGroup = WAVES > 8 ? 6 : kGroupMax;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;                              // [WAVES-1][MT][16][64] split-K partials
  float* xs = smem + (WAVES - 1) * MT * 16 * 64;  // [MT*32][K + 4] LayerNorm rows (kProLn)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // ksplit = 2 (kProNone + kDecResid only): blocks [0, n_tiles) take the first half of K and finish the
  // residual, blocks [n_tiles, 2 n_tiles) take the second half and leave their partial in g.part
  const int n_tiles = (g.N + 31) / 32;
  const int tile = blockIdx.x % n_tiles, khalf = blockIdx.x / n_tiles;
  const int K = g.K, B = g.B, xld = K + 4;

  const int kwave = (K / g.ksplit) / WAVES;
  const int k0 = khalf * (K / g.ksplit) + wid * kwave;
  const int nchunks = kwave >> 3;
  const float* wp = g.Wt + ((long)tile * (K >> 3) + (k0 >> 3)) * 256 + lane * 4;
  // the weight stream does not depend on the prologue: its first group goes in flight now so
  // that its HBM/L2 latency overlaps the residual / LayerNorm / combine work below.  Chunks
  // past nchunks re-read the last valid chunk (no branch around a load) and are never used.
  f32x4 w[kGroup];
#pragma unroll
  for (int i = 0; i < kGroup; ++i) {
    const int ci = i < nchunks ? i : nchunks - 1;
    w[i] = *reinterpret_cast<const f32x4*>(wp + (long)ci * 256);
  }

  // The epilogue's operands (bias, residual rows) do not depend on the product either: wave 0, the
  // only one that reaches the epilogue, requests them now.  Loaded after the split-K reduction they
  // cost one more dependent round trip to memory per launch, 4-5 us each when the encoder and two
  // other decoders keep the memory system busy (WT_DEC_KERNEL_TIMERS).
  constexpr bool kPreR = EPI == kDecResid && MT == 1;  // MT = 2 blocks sit at the 128-VGPR cap
  const int n_epi = tile * 32 + l31;
  float bias_pre = 0.0f;
  float r_pre[kPreR ? 16 : 1];
  if (wid == 0 && khalf == 0) {
    if ((EPI == kDecBias || EPI == kDecResid) && MT == 1) bias_pre = g.bias[n_epi < g.N ? n_epi : 0];
    if (kPreR) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = (r & 3) + 8 * (r >> 2) + 4 * lh;
        r_pre[r] = g.R[(long)(b < B ? b : B - 1) * g.ldy + (n_epi < g.N ? n_epi : 0)];
      }
    }
  }

  if (PRO == kProLn) {
    // LayerNorm of the residual stream; a wavefront handles 8 rows at once, one memory round
    // trip per pass.  With the embedding source the rows are also materialised once (block 0).
    const bool writer = (LNMODE == 2 || LNMODE == 3) && blockIdx.x == 0 && g.xout != nullptr;
    const int r8 = lane >> 3, sub = lane & 7;
    const RowSrc src{g.xin, g.xpart, g.ids, g.ids_stride, g.pos, g.tok_emb, g.pos_emb, g.n_vocab};
    constexpr int NV = NF4 > 0 ? NF4 : 1;
    static_assert(PRO != kProLn || WAVES == 4, "the LayerNorm prologue maps 4 waves x 8 rows");
    // gain and shift are row-independent: requested together with the rows, not after their statistics
    // (plain rows only: the embedding gather, LNMODE 2, the two-array sum, LNMODE 3, and two M-tiles already
    // sit at the register budget)
    constexpr bool kHoistLn = LNMODE == 0 && MT == 1;
    f32x4 lg[kHoistLn ? NV : 1], lb[kHoistLn ? NV : 1];
    if (kHoistLn) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        lg[j] = *reinterpret_cast<const f32x4*>(g.ln_g + (sub + 8 * j) * 4);
        lb[j] = *reinterpret_cast<const f32x4*>(g.ln_b + (sub + 8 * j) * 4);
      }
    }
#pragma unroll 1
    for (int pass = 0; pass < MT; ++pass) {
      const int row = pass * 32 + wid * 8 + r8;
      f32x4 v[NV];
      if (row < B) {
        load_row<NV, LNMODE>(v, src, row, sub, B, K);
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
          const f32x4 gg = kHoistLn ? lg[kHoistLn ? j : 0] : *reinterpret_cast<const f32x4*>(g.ln_g + c);
          const f32x4 bb = kHoistLn ? lb[kHoistLn ? j : 0] : *reinterpret_cast<const f32x4*>(g.ln_b + c);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gg[e] + bb[e];
          *reinterpret_cast<f32x4*>(&xs[row * xld + c]) = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j)
          *reinterpret_cast<f32x4*>(&xs[row * xld + (sub + 8 * j) * 4]) = f32x4{0, 0, 0, 0};
      }
    }
    __syncthreads();
  }

  const float* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int b = t * 32 + l31;
    b = b < B ? b : B - 1;
    xp[t] = (PRO == kProLn ? xs + (t * 32 + l31) * xld : g.X + (long)b * g.ldx) + k0 + 4 * lh;
  }
  // kProCombine: the A fragment is the combine of the cross-attention key-chunk partials
  // (o[64], m, l, pad; 68-float records), computed by the lane that consumes it — no LDS,
  // every load independent (host guarantees nchunks <= kGroup for this prologue)
  f32x4 xa[PRO == kProCombine ? kGroup : 1][MT];
  if (PRO == kProCombine) {
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
      if (i < nchunks) {
        const int col = k0 + 8 * i + 4 * lh;
        const int hh = col >> 6, dd = col & 63;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int rowb = t * 32 + l31;
          f32x4 o = {0, 0, 0, 0};
          if (rowb < B) {
            const float* p = g.cross_ws + ((long)(rowb * g.heads + hh) * CH) * 68;
            float mc[CH], lc[CH];
            f32x4 pv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {  // all loads first: CH is a compile-time constant
              mc[c] = p[c * 68 + 64];
              lc[c] = p[c * 68 + 65];
              pv[c] = *reinterpret_cast<const f32x4*>(p + c * 68 + dd);
            }
            float mx = mc[0];
#pragma unroll
            for (int c = 1; c < CH; ++c) mx = fmaxf(mx, mc[c]);
            float l = 0.0f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const float wgt = __expf(mc[c] - mx);
              o += wgt * pv[c];
              l += wgt * lc[c];
            }
            const float inv = 1.0f / l;
            o *= inv;
          }
          xa[i][t] = o;
        }
      }
    }
  }

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  for (int c0 = 0; c0 < nchunks; c0 += kGroup) {
    if (c0 > 0) {
#pragma unroll
      for (int i = 0; i < kGroup; ++i) {
        const int ci = c0 + i < nchunks ? c0 + i : nchunks - 1;
        w[i] = *reinterpret_cast<const f32x4*>(wp + (long)ci * 256);
      }
    }
    // kProNone reads its A fragments from global memory: issue them all, unguarded (clamped
    // index), before the first MFMA
    f32x4 xg[PRO == kProNone ? kGroup : 1][MT];
    if (PRO == kProNone) {
#pragma unroll
      for (int i = 0; i < kGroup; ++i) {
        const int ci = c0 + i < nchunks ? c0 + i : nchunks - 1;
#pragma unroll
        for (int t = 0; t < MT; ++t) xg[i][t] = *reinterpret_cast<const f32x4*>(xp[t] + ci * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
      if (c0 + i < nchunks) {
        f32x4 x[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if (PRO == kProCombine) {
            x[t] = xa[i][t];
          } else if (PRO == kProNone) {
            x[t] = xg[i][t];
          } else {
            x[t] = *reinterpret_cast<const f32x4*>(xp[t] + (c0 + i) * 8);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < MT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[t][j], w[i][j], acc[t], 0, 0, 0);
      }
    }
  }

  if (wid > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((wid - 1) * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wid > 0) return;
#pragma unroll 1
  for (int wv = 0; wv < WAVES - 1; ++wv)  // fixed order: wave 1, 2, 3, ... (rolled: 16 loads live)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += red[((wv * MT + t) * 16 + r) * 64 + lane];

  const int n = n_epi;
  const bool n_ok = n < g.N;
  const float bias = ((EPI == kDecBias || EPI == kDecResid) && n_ok) ? (MT == 1 ? bias_pre : g.bias[n]) : 0.0f;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int b = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = acc[t][r] + bias;
      if (EPI == kDecBias && g.gelu) v = gelu_erf(v);
      const bool ok = n_ok && b < B;
      if (EPI == kDecResid && khalf == 1) {  // second K-half: the raw partial, summed by the consumer
        if (ok) g.part[(long)b * g.ldy + n] = acc[t][r];
        continue;
      }
      // R may alias Y: each element is read and written by the same thread
      if (EPI == kDecResid && ok) v += kPreR ? r_pre[r] : g.R[(long)b * g.ldy + n];
      if (ok && g.Y) g.Y[(long)b * g.ldy + n] = v;
      if (EPI == kDecLogits) {
        // fold (value, column): larger value wins, then the larger column — the reference's
        // `>=` scan keeps the LAST maximal index (whisper.cpp:353)
        unsigned long long p = ok ? (((unsigned long long)ordered_bits(v) << 32) | (unsigned)n) : 0ull;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          const unsigned long long o2 = __shfl_xor(p, off, 64);
          p = o2 > p ? o2 : p;
        }
        // one record per (clip, tile); select_token reduces them (no same-address atomics)
        if (l31 == 0 && b < B) g.best[(long)b * gridDim.x + tile] = p;
      }
    }
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
  load_row<NF4, LNMODE>(v, src, row, sub, B, K);
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

template <int PRO, int EPI, int NF4, int LNMODE, int WAVES, int CH>
void launch_mt(const DecGemmDev& g, hipStream_t s) {
  const int n_tiles = (g.N + 31) / 32;
  const int MT = g.B <= 32 ? 1 : 2;
  const size_t smem =
      (size_t)((WAVES - 1) * MT * 16 * 64 + (PRO == kProLn ? MT * 32 * (g.K + 4) : 0)) * sizeof(float);
  const dim3 grid(n_tiles * (PRO == kProNone && EPI == kDecResid ? g.ksplit : 1));
  // dynamic LDS beyond the 64 KiB default needs an opt-in, once per kernel
  static const bool raised = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_gemm<PRO, EPI, 1, NF4, LNMODE, WAVES, CH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_gemm<PRO, EPI, 2, NF4, LNMODE, WAVES, CH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)raised;
  if (MT == 1) {
    hipLaunchKernelGGL((dec_gemm<PRO, EPI, 1, NF4, LNMODE, WAVES, CH>), grid, dim3(WAVES * 64), smem, s, g);
  } else {
    hipLaunchKernelGGL((dec_gemm<PRO, EPI, 2, NF4, LNMODE, WAVES, CH>), grid, dim3(WAVES * 64), smem, s, g);
  }
}

template <int LNMODE>
void launch_ln(const DecGemmDev& g, hipStream_t s) {
  switch (g.K) {
    case 128: launch_mt<kProLn, kDecBias, 4, LNMODE, 4, 1>(g, s); break;
    case 384: launch_mt<kProLn, kDecBias, 12, LNMODE, 4, 1>(g, s); break;
    case 512: launch_mt<kProLn, kDecBias, 16, LNMODE, 4, 1>(g, s); break;
    default: throw Error(kErrFormat, "decoder kernels support d_model 128, 384 or 512");
  }
}

}  // namespace

void launch_dec_gemm(const DecGemmArgs& a, int pro, int epi, hipStream_t s) {
  const int gelu = epi == kDecBiasGelu ? 1 : 0;
  if (epi == kDecBiasGelu) epi = kDecBias;
  DecGemmDev g{a.Wt,      a.N,       a.K,       a.B,        a.X,     a.ldx,    a.xin,  a.xout, a.ln_g,
               a.ln_b,    a.ids,     a.ids_stride, a.pos,   a.tok_emb, a.pos_emb, a.n_vocab,
               a.cross_ws, a.heads,  a.chunks,  a.bias,     gelu,    a.R,      a.Y,    a.ldy,  a.best,
               a.ksplit,  a.part,    a.xpart};
  // host-side shape contract: operands must match what the kernel indexes
  const bool wide = epi == kDecResid;  // N = d_model: 16 wavefronts split K
  if (a.B < 1 || a.B > 64 || a.K > 2048 || a.K % (wide ? 8 * a.resid_waves : 32) != 0 || (a.resid_waves != 4 && a.resid_waves != 8 && a.resid_waves != 16) ||
      (pro == kProCombine && (a.K / (8 * a.resid_waves) > (a.resid_waves > 8 ? 6 : 12) || a.K != a.heads * 64)) || (wide && (!a.R || !a.Y))) {
    throw Error(kErrInvalidArg, "decoder GEMM shape outside the kernel contract");
  }
  if (a.ksplit != 1 && !(a.ksplit == 2 && pro == kProNone && epi == kDecResid && a.part &&
                         a.K % (16 * a.resid_waves) == 0 && a.R != a.Y)) {
    // split K: residual GEMM only, with a partial buffer, out of place (the consumer completes the rows into R)
    throw Error(kErrInvalidArg, "decoder GEMM: K split needs the out-of-place residual form");
  }
  if (pro == kProLn) {
    if (epi != kDecBias) throw Error(kErrInvalidArg, "decoder GEMM: the LayerNorm prologue pairs with the bias epilogue");
    if (a.ids) {
      if (a.xpart) throw Error(kErrInvalidArg, "decoder GEMM: embedding rows have no pending partial");
      launch_ln<2>(g, s);
    } else if (a.xpart) {
      launch_ln<3>(g, s);
    } else {
      launch_ln<0>(g, s);
    }
    return;
  }
  const int key = pro * 8 + epi;
  switch (key) {
    case kProNone * 8 + kDecResid:
      switch (a.resid_waves) {
        case 4: launch_mt<kProNone, kDecResid, 0, 0, 4, 1>(g, s); break;
        case 8: launch_mt<kProNone, kDecResid, 0, 0, 8, 1>(g, s); break;
        default: launch_mt<kProNone, kDecResid, 0, 0, 16, 1>(g, s); break;
      }
      break;
    case kProNone * 8 + kDecBias: launch_mt<kProNone, kDecBias, 0, 0, 4, 1>(g, s); break;
    case kProNone * 8 + kDecLogits: launch_mt<kProNone, kDecLogits, 0, 0, 4, 1>(g, s); break;
    case kProCombine * 8 + kDecResid:
      switch (a.chunks) {  // compile-time chunk count keeps the partial loads independent
        case 1: launch_mt<kProCombine, kDecResid, 0, 0, 16, 1>(g, s); break;
        case 2: launch_mt<kProCombine, kDecResid, 0, 0, 16, 2>(g, s); break;
        case 4:
          if (a.resid_waves == 4) {
            launch_mt<kProCombine, kDecResid, 0, 0, 4, 4>(g, s);
          } else if (a.resid_waves == 8) {
            launch_mt<kProCombine, kDecResid, 0, 0, 8, 4>(g, s);
          } else {
            launch_mt<kProCombine, kDecResid, 0, 0, 16, 4>(g, s);
          }
          break;
        case 8: launch_mt<kProCombine, kDecResid, 0, 0, 16, 8>(g, s); break;
        default: throw Error(kErrInvalidArg, "cross_chunks must be 1, 2, 4 or 8");
      }
      break;
    default: throw Error(kErrInvalidArg, "unsupported decoder GEMM prologue / epilogue pair");
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
  const RowSrc src{xin, xpart, nullptr, 0, 0, nullptr, nullptr, 0};
  if (xpart) {
    launch_finalize_mode<3>(src, g, b, y, B, K, s);
  } else {
    launch_finalize_mode<0>(src, g, b, y, B, K, s);
  }
}

}  // namespace wt

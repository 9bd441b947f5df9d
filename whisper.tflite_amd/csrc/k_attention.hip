// Attention kernels for gfx950: the encoder's 1500x1500 self-attention (dense QK^T / PV
// contractions on fp32 MFMA with in-register online softmax) and the decoder's
// single-row self / cross attention over the persistent KV caches (HBM streaming).
// Together with k_gemm.hip these replace the attention ops inside the graphs the
// reference runs through tflite::Interpreter::Invoke() (whisper.tflite/whisper.cpp:295,
// :375; graph per export/generate_onnx.py:85-120).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------ encoder attention ---
// Block = 4 wavefronts = 128 query rows of one (clip, head); K/V tiles of 64 keys go
// global -> registers -> LDS once per block and are shared by the 4 wavefronts.
//
// "Swapped" products keep the softmax row on the lane:
//   S^T[key][q] = K . Q^T      A = K tile (LDS, ds_read_b128), B = Q (registers)
//   O^T[d][q]  += V^T . P^T    A = V tile (LDS, ds_read_b32),  B = P = exp2(S^T - m)
// The C/D layout puts query q = lane & 31 on the lane and 16 keys in the registers
// (row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)), so register r of S^T is, unmoved,
// the B operand of PV step r when V is read at that same key order.
constexpr int AQ = 128;      // queries per block
constexpr int AK = 64;       // keys per tile
constexpr int KLD = 68;      // K tile row stride (floats): conflict-free ds_read_b128
constexpr int VLD = 64;

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(256) void encoder_attention_f32(const float* __restrict__ qkv,
                                                             float* __restrict__ out, int T,
                                                             int heads) {
  __shared__ __attribute__((aligned(16))) float Ks[AK * KLD];
  __shared__ __attribute__((aligned(16))) float Vs[AK * VLD];

  const int d_model = heads * 64, ld = 3 * d_model;
  const int q_blocks = (T + AQ - 1) / AQ;
  // consecutive blocks on one XCD (blockIdx % 8 equal) walk the q-blocks of one
  // (clip, head), so its K/V stay in that XCD's L2
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / q_blocks, qb = logical % q_blocks;
  const int b = bh / heads, h = bh % heads;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* base = qkv + (long)b * T * ld + h * 64;

  // Q fragment: lane (q = l31, half lh) holds Q[q][8c + 4lh + j], pre-scaled by
  // d_head^-1/2 * log2(e) so that softmax is exp2(s - max)
  const int q_row = qb * AQ + wid * 32 + l31;
  const int q_ld = q_row < T ? q_row : T - 1;
  const float qscale = 0.125f * 1.44269504088896340736f;
  f32x4 qf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    qf[c] = *reinterpret_cast<const f32x4*>(base + (long)q_ld * ld + 8 * c + 4 * lh);
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[c][j] *= qscale;
  }

  f32x16 o0, o1;  // O^T tiles: d in [0,32) and [32,64)
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  // staging map: 64 rows x 16 float4 = 1024 float4 per operand, 4 per thread
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  const float* kbase = base + d_model + scol;
  const float* vbase = base + 2 * d_model + scol;
  f32x4 rk[4], rv[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = kt * AK + srow + 16 * i;
      if (key < T) {
        rk[i] = *reinterpret_cast<const f32x4*>(kbase + (long)key * ld);
        rv[i] = *reinterpret_cast<const f32x4*>(vbase + (long)key * ld);
      } else {
        rk[i] = f32x4{0, 0, 0, 0};
        rv[i] = f32x4{0, 0, 0, 0};
      }
    }
  };
  const int n_tiles = (T + AK - 1) / AK;
  load_tile(0);

  for (int kt = 0; kt < n_tiles; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(&Ks[(srow + 16 * i) * KLD + scol]) = rk[i];
      *reinterpret_cast<f32x4*>(&Vs[(srow + 16 * i) * VLD + scol]) = rv[i];
    }
    __syncthreads();
    if (kt + 1 < n_tiles) load_tile(kt + 1);

    // S^T for the two 32-key halves of the tile
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(&Ks[l31 * KLD + 8 * c + 4 * lh]);
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(&Ks[(32 + l31) * KLD + 8 * c + 4 * lh]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[j], qf[c][j], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[j], qf[c][j], s1, 0, 0, 0);
      }
    }
    if ((kt + 1) * AK > T) {  // last tile: keys past T do not exist
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kt * AK + crow(r, lh) >= T) s0[r] = -1e30f;
        if (kt * AK + 32 + crow(r, lh) >= T) s1[r] = -1e30f;
      }
    }
    // online softmax; the row (query) lives on lanes l and l ^ 32
    float tmax = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s1[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // raw v_exp_f32: arguments are <= 0 and far from the denormal range that exp2f() guards
      s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
      s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
      psum += s0[r] + s1[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    // rescale only when some row's running max moved (wave-uniform branch; after the first
    // tiles it rarely does): alpha == 1 exactly otherwise, so skipping is bit-identical
    if (__any(m_new != m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
      m_run = m_new;
    }
    l_run += psum;
    // O^T += V^T . P^T : step r contracts key crow(r, lh) (+32 for the second half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = crow(r, lh);
      const float v00 = Vs[key * VLD + l31];
      const float v01 = Vs[key * VLD + 32 + l31];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v00, s0[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v01, s0[r], o1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 + crow(r, lh);
      const float v10 = Vs[key * VLD + l31];
      const float v11 = Vs[key * VLD + 32 + l31];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v10, s1[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v11, s1[r], o1, 0, 0, 0);
    }
    __syncthreads();
  }

  if (q_row < T) {
    const float inv = 1.0f / l_run;
    float* orow = out + ((long)b * T + q_row) * d_model + h * 64;
    // lane holds d = crow(r, lh) (+32): groups of 4 consecutive d -> 16-byte stores
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a, c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = o0[4 * g + j] * inv;
        c[j] = o1[4 * g + j] * inv;
      }
      *reinterpret_cast<f32x4*>(orow + 8 * g + 4 * lh) = a;
      *reinterpret_cast<f32x4*>(orow + 32 + 8 * g + 4 * lh) = c;
    }
  }
}

// ------------------------------------- encoder attention on the bf16 cores ---
// Same algorithm and block shape as encoder_attention_f32, with both contractions on
// v_mfma_f32_32x32x16_bf16 through the exact three-plane split of bf16_split.h: Q (scaled), K
// and V are split between the global load and the LDS write / fragment registers, the
// probabilities P = exp2(S^T - m) are split in registers, and each product runs as the six
// significant plane products with fp32 accumulation (fp32-level error, 6/16 of the MFMA cycles).
//   S^T[key][q] = K . Q^T     A = K planes (LDS rows, ds_read_b128), B = Q planes (registers)
//   O^T[d][q]  += V^T . P^T   A = V^T planes (LDS image [d][key], two ds_read_b64), B = P planes
// The S^T accumulator keeps query q on lane (l31, lh) and key (r&3) + 8(r>>2) + 4lh in register r,
// so registers 8s..8s+7 are, unmoved, the B fragment of PV step s if V^T is read at keys
// 16s + 4lh + {0..3} and 16s + 8 + 4lh + {0..3}.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
constexpr int SKLD = 72;  // K plane row stride (bf16): 144 B, an odd multiple of 16 B
constexpr int SVLD = 68;  // V^T plane row stride (bf16): 136 B, conflict-free ds_read_b64 per half-wave

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4_t& v) { return __builtin_bit_cast(bf16x8, v); }

// NS = 3: six bf16 plane products; NS = 2: three fp16 plane products (two-plane fp16 split, bf16_split.h);
// NS = 1 (bf16 compute mode): the single product of the rounded operands
using half8 = __attribute__((ext_vector_type(8))) _Float16;
#define WT_MM16(A, B, ACC) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, A), __builtin_bit_cast(half8, B), ACC, 0, 0, 0)
#define WT_SPLIT_PRODUCTS(ACC, AF, BF)                                                        \
  if (NS == 1) {                                                                              \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[0], ACC, 0, 0, 0);                \
  } else if (NS == 2) {                                                                       \
    ACC = WT_MM16(AF[0], BF[1], ACC);                                                         \
    ACC = WT_MM16(AF[1], BF[0], ACC);                                                         \
    ACC = WT_MM16(AF[0], BF[0], ACC);                                                         \
  } else {                                                                                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[NS - 1], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[1], BF[1], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[NS - 1], BF[0], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[1], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[1], BF[0], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[0], ACC, 0, 0, 0);                  \
  }

template <int NW, int NS>  // NW wavefronts per block = 32 queries each, sharing the K/V tile staging; NS planes
// (NS = 2: the probabilities are scaled by 2^14 into fp16's normal range before their split and the
// output is scaled back with the softmax normalisation)
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void encoder_attention_split(const float* __restrict__ qkv,
                                                                                   float* __restrict__ out, int T,
                                                                                   int heads, float q_scale, float k_scale,
                                                                                   float v_scale) {
  constexpr int AQS = NW * 32;      // queries per block
  constexpr int PASSES = 8 / NW;    // staging passes per 64-key tile (NW * 64 threads)
  constexpr int KROWS = 64 / PASSES, VPAIRS = 32 / PASSES;
  __shared__ __attribute__((aligned(16))) unsigned short Kp[3 * AK * SKLD];
  __shared__ __attribute__((aligned(16))) unsigned short Vt[3 * 64 * SVLD];

  const int d_model = heads * 64, ld = 3 * d_model;
  const int q_blocks = (T + AQS - 1) / AQS;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / q_blocks, qb = logical % q_blocks;
  const int b = bh / heads, h = bh % heads;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* base = qkv + (long)b * T * ld + h * 64;

  constexpr float kPScale = NS == 2 ? 16384.0f : 1.0f;
  auto planes8 = [](const float (&x)[8], u32x4_t (&o)[3], float scale = 1.0f) {
    if (NS == 1) {
      o[0] = round8_bf16(x);
    } else if (NS == 2) {
      split8_f16x2(x, scale, o);
    } else {
      split8_planes(x, o);
    }
  };
  auto planes2 = [v_scale](float lo, float hi, unsigned (&w)[3]) {
    if (NS == 1) {
      w[0] = round2_bf16(lo, hi);
    } else if (NS == 2) {
      split2_f16x2(lo, hi, v_scale, w);
    } else {
      split2_planes(lo, hi, w);
    }
  };
  // Q planes: lane (q = l31, half lh) holds Q[q][16c + 8lh + 0..7] for k-step c, pre-scaled by
  // d_head^-1/2 * log2(e) in fp32 (as the fp32 kernel does) and then split
  const int q_row = qb * AQS + wid * 32 + l31;
  const int q_ld = q_row < T ? q_row : T - 1;
  const float qscale = 0.125f * 1.44269504088896340736f;
  bf16x8 qf[4][3];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(base + (long)q_ld * ld + 16 * c + 8 * lh);
    const f32x4 bq = *reinterpret_cast<const f32x4*>(base + (long)q_ld * ld + 16 * c + 8 * lh + 4);
    const float x[8] = {a[0] * qscale, a[1] * qscale, a[2] * qscale, a[3] * qscale,
                        bq[0] * qscale, bq[1] * qscale, bq[2] * qscale, bq[3] * qscale};
    u32x4_t o[3];
    planes8(x, o, NS == 2 ? q_scale : 1.0f);
#pragma unroll
    for (int p = 0; p < NS; ++p) qf[c][p] = as_bf16x8(o[p]);
  }

  f32x16 o0, o1;  // O^T tiles: d in [0,32) and [32,64)
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  // staging maps.  K: 8 threads per key row, 8 d each (two float4), 32 keys per pass.
  // V: thread = (key pair, 4 d); the pair becomes one dword of the [d][key] image.
  const int ksrow = tid >> 3, kscol = (tid & 7) * 8;
  const int vpair = tid >> 4, vdcol = (tid & 15) * 4;
  const float* kbase = base + d_model + kscol;
  const float* vbase = base + 2 * d_model + vdcol;
  f32x4 rk[2 * PASSES], rv[2 * PASSES];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      // rows past T are loaded from row T - 1 and zeroed by a select (no divergent control flow)
      const int key = kt * AK + ksrow + KROWS * i;
      const int kc = key < T ? key : T - 1;
      const float kz = key < T ? 1.0f : 0.0f;
      rk[2 * i] = *reinterpret_cast<const f32x4*>(kbase + (long)kc * ld) * kz;
      rk[2 * i + 1] = *reinterpret_cast<const f32x4*>(kbase + (long)kc * ld + 4) * kz;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int vkey = kt * AK + 2 * (vpair + VPAIRS * i) + e;
        const int vc = vkey < T ? vkey : T - 1;
        const float vz = vkey < T ? 1.0f : 0.0f;
        rv[2 * i + e] = *reinterpret_cast<const f32x4*>(vbase + (long)vc * ld) * vz;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const float x[8] = {rk[2 * i][0], rk[2 * i][1], rk[2 * i][2], rk[2 * i][3],
                          rk[2 * i + 1][0], rk[2 * i + 1][1], rk[2 * i + 1][2], rk[2 * i + 1][3]};
      u32x4_t o[3];
      planes8(x, o, NS == 2 ? k_scale : 1.0f);
#pragma unroll
      for (int p = 0; p < NS; ++p)
        *reinterpret_cast<u32x4_t*>(&Kp[p * AK * SKLD + (ksrow + KROWS * i) * SKLD + kscol]) = o[p];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned w[3];
        planes2(rv[2 * i][e], rv[2 * i + 1][e], w);
#pragma unroll
        for (int p = 0; p < NS; ++p)
          *reinterpret_cast<unsigned*>(&Vt[p * 64 * SVLD + (vdcol + e) * SVLD + 2 * (vpair + VPAIRS * i)]) = w[p];
      }
    }
  };
  const int n_tiles = (T + AK - 1) / AK;
  load_tile(0);

  for (int kt = 0; kt < n_tiles; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < n_tiles) load_tile(kt + 1);

    // S^T for the two 32-key halves of the tile
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 k0[3], k1[3];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        k0[p] = *reinterpret_cast<const bf16x8*>(&Kp[p * AK * SKLD + l31 * SKLD + 16 * c + 8 * lh]);
        k1[p] = *reinterpret_cast<const bf16x8*>(&Kp[p * AK * SKLD + (32 + l31) * SKLD + 16 * c + 8 * lh]);
      }
      WT_SPLIT_PRODUCTS(s0, k0, qf[c])
      WT_SPLIT_PRODUCTS(s1, k1, qf[c])
    }
    if (NS == 2) {  // the operand scales of the fp16 planes leave the scores
      const float s_inv = 1.0f / (q_scale * k_scale);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] *= s_inv;
        s1[r] *= s_inv;
      }
    }
    if ((kt + 1) * AK > T) {  // last tile: keys past T do not exist
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kt * AK + crow(r, lh) >= T) s0[r] = -1e30f;
        if (kt * AK + 32 + crow(r, lh) >= T) s1[r] = -1e30f;
      }
    }
    // online softmax; the row (query) lives on lanes l and l ^ 32
    float tmax = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s1[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
      s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
      psum += s0[r] + s1[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    if (__any(m_new != m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
      m_run = m_new;
    }
    l_run += psum;
    // O^T += V^T . P^T, 16 keys per step
    auto pv_half = [&](const f32x16& sp, const int hf) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float pv[8] = {sp[8 * s2], sp[8 * s2 + 1], sp[8 * s2 + 2], sp[8 * s2 + 3],
                             sp[8 * s2 + 4], sp[8 * s2 + 5], sp[8 * s2 + 6], sp[8 * s2 + 7]};
        u32x4_t po[3];
        planes8(pv, po, kPScale);
        bf16x8 pf[3];
#pragma unroll
        for (int p = 0; p < NS; ++p) pf[p] = as_bf16x8(po[p]);
        const int key0 = hf * 32 + 16 * s2 + 4 * lh;
        bf16x8 v0[3], v1[3];
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          const unsigned short* r0 = &Vt[p * 64 * SVLD + l31 * SVLD + key0];
          const unsigned short* r1 = &Vt[p * 64 * SVLD + (32 + l31) * SVLD + key0];
          const u32x2 a0 = *reinterpret_cast<const u32x2*>(r0), a1 = *reinterpret_cast<const u32x2*>(r0 + 8);
          const u32x2 b0 = *reinterpret_cast<const u32x2*>(r1), b1 = *reinterpret_cast<const u32x2*>(r1 + 8);
          v0[p] = as_bf16x8(u32x4_t{a0[0], a0[1], a1[0], a1[1]});
          v1[p] = as_bf16x8(u32x4_t{b0[0], b0[1], b1[0], b1[1]});
        }
        WT_SPLIT_PRODUCTS(o0, v0, pf)
        WT_SPLIT_PRODUCTS(o1, v1, pf)
      }
    };
    pv_half(s0, 0);
    pv_half(s1, 1);
    __syncthreads();
  }

  if (q_row < T) {
    const float inv = (1.0f / (kPScale * (NS == 2 ? v_scale : 1.0f))) / l_run;
    float* orow = out + ((long)b * T + q_row) * d_model + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a, c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = o0[4 * g + j] * inv;
        c[j] = o1[4 * g + j] * inv;
      }
      *reinterpret_cast<f32x4*>(orow + 8 * g + 4 * lh) = a;
      *reinterpret_cast<f32x4*>(orow + 32 + 8 * g + 4 * lh) = c;
    }
  }
}
#undef WT_SPLIT_PRODUCTS
#undef WT_MM16

// ------------------------------------------------- decoder self attention ---
// One block = one wavefront per (clip, head).  Appends the k, v of `npos` new positions (pos0 .. pos0 + npos - 1;
// npos = 1 for a generated position, the whole prompt in the first pass) to the cache, then for each new position
// lane j scores cached position j and lane d accumulates output d.  qkv / out rows: row(p) = p * B + b.
// BF: the cache rows are bf16 (bf16 storage mode); the new position's k and v are rounded before they are used, so a
// row contributes the same values whether it was just computed or read back.
template <bool BF>
__global__ void self_attention_step(const float* __restrict__ qkv, void* __restrict__ kcache_v,
                                    void* __restrict__ vcache_v, int cap, int pos0, int npos, int B,
                                    float* __restrict__ out, int heads) {
  // Everything this (clip, head) needs is requested in one batch of independent loads — the new q/k/v rows and the
  // cached K and V rows of positions < pos0 (16 lanes x 16 B per row; 8 B in the bf16 mode) — and staged in LDS;
  // scores, softmax and the weighted sum then run out of LDS.  One round trip to memory.
  using CT = typename std::conditional<BF, unsigned short, float>::type;
  using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, lane = threadIdx.x;
  const int d = heads * 64;
  CT* kc = static_cast<CT*>(kcache_v) + ((long)b * cap) * d + h * 64;
  CT* vc = static_cast<CT*>(vcache_v) + ((long)b * cap) * d + h * 64;
  __shared__ __attribute__((aligned(16))) float Ks[32][68];
  __shared__ __attribute__((aligned(16))) float Vs[32][64];
  __shared__ float qs[64];
  __shared__ float ps[64];
  const int r16 = lane >> 4, c4 = (lane & 15) * 4;
  f32x4 kr[8], vr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // rows 4i + r16 < pos0 (pos0 <= 31); rows past pos0 re-read row 0
    const int j = 4 * i + r16;
    const int jj = j < pos0 ? j : 0;
    if (BF) {
      const u32x2 k2 = *reinterpret_cast<const u32x2*>(kc + (long)jj * d + c4);
      const u32x2 v2 = *reinterpret_cast<const u32x2*>(vc + (long)jj * d + c4);
      kr[i] = f32x4{bf16_lo(k2[0]), bf16_hi(k2[0]), bf16_lo(k2[1]), bf16_hi(k2[1])};
      vr[i] = f32x4{bf16_lo(v2[0]), bf16_hi(v2[0]), bf16_lo(v2[1]), bf16_hi(v2[1])};
    } else {
      kr[i] = *reinterpret_cast<const f32x4*>(kc + (long)jj * d + c4);
      vr[i] = *reinterpret_cast<const f32x4*>(vc + (long)jj * d + c4);
    }
  }
  for (int p = 0; p < npos; ++p) {
    const float* row = qkv + ((long)p * B + b) * 3 * d;
    float knew = row[d + h * 64 + lane];
    float vnew = row[2 * d + h * 64 + lane];
    if (BF) {
      const unsigned pk = pack_bf16x2(knew, vnew);
      kc[(long)(pos0 + p) * d + lane] = (CT)(pk & 0xFFFFu);
      vc[(long)(pos0 + p) * d + lane] = (CT)(pk >> 16);
      knew = bf16_lo(pk);
      vnew = bf16_hi(pk);
    } else {
      kc[(long)(pos0 + p) * d + lane] = knew;
      vc[(long)(pos0 + p) * d + lane] = vnew;
    }
    Ks[pos0 + p][lane] = knew;
    Vs[pos0 + p][lane] = vnew;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = 4 * i + r16;
    if (j < pos0) {
      *reinterpret_cast<f32x4*>(&Ks[j][c4]) = kr[i];
      *reinterpret_cast<f32x4*>(&Vs[j][c4]) = vr[i];
    }
  }
  for (int p = 0; p < npos; ++p) {
    const long r = (long)p * B + b;
    __syncthreads();  // staging complete / the previous position is done with qs and ps
    qs[lane] = qkv[r * 3 * d + h * 64 + lane] * 0.125f;
    __syncthreads();
    const int n = pos0 + p + 1;  // causal: keys 0 .. pos0 + p
    float s = -1e30f;
    if (lane < n) {
      float acc = 0.0f;
#pragma unroll 8
      for (int c = 0; c < 64; ++c) acc += qs[c] * Ks[lane][c];
      s = acc;
    }
    float mx = s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float pe = lane < n ? __expf(s - mx) : 0.0f;
    float sum = pe;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    ps[lane] = pe / sum;
    __syncthreads();
    float o = 0.0f;
    for (int j = 0; j < n; ++j) o += ps[j] * Vs[j][lane];
    out[r * d + h * 64 + lane] = o;
  }
}

// ------------------------------------------------ decoder cross attention ---
// One block per (clip, head, key chunk), NQ query rows (NQ = 1 for a generated position; the prompt positions of
// the first pass share one sweep of the cache).  The block first makes its own queries — q = LayerNorm(x[row]) .
// Wq[head slice]^T + bq, a 64 x DM matrix-vector product per row, 98 KB of weights from L2 — so the separate
// "LayerNorm + query projection" launch of every layer is gone; the first K/V rows are requested before that, and
// the chunk's K and V slabs ([keys][64] floats, 256 B per key, contiguous) are then streamed ONCE in a single pass:
// 16 lanes own a key row (16-byte loads), each 16-lane group runs an online softmax over its keys (4 keys = 8
// loads of 16 B per lane in flight), and the 16 group partials are merged through LDS at the end.
// Output per (row, head, chunk): o[64] (unnormalised), m (running max, natural-log units), l (sum).
// Wq_t: the query projection re-laid-out for this product: [head][DM / 4][64 outputs][4 k] (cross_q_layout()).
// BF (bf16 storage mode): the cache rows are 64 bf16 = 128 B, 8 lanes own a key row (16-byte loads of 8 elements) and
// a block sweeps 32 keys per step instead of 16: half the bytes AND half the per-key shuffle / exp2 work per lane.
template <int NQ, int DM, bool BF>
__global__ __launch_bounds__(256) void cross_attention_step(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                            const float* __restrict__ ln_b,
                                                            const float* __restrict__ Wq_t, const float* __restrict__ bq,
                                                            const void* __restrict__ kc_v, const void* __restrict__ vc_v,
                                                            float* __restrict__ ws, int B, int heads, int T,
                                                            int chunks) {
  using CT = typename std::conditional<BF, unsigned short, float>::type;
  constexpr int LPK = BF ? 8 : 16;   // lanes per key row
  constexpr int EPL = 64 / LPK;      // elements per lane: one 16-byte load either way
  constexpr int NG = 256 / LPK;      // key rows in flight per load round
  __shared__ __attribute__((aligned(16))) float lnx[NQ][DM];
  __shared__ __attribute__((aligned(16))) float qpart[4][NQ][64];
  __shared__ __attribute__((aligned(16))) float qs[NQ][64];
  __shared__ __attribute__((aligned(16))) float go[NQ][NG * 64];
  __shared__ float gm[NQ][NG], gl[NQ][NG];
  __shared__ float red[4][NQ];
  const int chunk = blockIdx.x % chunks, bh = blockIdx.x / chunks;
  const int b = bh / heads, h = bh % heads;
  const int per = (T + chunks - 1) / chunks;
  const int k_begin = chunk * per, k_end = min(T, k_begin + per), nk = k_end - k_begin;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, grp = tid / LPK, gl16 = tid % LPK;
  constexpr float kLog2e = 1.44269504088896340736f;
  constexpr int U = 4;
  constexpr int C4 = DM / 4;    // float4 per row
  constexpr int CW = DM / 16;   // float4 of the k-quarter one wavefront contracts

  // (0) the first K/V rows: nothing below touches them before the queries exist.  The cache is read once per launch
  // and is far larger than L2 + Infinity Cache: streaming (non-temporal) loads keep those for the decoder's weights.
  const CT* kb = static_cast<const CT*>(kc_v) + ((long)bh * T + k_begin) * 64 + gl16 * EPL;
  const CT* vb = static_cast<const CT*>(vc_v) + ((long)bh * T + k_begin) * 64 + gl16 * EPL;
  u32x4_t kv[U], vv[U];  // 16 bytes: four floats, or eight bf16
#pragma unroll
  for (int u = 0; u < U; ++u) {  // unguarded loads: keys past the chunk re-read its last row
    const int k = grp + NG * u;
    const int kk = k < nk ? k : nk - 1;
    kv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kb + (long)kk * 64));
    vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vb + (long)kk * 64));
  }
  // (1) query-projection weights of this head, k-quarter `wid`, output `lane`: CW independent 16-byte loads
  f32x4 wq[CW];
  const float* wqp = Wq_t + (((long)h * C4 + wid * CW) * 64 + lane) * 4;
#pragma unroll
  for (int c = 0; c < CW; ++c) wq[c] = *reinterpret_cast<const f32x4*>(wqp + (long)c * 256);
  // (2) LayerNorm of the NQ residual rows (two-pass statistics, eps 1e-5, as the LN-fused GEMMs)
  const int c4 = tid < C4 ? tid : C4 - 1;
  f32x4 xv[NQ];
#pragma unroll
  for (int p = 0; p < NQ; ++p) xv[p] = *reinterpret_cast<const f32x4*>(x + ((long)p * B + b) * DM + c4 * 4);
  const f32x4 gg = *reinterpret_cast<const f32x4*>(ln_g + c4 * 4), bb = *reinterpret_cast<const f32x4*>(ln_b + c4 * 4);
  const bool own = tid < C4;
  float mean[NQ], rstd[NQ];
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    float s = own ? (xv[p][0] + xv[p][1]) + (xv[p][2] + xv[p][3]) : 0.0f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[wid][p] = s;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NQ; ++p) mean[p] = ((red[0][p] + red[1][p]) + (red[2][p] + red[3][p])) / (float)DM;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    float q = 0.0f;
    if (own) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = xv[p][e] - mean[p];
        q += t * t;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    if (lane == 0) red[wid][p] = q;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    rstd[p] = rsqrtf(((red[0][p] + red[1][p]) + (red[2][p] + red[3][p])) / (float)DM + 1e-5f);
    if (own) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (xv[p][e] - mean[p]) * rstd[p] * gg[e] + bb[e];
      *reinterpret_cast<f32x4*>(&lnx[p][c4 * 4]) = o;
    }
  }
  __syncthreads();
  // (3) q[p][lane] partial over this wavefront's k-quarter; LDS reads of lnx are wave-uniform (broadcast)
  float qa[NQ];
#pragma unroll
  for (int p = 0; p < NQ; ++p) qa[p] = 0.0f;
#pragma unroll
  for (int c = 0; c < CW; ++c) {
#pragma unroll
    for (int p = 0; p < NQ; ++p) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(&lnx[p][(wid * CW + c) * 4]);
      qa[p] += (wq[c][0] * l4[0] + wq[c][1] * l4[1]) + (wq[c][2] * l4[2] + wq[c][3] * l4[3]);
    }
  }
#pragma unroll
  for (int p = 0; p < NQ; ++p) qpart[wid][p][lane] = qa[p];
  __syncthreads();
  constexpr float kScale = 0.125f * kLog2e;  // scores in log2 units: p = exp2(s - m)
  if (tid < NQ * 64) {
    const int p = tid >> 6;
    qs[p][lane] = (((qpart[0][p][lane] + qpart[1][p][lane]) + (qpart[2][p][lane] + qpart[3][p][lane])) + bq[h * 64 + lane]) * kScale;
  }
  __syncthreads();
  float qv[NQ][EPL];
#pragma unroll
  for (int p = 0; p < NQ; ++p)
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[p][e] = qs[p][gl16 * EPL + e];

  // (4) one sweep of the chunk's keys
  float m[NQ], l[NQ];
  float o[NQ][EPL];
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    m[p] = -1e30f;
    l[p] = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[p][e] = 0.0f;
  }
  for (int k0 = grp; k0 < nk; k0 += NG * U) {
    if (k0 != grp) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + NG * u;
        const int kk = k < nk ? k : nk - 1;
        kv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kb + (long)kk * 64));
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vb + (long)kk * 64));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool past = k0 + NG * u >= nk;
      float kf[EPL], vf[EPL];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (BF) {
          kf[2 * e] = bf16_lo(kv[u][e]);
          kf[2 * e + 1] = bf16_hi(kv[u][e]);
          vf[2 * e] = bf16_lo(vv[u][e]);
          vf[2 * e + 1] = bf16_hi(vv[u][e]);
        } else {
          kf[e % EPL] = __uint_as_float(kv[u][e]);
          vf[e % EPL] = __uint_as_float(vv[u][e]);
        }
      }
#pragma unroll
      for (int p = 0; p < NQ; ++p) {
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += kf[e] * qv[p][e];
        if (!BF) s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (past) s = -1e30f;
        const float mn = fmaxf(m[p], s);
        const float a = exp2f(m[p] - mn), pr = exp2f(s - mn);
        l[p] = l[p] * a + pr;
#pragma unroll
        for (int j = 0; j < EPL; ++j) o[p][j] = o[p][j] * a + pr * vf[j];
        m[p] = mn;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) go[p][grp * 64 + gl16 * EPL + e] = o[p][e];
    if (gl16 == 0) {
      gm[p][grp] = m[p];
      gl[p][grp] = l[p];
    }
  }
  __syncthreads();
  if (tid < NQ * 64) {
    const int p = tid >> 6;
    float mx = gm[p][0];
#pragma unroll
    for (int gI = 1; gI < NG; ++gI) mx = fmaxf(mx, gm[p][gI]);
    float acc = 0.0f, lsum = 0.0f;
#pragma unroll
    for (int gI = 0; gI < NG; ++gI) {
      const float w = exp2f(gm[p][gI] - mx);
      acc += w * go[p][gI * 64 + lane];
      lsum += w * gl[p][gI];
    }
    float* dst = ws + ((((long)p * B + b) * heads + h) * chunks + chunk) * 68;
    dst[lane] = acc;
    if (lane == 0) {
      dst[64] = mx * (1.0f / kLog2e);  // natural-log units for the combine prologue
      dst[65] = lsum;
    }
  }
}

}  // namespace

void launch_encoder_attention(const float* qkv, float* out, int batch, int T, int heads, int variant,
                              hipStream_t s, float q_scale, float k_scale, float v_scale) {
  const int q_blocks = (T + AQ - 1) / AQ;
  const dim3 grid(batch * heads * q_blocks);
  (void)q_scale; (void)k_scale; (void)v_scale;
  if (variant == 0) {
    hipLaunchKernelGGL(encoder_attention_f32, grid, dim3(256), 0, s, qkv, out, T, heads);
  } else if (variant == 1) {  // three bf16 planes, six products: full fp32 operand range (the engine's fall-back form)
    hipLaunchKernelGGL((encoder_attention_split<4, 3>), grid, dim3(256), 0, s, qkv, out, T, heads, 1.0f, 1.0f, 1.0f);
  } else {
    throw Error(kErrInvalidArg, "fp32-storage encoder attention: variant must be 0 (fp32 MFMA) or 1 (three bf16 planes)");
  }
}

void launch_self_attention(const float* qkv, void* kcache, void* vcache, int cap, int pos0, int npos, float* out,
                           int batch, int heads, hipStream_t s, bool bf16) {
  // the kernel stages at most 32 cached rows in LDS
  if (pos0 < 0 || npos < 1 || pos0 + npos > 32 || pos0 + npos > cap) {
    throw Error(kErrInvalidArg, "decoder self-attention: positions outside [0, 31]");
  }
  if (bf16) {
    hipLaunchKernelGGL(self_attention_step<true>, dim3(batch * heads), dim3(64), 0, s, qkv, kcache, vcache, cap, pos0,
                       npos, batch, out, heads);
  } else {
    hipLaunchKernelGGL(self_attention_step<false>, dim3(batch * heads), dim3(64), 0, s, qkv, kcache, vcache, cap, pos0,
                       npos, batch, out, heads);
  }
}

template <int NQ, bool BF>
static void launch_cross_nq(const CrossAttnArgs& a, hipStream_t s) {
  const dim3 grid(a.batch * a.heads * a.chunks);
  switch (a.heads * 64) {
    case 128: hipLaunchKernelGGL((cross_attention_step<NQ, 128, BF>), grid, dim3(256), 0, s, a.x, a.ln_g, a.ln_b, a.wq_t, a.bq, a.kc, a.vc, a.ws, a.batch, a.heads, a.T, a.chunks); break;
    case 384: hipLaunchKernelGGL((cross_attention_step<NQ, 384, BF>), grid, dim3(256), 0, s, a.x, a.ln_g, a.ln_b, a.wq_t, a.bq, a.kc, a.vc, a.ws, a.batch, a.heads, a.T, a.chunks); break;
    case 512: hipLaunchKernelGGL((cross_attention_step<NQ, 512, BF>), grid, dim3(256), 0, s, a.x, a.ln_g, a.ln_b, a.wq_t, a.bq, a.kc, a.vc, a.ws, a.batch, a.heads, a.T, a.chunks); break;
    default: throw Error(kErrFormat, "decoder kernels support d_model 128, 384 or 512");
  }
}

template <bool BF>
static void launch_cross_bf(const CrossAttnArgs& a, hipStream_t s) {
  switch (a.nq) {
    case 1: launch_cross_nq<1, BF>(a, s); break;
    case 2: launch_cross_nq<2, BF>(a, s); break;
    case 3: launch_cross_nq<3, BF>(a, s); break;
    case 4: launch_cross_nq<4, BF>(a, s); break;
    default: throw Error(kErrInvalidArg, "cross attention: 1..4 query rows per clip");
  }
}

void launch_cross_attention(const CrossAttnArgs& a, hipStream_t s) {
  if (a.batch < 1 || a.T < 1 || a.chunks < 1 || a.chunks > a.T) throw Error(kErrInvalidArg, "cross attention: bad shape");
  if (a.bf16) {
    launch_cross_bf<true>(a, s);
  } else {
    launch_cross_bf<false>(a, s);
  }
}

}  // namespace wt

// Encoder self-attention on pre-split operands (gfx950): softmax(q k^T / 8) v per (clip, head), S = 1500, d_head 64,
// non-causal, flash-style.  "Swapped" products S^T = K . Q^T and O^T += V^T . P^T, both contractions as three
// v_mfma_f32_16x16x32_f16 plane products with fp32 accumulation; q, k and v ARRIVE as two fp16 planes each (written by
// the qkv GEMM's epilogue, q already multiplied by d_head^-1/2 * log2(e)), so that
//   * a K/V tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4 in its SGPR-base form: no staging registers, no
//     ds_write, no vector instruction in front of a load).  K is double-buffered, V single: 48 KB per block and < 168
//     VGPRs, so THREE blocks share a CU;
//   * V stays row-major [key][d] in LDS and its transposed MFMA fragments come from ds_read_b64_tr_b16, the hardware
//     transpose read of gfx950 (each 16-lane group fetches a 4-key x 16-d block column-major);
//   * both LDS images are XOR-swizzled so that the fragment reads are conflict-free — K rows (128 B): 16-byte chunk ^=
//     (key >> 1) & 7 (ds_read_b128 in 16-lane groups: 16 keys x one chunk); V rows: 32-byte unit ^= (key >> 1) & 3 (a
//     32-lane half of a transposed read covers eight consecutive keys x 32 B = one 256-byte bank row) — applied to the
//     per-lane SOURCE address of the DMA, whose LDS destination is lane-linear;
//   * the running maximum is only raised when a tile's maximum exceeds it by more than kDefer = 3 (in the log2
//     domain): probabilities then reach at most 2^3, their fp16 planes 2^(12 + 3) < 65504, and the multiplies that
//     rescale O disappear from almost every tile;
//   * the result leaves as two fp16 planes scaled for the out-projection GEMM.
// Only the probabilities P = exp2(S^T - m) are split in registers (they are born there).
//
// Who holds what (round 4: 16 x 16 x 32 products; rounds 2-3 used 32 x 32 x 16 ones with a query row on two lanes —
// the matrix pipe is power-bound, DESIGN.md 4.1, and the smaller shape does the same arithmetic on less energy: 327.7
// against 332.6 us per layer in bursts, 151.2 against 149.3 k audio-sec/s end to end, tools/ab_attn.sh on the two builds).
// Lane (c = l & 15, g = l >> 4); a wave's 32 queries are two groups qa of 16, query 16 qa + c on lanes c + 16 g (four
// lanes per query):
//   S^T tile (key group kg of 16, qa) = K_kg . Q_qa^T: A = K rows (lane: key 16 kg + c, d chunk g of the k-step), B = Q;
//     the lane holds keys 16 kg + 4 g + r (r = 0..3) of query c;
//   P^T as the B operand of O^T += V^T . P^T over 32 keys (key groups 2 kp, 2 kp + 1): the lane's eight values are
//     its four of each group, unmoved — k index (g, j) = key 32 kp + 16 (j >> 2) + 4 g + (j & 3) — and the V^T A operand
//     takes the same keys with two transposed reads (4 keys x 16 d each) per d group dg.
// The four lanes of a row keep the same running maximum, their own partial row sums (added once after the last tile)
// and compare their OWN scores with the running maximum: the common tile has no cross-lane traffic at all (the row
// maximum is formed, by two LDS shuffles, only in a tile that raises it).
// Tried and not kept: v_permlane32_swap instead of the LDS shuffle (1.5 % slower); scaling the scores first with packed
// multiplies, v_max3 chains and packed adds — 135 instead of 180 vector instructions per tile — 0-4 % slower; round 3's
// ring of three tile buffers and inline-asm v_max3 (tools/experiments/README.md).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef short i16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int AK = 64;              // keys per tile
constexpr int kPlaneBytes = AK * 128;  // one plane of a K or V tile in LDS

// BF: bf16 storage mode — q, k, v and the output are ONE bf16 plane each (plane offsets unused), one
// v_mfma_f32_16x16x32_bf16 per product, probabilities rounded to bf16; s_inv then carries the softmax scale itself
// (d_head^-1/2 * log2 e: the qkv GEMM does not pre-scale q in that mode) and o_scale is 1.
constexpr float kDefer = 3.0f;   // log2 of the factor a tile's maximum may exceed the running maximum by without a rescale
constexpr float kPShift = 12.0f;  // probabilities are scaled by 2^12 before their fp16 split: 2^(12 + kDefer) < 65504


using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool BF>
__global__ __launch_bounds__(256, BF ? 4 : 3) void encoder_attention_planes(const _Float16* __restrict__ qkv, long plane,
                                                                     _Float16* __restrict__ out, long out_plane, int T,
                                                                     int heads, float s_inv, float o_scale) {
  constexpr int NP = BF ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int kVOff = 2 * NP * kPlaneBytes;
  using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

  const int d_model = heads * 64, ld = 3 * d_model;
  const int q_blocks = (T + 127) / 128;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / q_blocks, qb = logical % q_blocks;
  const int b = bh / heads, h = bh % heads;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const _Float16* base = qkv + (long)b * T * ld + h * 64;

  // Q fragments: lane (query 16 qa + c, chunk g) holds Q[q][32 ks + 8 g .. + 7]
  half8 qh[2][2], ql[2][2];
#pragma unroll
  for (int qa = 0; qa < 2; ++qa) {
    const int q_row = qb * 128 + wid * 32 + 16 * qa + c;
    const int q_ld = q_row < T ? q_row : T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qh[qa][ks] = *reinterpret_cast<const half8*>(base + (long)q_ld * ld + 32 * ks + 8 * g);
      if (!BF) ql[qa][ks] = *reinterpret_cast<const half8*>(base + plane + (long)q_ld * ld + 32 * ks + 8 * g);
    }
  }

  f32x4 o[4][2];  // O^T tiles: d = 16 dg + 4 g + r of query 16 qa + c
#pragma unroll
  for (int dg = 0; dg < 4; ++dg)
#pragma unroll
    for (int qa = 0; qa < 2; ++qa) o[dg][qa] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float m_run[2] = {0.0f, 0.0f}, l_run[2] = {0.0f, 0.0f};  // m_run: set from the first tile

  // LDS-DMA of a K or V tile: one wave-instruction copies 8 rows x 128 B (1 KiB) of one plane; wave w issues row groups
  // w and w + 4 of every plane.  (uniform base) + (32-bit per-lane byte offset that never changes): row drow (+ 32) of
  // the tile, the q / k / v column block, the swizzled 16-byte chunk.  Only the sequence's last tile has rows past T: it
  // uses offsets clamped to row T - 1 (their scores are masked).
  const int drow = 8 * wid + (lane >> 3);
  const int kch = (lane & 7) ^ ((drow >> 1) & 7);
  const int vch = (lane & 7) ^ (((drow >> 1) & 3) << 1);
  const unsigned row_b = 2u * (unsigned)ld;
  const unsigned offK0 = (unsigned)drow * row_b + 2u * (unsigned)(d_model + kch * 8), offK1 = offK0 + 32u * row_b;
  const unsigned offV0 = (unsigned)drow * row_b + 2u * (unsigned)(2 * d_model + vch * 8), offV1 = offV0 + 32u * row_b;
  const int last0 = ((T + AK - 1) / AK - 1) * AK;
  const unsigned c0r = (unsigned)((last0 + drow < T ? last0 + drow : T - 1) - last0);
  const unsigned c1r = (unsigned)((last0 + drow + 32 < T ? last0 + drow + 32 : T - 1) - last0);
  const unsigned offK0c = c0r * row_b + 2u * (unsigned)(d_model + kch * 8), offK1c = c1r * row_b + 2u * (unsigned)(d_model + kch * 8);
  const unsigned offV0c = c0r * row_b + 2u * (unsigned)(2 * d_model + vch * 8), offV1c = c1r * row_b + 2u * (unsigned)(2 * d_model + vch * 8);
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  auto dma_tile = [&](const bool is_v, int kt, unsigned char* dst) {
    const unsigned long long sb = reinterpret_cast<unsigned long long>(base) + (unsigned long long)kt * (unsigned long long)(AK * 2) * (unsigned long long)ld;
    const unsigned d = lds0 + (unsigned)(dst - lds) + (unsigned)wid * 1024u;
    const bool lastp = kt * AK + AK > T;
#pragma unroll
    for (int p = 0; p < (BF ? 1 : 2); ++p) {
      const unsigned long long sbp = sb + (unsigned long long)p * 2ull * (unsigned long long)plane;
      if (lastp) {
        lds_dma16_sgpr(is_v ? offV0c : offK0c, sbp, d + p * kPlaneBytes);
        lds_dma16_sgpr(is_v ? offV1c : offK1c, sbp, d + p * kPlaneBytes + 4096);
      } else {
        lds_dma16_sgpr(is_v ? offV0 : offK0, sbp, d + p * kPlaneBytes);
        lds_dma16_sgpr(is_v ? offV1 : offK1, sbp, d + p * kPlaneBytes + 4096);
      }
    }
  };
  // fragment addresses.  K: lane (key 16 kg + c, chunk 4 ks + g): the swizzle term (key >> 1) & 7 = (c >> 1) & 7
  const int kfx = (c >> 1) & 7;
  // V^T by transposed reads: lane i = c of group g supplies the address of key row 4 g + (i >> 2) (+ 32 kp + 16 h),
  // d columns 16 dg + 4 (i & 3) .. + 3: 16-byte chunk 2 dg + ((i & 3) >> 1), byte 8 (i & 1) inside it; the swizzle term
  // ((key >> 1) & 3) << 1 depends on the lane only
  const int vrow = 4 * g + (c >> 2);
  const int vsw = ((vrow >> 1) & 3) << 1;
  const unsigned char* const vlane = lds + kVOff + vrow * 128 + ((c & 1) << 3);
  const int vcp = (c & 3) >> 1;

  const int n_tiles = (T + AK - 1) / AK;
  dma_tile(false, 0, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  dma_tile(true, 0, lds + kVOff);
  if (n_tiles > 1) dma_tile(false, 1, lds + NP * kPlaneBytes);
  for (int kt = 0; kt < n_tiles; ++kt) {
    const unsigned char* const kb = lds + (kt & 1) * NP * kPlaneBytes;
    f32x4 sacc[4][2];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
#pragma unroll
      for (int qa = 0; qa < 2; ++qa) sacc[kg][qa] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned char* kp_ = kb + (16 * kg + c) * 128 + (((4 * ks + g) ^ kfx) << 4);
        const half8 kh = *reinterpret_cast<const half8*>(kp_);
        if constexpr (BF) {
#pragma unroll
          for (int qa = 0; qa < 2; ++qa)
            sacc[kg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kh), __builtin_bit_cast(bf16x8, qh[qa][ks]), sacc[kg][qa], 0, 0, 0);
        } else {
          const half8 kl = *reinterpret_cast<const half8*>(kp_ + kPlaneBytes);
#pragma unroll
          for (int qa = 0; qa < 2; ++qa) {
            sacc[kg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[qa][ks], sacc[kg][qa], 0, 0, 0);
            sacc[kg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[qa][ks], sacc[kg][qa], 0, 0, 0);
            sacc[kg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[qa][ks], sacc[kg][qa], 0, 0, 0);
          }
        }
      }
    }
    if ((kt + 1) * AK > T) {  // last tile: keys past T do not exist
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * AK + 16 * kg + 4 * g + r >= T) sacc[kg][0][r] = sacc[kg][1][r] = -1e30f;
    }
    // online softmax, per query group.  The scores still carry the operand scales of the planes (s_inv > 0 takes them
    // out); the probabilities carry 2^12 (fp16's normal range for their low plane), l_run and O carry it too, and it
    // cancels in O / l.  Deferred maximum: raise m_run (and rescale O, l) only when some lane's scores exceed it by more
    // than kDefer; otherwise the probabilities of this tile are at most 2^kDefer, which the planes hold.
    // The exponent t = s * s_inv + (2^12 shift) - m_run is formed FIRST, against the running maximum as it stands (one fma
    // per score; the maximum of fma results needs no canonicalising v_max x, x, x per score, which the maximum of raw
    // MFMA results did): t <= shift + kDefer in every lane is the common tile, and exp2(t) follows at once.
    const float t_cap = (BF ? 0.0f : kPShift) + kDefer;
    if (kt == 0) {  // the running maximum starts as the first tile's row maximum (an exponent formed against "minus infinity" would lose the score)
      // (hipcc if-converts this block — its maxima and four shuffles run in every tile behind a select; taking it out of
      // the loop, the running maximum starting from the row's score against key 0, measured 2.5 % SLOWER: 315 against 308 us)
#pragma unroll
      for (int qa = 0; qa < 2; ++qa) {
        float m = fmaxf(fmaxf(sacc[0][qa][0], sacc[0][qa][1]), fmaxf(sacc[0][qa][2], sacc[0][qa][3]));
#pragma unroll
        for (int kg = 1; kg < 4; ++kg)
          m = fmaxf(m, fmaxf(fmaxf(sacc[kg][qa][0], sacc[kg][qa][1]), fmaxf(sacc[kg][qa][2], sacc[kg][qa][3])));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        m_run[qa] = m * s_inv;
      }
    }
    float lmax[2];
#pragma unroll
    for (int qa = 0; qa < 2; ++qa) {
      const float sh0 = (BF ? 0.0f : kPShift) - m_run[qa];
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[kg][qa][r] = fmaf(sacc[kg][qa][r], s_inv, sh0);
      lmax[qa] = fmaxf(fmaxf(sacc[0][qa][0], sacc[0][qa][1]), fmaxf(sacc[0][qa][2], sacc[0][qa][3]));
#pragma unroll
      for (int kg = 1; kg < 4; ++kg)
        lmax[qa] = fmaxf(lmax[qa], fmaxf(fmaxf(sacc[kg][qa][0], sacc[kg][qa][1]), fmaxf(sacc[kg][qa][2], sacc[kg][qa][3])));
    }
    if (__any(lmax[0] > t_cap || lmax[1] > t_cap)) {
#pragma unroll
      for (int qa = 0; qa < 2; ++qa) {
        float tmax = fmaxf(lmax[qa], __shfl_xor(lmax[qa], 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // the row's maximum in score units is m_run + tmax - shift: the new running maximum, and what every exponent drops by
        const float delta = fminf(0.0f, (BF ? 0.0f : kPShift) - tmax);  // m_run - m_new
        const float alpha = __builtin_amdgcn_exp2f(delta);
        l_run[qa] *= alpha;
#pragma unroll
        for (int dg = 0; dg < 4; ++dg) o[dg][qa] *= alpha;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[kg][qa][r] += delta;
        m_run[qa] -= delta;
      }
    }
#pragma unroll
    for (int qa = 0; qa < 2; ++qa) {
      float psum = 0.0f;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sacc[kg][qa][r] = __builtin_amdgcn_exp2f(sacc[kg][qa][r]);
          psum += sacc[kg][qa][r];
        }
      l_run[qa] += psum;  // this lane's keys only
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // O^T += V^T . P^T, 32 keys per step
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      half8 ph[2], pl[2];
#pragma unroll
      for (int qa = 0; qa < 2; ++qa) {
        u32x4 phu, plu;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4& sp = sacc[2 * kp + (e >> 1)][qa];
          unsigned hh, ll = 0;
          if constexpr (BF) {
            hh = pack_bf16x2(sp[2 * (e & 1)], sp[2 * (e & 1) + 1]);
          } else {
            split_f16x2(sp[2 * (e & 1)], sp[2 * (e & 1) + 1], &hh, &ll);
          }
          phu[e] = hh;
          plu[e] = ll;
        }
        ph[qa] = __builtin_bit_cast(half8, phu);
        pl[qa] = __builtin_bit_cast(half8, plu);
      }
#pragma unroll
      for (int dg = 0; dg < 4; ++dg) {
        half8 vh, vl;
#pragma unroll
        for (int hh2 = 0; hh2 < 2; ++hh2) {
          const unsigned char* r0 = vlane + (32 * kp + 16 * hh2) * 128 + (((2 * dg + vcp) ^ vsw) << 4);
          const i16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(r0));
          const half4 a0h = __builtin_bit_cast(half4, a0);
#pragma unroll
          for (int e = 0; e < 4; ++e) vh[4 * hh2 + e] = a0h[e];
          if constexpr (!BF) {
            const i16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(r0 + kPlaneBytes));
            const half4 a1h = __builtin_bit_cast(half4, a1);
#pragma unroll
            for (int e = 0; e < 4; ++e) vl[4 * hh2 + e] = a1h[e];
          }
        }
#pragma unroll
        for (int qa = 0; qa < 2; ++qa) {
          if constexpr (BF) {
            o[dg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vh), __builtin_bit_cast(bf16x8, ph[qa]), o[dg][qa], 0, 0, 0);
          } else {
            o[dg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[qa], o[dg][qa], 0, 0, 0);
            o[dg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[qa], o[dg][qa], 0, 0, 0);
            o[dg][qa] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[qa], o[dg][qa], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < n_tiles) dma_tile(true, kt + 1, lds + kVOff);
    if (kt + 2 < n_tiles) dma_tile(false, kt + 2, lds + (kt & 1) * NP * kPlaneBytes);
  }

#pragma unroll
  for (int qa = 0; qa < 2; ++qa) {
    float l = l_run[qa];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int q_row = qb * 128 + wid * 32 + 16 * qa + c;
    if (q_row < T) {
      const float inv = o_scale / l;
      _Float16* orow = out + ((long)b * T + q_row) * d_model + h * 64;
      using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
#pragma unroll
      for (int dg = 0; dg < 4; ++dg) {
        unsigned ah[2], al[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (BF) {
            ah[j] = pack_bf16x2(o[dg][qa][2 * j] * inv, o[dg][qa][2 * j + 1] * inv);
          } else {
            split_f16x2(o[dg][qa][2 * j] * inv, o[dg][qa][2 * j + 1] * inv, &ah[j], &al[j]);
          }
        }
        *reinterpret_cast<u32x2*>(orow + 16 * dg + 4 * g) = u32x2{ah[0], ah[1]};
        if constexpr (!BF) *reinterpret_cast<u32x2*>(orow + out_plane + 16 * dg + 4 * g) = u32x2{al[0], al[1]};
      }
    }
  }
}

}  // namespace

void launch_encoder_attention_planes(const unsigned short* qkv, long plane, unsigned short* out, long out_plane,
                                     int batch, int T, int heads, float q_scale, float k_scale, float v_scale,
                                     float out_scale, hipStream_t stream) {
  if (batch < 1 || T < 1 || heads < 1 || (3 * heads * 64) % 8 != 0) throw Error(kErrInvalidArg, "encoder attention: bad shape");
  const int q_blocks = (T + 127) / 128;
  WT_LAUNCH_TIMED(encoder_attention_planes<false>, dim3(batch * heads * q_blocks), dim3(256), 6 * kPlaneBytes, stream,
                     reinterpret_cast<const _Float16*>(qkv), plane, reinterpret_cast<_Float16*>(out), out_plane, T, heads,
                     1.0f / (q_scale * k_scale), out_scale / v_scale);
}

void launch_encoder_attention_bf16(const unsigned short* qkv, unsigned short* out, int batch, int T, int heads,
                                   hipStream_t stream) {
  if (batch < 1 || T < 1 || heads < 1) throw Error(kErrInvalidArg, "encoder attention: bad shape");
  const int q_blocks = (T + 127) / 128;
  WT_LAUNCH_TIMED(encoder_attention_planes<true>, dim3(batch * heads * q_blocks), dim3(256), 3 * kPlaneBytes, stream,
                     reinterpret_cast<const _Float16*>(qkv), 0L, reinterpret_cast<_Float16*>(out), 0L, T, heads,
                     0.125f * 1.44269504088896340736f, 1.0f);
}

}  // namespace wt

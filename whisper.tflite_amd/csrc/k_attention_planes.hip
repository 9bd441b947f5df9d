// Encoder self-attention on pre-split operands (gfx950): softmax(q k^T / 8) v per (clip, head), S = 1500, d_head 64,
// non-causal, flash-style.  Same algorithm and block shape as encoder_attention_split<4, 2> (k_attention.hip) —
// "swapped" products S^T = K . Q^T and O^T += V^T . P^T so that a softmax row sits on a lane, both contractions
// as three v_mfma_f32_32x32x16_f16 plane products with fp32 accumulation — but q, k and v ARRIVE as two fp16
// planes each (written by the qkv GEMM's epilogue, q already multiplied by d_head^-1/2 * log2(e)), so that
//   * a K/V tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no VALU;
//     round 2 moved it through 32 VGPRs + ds_write_b128, round 1 re-split every element on the VALU in each of the 12
//     query blocks that stream it).  K is double-buffered, V single: 48 KB per block and 164 VGPRs, so THREE blocks
//     share a CU (three wavefronts per SIMD to overlap one's softmax with another's MFMAs; round 2: two);
//   * V stays row-major [key][d] in LDS and its transposed MFMA fragments come from ds_read_b64_tr_b16, the hardware
//     transpose read of gfx950 (each 16-lane group fetches a 4-key x 16-d block column-major);
//   * both LDS images are XOR-swizzled in 16-byte chunks so that the fragment reads are conflict-free:
//     K rows (128 B): chunk ^= (key >> 1) & 7 (ds_read_b128, 16-lane groups); V rows: chunk ^= ((key >> 1) & 1) << 2
//     (a 32-lane half of the tr read covers 4 keys x 64 B = one 256-byte bank row) — applied to the per-lane SOURCE
//     address of the DMA, whose LDS destination is lane-linear;
//   * the running maximum is only raised when a tile's maximum exceeds it by more than kDefer = 3 (in the log2
//     domain): probabilities then reach at most 2^3, their fp16 planes 2^(12 + 3) < 65504, and the 32 multiplies per
//     lane that rescale O disappear from almost every tile (with 32 query rows per wavefront SOME row used to raise
//     its maximum in most tiles);
//   * the result leaves as two fp16 planes scaled for the out-projection GEMM.
// Only the probabilities P = exp2(S^T - m) are split in registers (they are born there).
#include <hip/hip_runtime.h>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef short i16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int AK = 64;              // keys per tile
constexpr int kPlaneBytes = AK * 128;  // one plane of a K or V tile in LDS

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#define WT_MM16(A, B, ACC) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, ACC, 0, 0, 0)

// BF: bf16 storage mode — q, k, v and the output are ONE bf16 plane each (plane offsets unused), one
// v_mfma_f32_32x32x16_bf16 per product, probabilities rounded to bf16; s_inv then carries the softmax scale itself
// (d_head^-1/2 * log2 e: the qkv GEMM does not pre-scale q in that mode) and o_scale is 1.
constexpr float kDefer = 3.0f;   // log2 of the factor a tile's maximum may exceed the running maximum by without a rescale
constexpr float kPShift = 12.0f;  // probabilities are scaled by 2^12 before their fp16 split: 2^(12 + kDefer) < 65504

template <bool BF>
__global__ __launch_bounds__(256, BF ? 4 : 3) void encoder_attention_planes(const _Float16* __restrict__ qkv, long plane,
                                                                   _Float16* __restrict__ out, long out_plane, int T,
                                                                   int heads, float s_inv, float o_scale) {
  // [K0 hi][K0 lo][K1 hi][K1 lo][V hi][V lo], 8 KB each (K double-buffered): 48 KB, three blocks per CU; bf16 mode
  // has one plane per tensor ([K0][K1][V]: 24 KB, four blocks per CU)
  // (dynamic LDS: with a static __shared__ array hipcc knows that the DMA in flight writes the array its ds_reads
  // come from and puts an s_waitcnt vmcnt(0) in front of them, which would wait for every prefetch right at its issue)
  constexpr int NP = BF ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int kVOff = 2 * NP * kPlaneBytes;
  using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

  const int d_model = heads * 64, ld = 3 * d_model;
  const int q_blocks = (T + 127) / 128;
  // consecutive blocks on one XCD (blockIdx % 8 equal) walk the q-blocks of one (clip, head), so its K/V stay in
  // that XCD's L2
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / q_blocks, qb = logical % q_blocks;
  const int b = bh / heads, h = bh % heads;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const _Float16* base = qkv + (long)b * T * ld + h * 64;

  // Q planes: lane (q = l31, half lh) holds Q[q][16c + 8lh + 0..7] for k-step c
  const int q_row = qb * 128 + wid * 32 + l31;
  const int q_ld = q_row < T ? q_row : T - 1;
  half8 qh[4], ql[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    qh[c] = *reinterpret_cast<const half8*>(base + (long)q_ld * ld + 16 * c + 8 * lh);
    if (!BF) ql[c] = *reinterpret_cast<const half8*>(base + plane + (long)q_ld * ld + 16 * c + 8 * lh);
  }

  f32x16 o0, o1;  // O^T tiles: d in [0,32) and [32,64)
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  // LDS-DMA of a K or V tile: one wave-instruction copies 8 rows x 128 B (1 KiB) of one plane; LDS slot (row
  // lane >> 3, chunk lane & 7) of instruction (plane p, row group g) is base + p * 8 KB + g * 1 KB + lane * 16 (the
  // destination is lane-linear), and takes the global chunk (lane & 7) ^ swizzle(row) of that row.  Wave w issues
  // (p, g) = (0, w), (0, w + 4) and, with two planes, (1, w), (1, w + 4): rows 8 w + (lane >> 3) and 32 more.
  const int drow = 8 * wid + (lane >> 3);                       // + 32 for the second row group
  const int kch = (lane & 7) ^ ((drow >> 1) & 7);               // (row + 32) >> 1 has the same low three bits
  const int vch = (lane & 7) ^ (((drow >> 1) & 1) << 2);
  // The loads are issued in the SGPR-base form of global_load_lds_dwordx4, written out (the builtin only selects the
  // per-lane 64-bit address form): base = this (clip, head)'s rows of tile kt (+ plane), advanced by scalar
  // instructions; the per-lane part is a 32-bit byte offset that never changes — row drow (+ 32) of the tile, the q / k /
  // v column block, the swizzled 16-byte chunk.  So NO vector instruction stands in front of a load: on this part a
  // wavefront's VALU instruction waits for the other wavefronts' MFMA bursts (tools/mfma_valu_overlap.hip), and the
  // row * ld multiplies, clamps and readfirstlanes of the per-lane form delayed every prefetch by such a burst.
  // Only the sequence's last tile has rows past T: it uses offsets clamped to row T - 1 (their scores are masked).
  const unsigned row_b = 2u * (unsigned)ld;  // bytes per row
  const unsigned offK0 = (unsigned)drow * row_b + 2u * (unsigned)(d_model + kch * 8), offK1 = offK0 + 32u * row_b;
  const unsigned offV0 = (unsigned)drow * row_b + 2u * (unsigned)(2 * d_model + vch * 8), offV1 = offV0 + 32u * row_b;
  const int last0 = ((T + AK - 1) / AK - 1) * AK;
  const unsigned c0r = (unsigned)((last0 + drow < T ? last0 + drow : T - 1) - last0);
  const unsigned c1r = (unsigned)((last0 + drow + 32 < T ? last0 + drow + 32 : T - 1) - last0);
  const unsigned offK0c = c0r * row_b + 2u * (unsigned)(d_model + kch * 8), offK1c = c1r * row_b + 2u * (unsigned)(d_model + kch * 8);
  const unsigned offV0c = c0r * row_b + 2u * (unsigned)(2 * d_model + vch * 8), offV1c = c1r * row_b + 2u * (unsigned)(2 * d_model + vch * 8);
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  auto dma16 = [&](unsigned voff, unsigned long long sb, unsigned dst) { lds_dma16_sgpr(voff, sb, dst); };
  auto dma_tile = [&](const bool is_v, int kt, unsigned char* dst) {
    const unsigned long long sb = reinterpret_cast<unsigned long long>(base) + (unsigned long long)kt * (unsigned long long)(AK * 2) * (unsigned long long)ld;
    const unsigned d = lds0 + (unsigned)(dst - lds) + (unsigned)wid * 1024u;
    const bool lastp = kt * AK + AK > T;
#pragma unroll
    for (int p = 0; p < (BF ? 1 : 2); ++p) {
      const unsigned long long sbp = sb + (unsigned long long)p * 2ull * (unsigned long long)plane;
      if (lastp) {
        dma16(is_v ? offV0c : offK0c, sbp, d + p * kPlaneBytes);
        dma16(is_v ? offV1c : offK1c, sbp, d + p * kPlaneBytes + 4096);
      } else {
        dma16(is_v ? offV0 : offK0, sbp, d + p * kPlaneBytes);
        dma16(is_v ? offV1 : offK1, sbp, d + p * kPlaneBytes + 4096);
      }
    }
  };
  // fragment addresses.  K: lane (key l31 (+32), half lh), k-step c -> chunk 2c + lh of its row.
  const int kfx = (l31 >> 1) & 7;
  // V^T via transposed reads: 16-lane group gi = lane >> 4 covers d = 32 dt + 16 (gi & 1) + (lane & 15) and keys
  // 16 s + 8 ri + 4 lh + 0..3; lane 4q + p of the group supplies the address of key row q, d columns 4p .. 4p + 3
  const int vq = (lane >> 2) & 3, vp = lane & 3, vg = (lane >> 4) & 1;

  const int n_tiles = (T + AK - 1) / AK;
  // Schedule per tile t (two barriers, the loads never waited for right after their issue):
  //   QK_t from K buffer t & 1 | softmax | vmcnt(0) + barrier: V_t and K_t+1 have landed, everybody is done with K_t
  //   | PV_t | barrier: everybody is done with V_t -> issue V_t+1, and K_t+2 into buffer t & 1
  dma_tile(false, 0, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K_0 (and this wave's Q fragments)
  __builtin_amdgcn_s_barrier();
  dma_tile(true, 0, lds + kVOff);
  if (n_tiles > 1) dma_tile(false, 1, lds + NP * kPlaneBytes);
  for (int kt = 0; kt < n_tiles; ++kt) {
    const unsigned char* const kb = lds + (kt & 1) * NP * kPlaneBytes;
    // S^T for the two 32-key halves of the tile
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int slot = ((2 * c + lh) ^ kfx) << 4;
      const half8 k0h = *reinterpret_cast<const half8*>(kb + l31 * 128 + slot);
      const half8 k1h = *reinterpret_cast<const half8*>(kb + (32 + l31) * 128 + slot);
      if constexpr (BF) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k0h), __builtin_bit_cast(bf16x8, qh[c]), s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k1h), __builtin_bit_cast(bf16x8, qh[c]), s1, 0, 0, 0);
        continue;
      }
      const half8 k0l = *reinterpret_cast<const half8*>(kb + kPlaneBytes + l31 * 128 + slot);
      const half8 k1l = *reinterpret_cast<const half8*>(kb + kPlaneBytes + (32 + l31) * 128 + slot);
      s0 = WT_MM16(k0h, ql[c], s0);
      s0 = WT_MM16(k0l, qh[c], s0);
      s0 = WT_MM16(k0h, qh[c], s0);
      s1 = WT_MM16(k1h, ql[c], s1);
      s1 = WT_MM16(k1l, qh[c], s1);
      s1 = WT_MM16(k1h, qh[c], s1);
    }
    if ((kt + 1) * AK > T) {  // last tile: keys past T do not exist
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kt * AK + crow(r, lh) >= T) s0[r] = -1e30f;
        if (kt * AK + 32 + crow(r, lh) >= T) s1[r] = -1e30f;
      }
    }
    // online softmax; the row (query) lives on lanes l and l ^ 32.  The scores still carry the operand scales of
    // the planes (s_inv > 0 takes them out): the maximum is taken on the raw values, and s_inv, the running maximum
    // and the 2^12 that puts the probabilities into fp16's normal range all go into ONE fma per score in front of
    // the exp2.  l_run and O then both carry the 2^12, which cancels in O / l.
    // (round 4) No cross-lane traffic in the common tile: the lane compares the maximum of ITS 32 scores with the running
    // maximum (kept equal in both lanes of a row) — only when some lane of the wave exceeds it by more than kDefer is
    // the row maximum formed (one LDS shuffle) and O, l rescaled; the row sum stays a per-lane partial until the end.
    // (tried and not kept: v_permlane32_swap instead of the LDS shuffle, 1.5 % slower; scaling the scores first with
    // packed multiplies, v_max3 chains and packed adds — 135 instead of 180 vector instructions per tile — 0-4 % slower,
    // tools/ab_attn.sh: the loop is not bound by the count of vector instructions)
    float lmax = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) lmax = fmaxf(lmax, fmaxf(s0[r], s1[r]));
    lmax *= s_inv;
    // deferred maximum: raise m_run (and rescale O, l) only when some row's tile maximum exceeds it by more than
    // kDefer; otherwise the probabilities of this tile are at most 2^kDefer, which the planes hold
    if (__any(lmax > m_run + kDefer)) {
      const float tmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
      m_run = m_new;
    }
    const float shift = (BF ? 0.0f : kPShift) - m_run;
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], s_inv, shift));
      s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], s_inv, shift));
      psum += s0[r] + s1[r];
    }
    l_run += psum;  // this lane's keys only: the two halves of a row are added once, after the last tile
    // V_t and K_t+1 have landed (this wave's parts: vmcnt; everybody's: the barrier), and every wave is done with K_t
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // O^T += V^T . P^T, 16 keys per step: registers 8 s2 .. 8 s2 + 7 of S^T are, unmoved, the B fragment
    auto pv_half = [&](const f32x16& sp, const int hf) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 phu, plu;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned hh, ll = 0;
          if constexpr (BF) {
            hh = pack_bf16x2(sp[8 * s2 + 2 * e], sp[8 * s2 + 2 * e + 1]);
          } else {
            split_f16x2(sp[8 * s2 + 2 * e], sp[8 * s2 + 2 * e + 1], &hh, &ll);
          }
          phu[e] = hh;
          plu[e] = ll;
        }
        const half8 ph = __builtin_bit_cast(half8, phu), pl = __builtin_bit_cast(half8, plu);
        // element j of lane half lh of that fragment is key 16 s2 + 8 (j >> 2) + 4 lh + (j & 3) (the C/D row map of
        // the S^T accumulator): the V^T fragment takes its keys in the same order
        const int key0 = hf * 32 + 16 * s2 + 4 * lh;
        half8 v0h, v0l, v1h, v1l;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
          const int key = key0 + 8 * ri + vq;
          const int vsw = ((key >> 1) & 1) << 2;
          // d tile 0: chunk = 2 vg + (vp >> 1) (+ 4 for d tile 1), byte 8 (vp & 1) inside the chunk
          const unsigned char* r0 = lds + kVOff + key * 128 + ((vp & 1) << 3);
          const int c0 = ((2 * vg + (vp >> 1)) ^ vsw) << 4, c1 = ((4 + 2 * vg + (vp >> 1)) ^ vsw) << 4;
          const i16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(r0 + c0));
          const i16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(r0 + c1));
          const half4 a0h = __builtin_bit_cast(half4, a0), b0h = __builtin_bit_cast(half4, b0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0h[4 * ri + e] = a0h[e];
            v1h[4 * ri + e] = b0h[e];
          }
          if constexpr (!BF) {
            const i16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(r0 + kPlaneBytes + c0));
            const i16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(r0 + kPlaneBytes + c1));
            const half4 a1h = __builtin_bit_cast(half4, a1), b1h = __builtin_bit_cast(half4, b1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0l[4 * ri + e] = a1h[e];
              v1l[4 * ri + e] = b1h[e];
            }
          }
        }
        if constexpr (BF) {
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0h), __builtin_bit_cast(bf16x8, ph), o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1h), __builtin_bit_cast(bf16x8, ph), o1, 0, 0, 0);
        } else {
          o0 = WT_MM16(v0h, pl, o0);
          o0 = WT_MM16(v0l, ph, o0);
          o0 = WT_MM16(v0h, ph, o0);
          o1 = WT_MM16(v1h, pl, o1);
          o1 = WT_MM16(v1l, ph, o1);
          o1 = WT_MM16(v1h, ph, o1);
        }
      }
    };
    pv_half(s0, 0);
    pv_half(s1, 1);
    // every wave is done with V_t (its fragment reads were consumed by the MFMAs above); K_t+2 stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // (K_t+2 could go out one phase earlier, right after the barrier above; hipcc then puts an s_waitcnt vmcnt(0) of
    // its own in front of the V fragment reads — it cannot see that the DMA targets the other buffer — and the
    // prefetch would be waited for at its issue)
    if (kt + 1 < n_tiles) dma_tile(true, kt + 1, lds + kVOff);
    if (kt + 2 < n_tiles) dma_tile(false, kt + 2, lds + (kt & 1) * NP * kPlaneBytes);
  }

  l_run += __shfl_xor(l_run, 32, 64);
  if (q_row < T) {
    const float inv = o_scale / l_run;  // o_scale = out_scale / v_scale (the 2^12 of the probabilities is in l_run too)
    _Float16* orow = out + ((long)b * T + q_row) * d_model + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned ah[2], al[2], ch[2], cl[2];
      using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (BF) {
          ah[j] = pack_bf16x2(o0[4 * g + 2 * j] * inv, o0[4 * g + 2 * j + 1] * inv);
          ch[j] = pack_bf16x2(o1[4 * g + 2 * j] * inv, o1[4 * g + 2 * j + 1] * inv);
        } else {
          split_f16x2(o0[4 * g + 2 * j] * inv, o0[4 * g + 2 * j + 1] * inv, &ah[j], &al[j]);
          split_f16x2(o1[4 * g + 2 * j] * inv, o1[4 * g + 2 * j + 1] * inv, &ch[j], &cl[j]);
        }
      }
      *reinterpret_cast<u32x2*>(orow + 8 * g + 4 * lh) = u32x2{ah[0], ah[1]};
      *reinterpret_cast<u32x2*>(orow + 32 + 8 * g + 4 * lh) = u32x2{ch[0], ch[1]};
      if constexpr (!BF) {
        *reinterpret_cast<u32x2*>(orow + out_plane + 8 * g + 4 * lh) = u32x2{al[0], al[1]};
        *reinterpret_cast<u32x2*>(orow + out_plane + 32 + 8 * g + 4 * lh) = u32x2{cl[0], cl[1]};
      }
    }
  }
}
#undef WT_MM16

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

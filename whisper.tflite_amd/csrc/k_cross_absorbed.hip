// Decoder cross-attention with the key / value projections ABSORBED into the query and output sides (gfx950).
//
// The reference's decoder graph (whisper.tflite/whisper.cpp:375) computes, per layer and head h,
//   k_j = Wk_h e_j,  v_j = Wv_h e_j + bv_h   (e_j = encoder output row j),
//   o_h = sum_j softmax_j(q_h . k_j / 8) v_j.
// Rounds 1-2 cached k_j, v_j of all layers per clip (4 layers x 2 x 1500 x 384 fp32 = 18.4 MB) and streamed one
// layer's K and V (4.6 MB per clip) from HBM for every decoder position: 15.9 GB per 32-clip batch, 83 % of the
// decoder's CU time at the ~24 GB/s a CU can pull from HBM.  The same arithmetic, re-associated:
//   q_h . k_j = (Wk_h^T q_h) . e_j = q'_h . e_j            (q'_h in R^d_model, made by a small GEMM)
//   sum_j p_j v_j = Wv_h (sum_j p_j e_j) + bv_h = Wv_h c_h + bv_h
// needs only E itself — ONE [1500][384] matrix per clip, shared by all heads AND all layers (2.3 MB per clip and
// sweep instead of 4.6 MB, and no cross-KV projection GEMM in the encoder at all).  The price is arithmetic: heads x
// d_model instead of d_head multiply-adds per key element, which is why this kernel is a matrix-core kernel:
//   S^T[key][qc]  = E_tile . Q'^T        (qc = query column = position x head, <= 16 of them)
//   C^T[d][qc]   += E_tile^T . P^T       (P = exp2(S - m))
// on v_mfma_f32_16x16x32_f16 with every operand as two fp16 planes (three products, fp32 accumulation: the fp32-level
// contraction of csrc/bf16_split.h).  E arrives as planes (the encoder's final LayerNorm writes them, as it did for
// the cross-KV GEMM), Q' and P are split in registers.  10 % of the matrix pipe keeps up with the stream.
//
// Block = (clip, key chunk), 4 wavefronts.  Tiles of 32 keys go global -> LDS by LDS-DMA into a ring of NST stages
// (prefetch distance NST - 1 tiles); one image serves the row reads of the score product (ds_read_b128) and the
// transposed reads of the context product (ds_read_b64_tr_b16): [plane][panel of 128 columns][32 keys][256 B], the
// 32-byte units of a row XORed with key & 7 (conflict-free for both, MI355X_MICROARCH.md LDS table), the swizzle
// applied to the DMA's per-lane SOURCE address.  Wave w computes the scores of key sub-tile w & 1 over d-half w >> 1
// (the halves are added through LDS), every wave then holds all 32 x 16 probabilities and accumulates its own
// quarter of the d range.  Output per (row, head, chunk): unnormalised c[d_model], m, l — combined over chunks by
// cross_absorbed_combine, which also applies the head's value projection Wv_h c + bv_h; the ordinary cross
// out-projection follows.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

struct CrossAbsDev {
  const float* qp;       // [rows][heads * DM] absorbed queries (log2 domain: d_head^-1/2 * log2 e folded in)
  const _Float16* e;     // E planes: hi [clips * T][DM], lo at e + e_plane (bf16 storage mode: ONE bf16 plane)
  const _Float16* eg[4]; // clip b reads eg[b / split] at clip b % split: up to four encoder batches decoded by one chain
  int split;
  long e_plane;
  float e_scale;         // power of two baked into the planes
  float* ws;             // [rows][heads][chunks][DM + 4]: c[DM], m (natural log units), l, pad
  int B, H, T, chunks, nq, p0, tiles_per_chunk;
};

constexpr float kDeferA = 3.0f, kPShiftA = 12.0f;  // as in k_attention_planes.hip

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// Transposed reads of the E^T fragments of TWO 16-column blocks (4 keys x 16 columns each, hi and lo plane, two key
// blocks: eight ds_read_b64_tr_b16) and the wait for them, in ONE asm statement: as a builtin hipcc puts an
// s_waitcnt vmcnt(0) in front of the read while prefetch DMAs are in flight (it cannot tell that they target other
// ring stages); as separate asm statements the results would be "ready" for the compiler before they have landed.
struct TrFrag {
  u32x2 h0, h1, g0, g1;
};
__device__ __forceinline__ void ds_read_tr16_x2(unsigned a0, unsigned a1, unsigned b0, unsigned b1, unsigned plane, TrFrag& x,
                                                TrFrag& y) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\t"
      "ds_read_b64_tr_b16 %1, %9\n\t"
      "ds_read_b64_tr_b16 %2, %12\n\t"
      "ds_read_b64_tr_b16 %3, %13\n\t"
      "ds_read_b64_tr_b16 %4, %10\n\t"
      "ds_read_b64_tr_b16 %5, %11\n\t"
      "ds_read_b64_tr_b16 %6, %14\n\t"
      "ds_read_b64_tr_b16 %7, %15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(x.h0), "=&v"(x.h1), "=&v"(x.g0), "=&v"(x.g1), "=&v"(y.h0), "=&v"(y.h1), "=&v"(y.g0), "=&v"(y.g1)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(a0 + plane), "v"(a1 + plane), "v"(b0 + plane), "v"(b1 + plane)
      : "memory");
}

// the same for ONE plane (bf16 storage mode): four reads
__device__ __forceinline__ void ds_read_tr16_x2_one_plane(unsigned a0, unsigned a1, unsigned b0, unsigned b1, TrFrag& x, TrFrag& y) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4\n\t"
      "ds_read_b64_tr_b16 %1, %5\n\t"
      "ds_read_b64_tr_b16 %2, %6\n\t"
      "ds_read_b64_tr_b16 %3, %7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(x.h0), "=&v"(x.h1), "=&v"(y.h0), "=&v"(y.h1)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "memory");
}

// BF: bf16 storage mode (BASELINE configs[3]) — E is ONE bf16 plane, Q' and P are rounded to bf16 in registers, one
// v_mfma_f32_16x16x32_bf16 per product instead of three f16 ones, no operand scales (bf16 has the fp32 range).
#ifdef WT_ABS_STAMPS  // diagnostic build (tools/cross_abs_phase_probe.hip): where a block's cycles go
__device__ long long g_abs_stamps[4096 * 16];
#define ABS_ACC(i)                                    \
  do {                                                \
    const long long t__ = __builtin_readcyclecounter(); \
    ph_[i] += t__ - tl_;                              \
    tl_ = t__;                                        \
  } while (0)
#define ABS_STAMP(i) st_[i] = __builtin_readcyclecounter()
#else
#define ABS_STAMP(i) \
  do {               \
  } while (0)
#define ABS_ACC(i) \
  do {             \
  } while (0)
#endif

template <int DM, int NST, bool BF>
__global__ __launch_bounds__(320) void cross_absorbed_attention(CrossAbsDev a) {
  using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
  constexpr int P = DM / 128;                  // column panels of 128 halfs (256-byte LDS rows)
  constexpr int kPlane = P * 8192;             // bytes of one plane of a 32-key tile
  constexpr int NP = BF ? 1 : 2;
  constexpr int kStage = NP * kPlane;
  constexpr int KS = DM / 64;                  // 32-deep k-steps of one d-half (score product)
  constexpr int DT = DM / 64;                  // 16-wide d tiles per wavefront (context product)
  constexpr int IPT = 8 * NP * P;              // LDS-DMA instructions per tile (1 KiB each), all issued by the loader wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* const xch = reinterpret_cast<float*>(lds + NST * kStage);  // [2][4 waves][64 lanes][4]
#ifdef WT_ABS_STAMPS
  long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long ph_[6] = {0, 0, 0, 0, 0, 0}, tl_ = 0;
  const long long rt0_ = (long long)__builtin_amdgcn_s_memrealtime();
  ABS_STAMP(0);
#endif

  const int b = blockIdx.x / a.chunks, ck = blockIdx.x % a.chunks;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qc = lane & 15, lq = lane >> 4;
  const int sub = wid & 1, khalf = wid >> 1;
  const int n_qc = a.nq * a.H;
  const bool q_ok = qc < n_qc;
  const int qpos = q_ok ? qc / a.H : 0, qh_i = q_ok ? qc % a.H : 0;
  const long row = (long)(a.p0 + qpos) * a.B + b;

  // ---- key range of this chunk, in tiles of 32
  const int key_lo = ck * a.tiles_per_chunk * 32;
  int key_hi = key_lo + a.tiles_per_chunk * 32;
  key_hi = key_hi < a.T ? key_hi : a.T;
  const int n_tiles = key_hi > key_lo ? (key_hi - key_lo + 31) / 32 : 0;

  // ---- LDS-DMA, issued by the LOADER wave (wavefront 4, round 4).  Instruction i of a tile copies 4 keys x 256 B of
  // (plane, panel, key group kg): LDS slot (key lane >> 4, chunk lane & 15) takes the global chunk (lane & 15) ^ ((key & 7)
  // << 1) of that key row.  In rounds 3 the four computing waves issued twelve instructions each at the head of every
  // tile and stood 850-1260 cycles in the issue (the CU's vector memory path takes ~16 cycles per KiB: 48 KiB per tile),
  // a quarter of a tile's 3700 cycles (tools/cross_abs_phase_probe.hip); a fifth wave does nothing else, beside them.
  // SGPR-base form: the per-lane part of an address is the key row lq inside the instruction's four keys and the swizzled
  // chunk, whose key & 7 = 4 (kg & 1) + lq takes two values — two constant 32-bit offsets; plane, panel, key group and
  // tile are a scalar base.  A tile that reaches past the clip's last key row takes the clamping per-lane form.
  const int eb = b / a.split;  // (block-uniform)
  const _Float16* const ebase = (eb == 0 ? a.e : eb == 1 ? a.eg[1] : eb == 2 ? a.eg[2] : a.eg[3]) + (long)(b - eb * a.split) * a.T * DM;
  const unsigned long long ebytes = reinterpret_cast<unsigned long long>(ebase);
  const unsigned lds_ring = lds_addr(lds);
  const unsigned voff_even = (unsigned)(lq * DM * 2 + (((lane & 15) ^ ((lq & 7) << 1)) << 4));
  const unsigned voff_odd = (unsigned)(lq * DM * 2 + (((lane & 15) ^ (((4 + lq) & 7) << 1)) << 4));
  auto dma_part = [&](int t, int i0, int i1) {  // instructions [i0, i1) of tile t
    if (t >= n_tiles) return;
    const int stage = t % NST;
    if (key_lo + t * 32 + 32 <= a.T) {
      const unsigned long long tb = ebytes + (unsigned long long)(key_lo + t * 32) * (DM * 2);
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        const int plane = i / (8 * P), rem = i % (8 * P), panel = rem / 8, kg = rem % 8;
        const unsigned long long sb = tb + (unsigned long long)plane * (unsigned long long)a.e_plane * 2 + (unsigned)((4 * kg * DM + panel * 128) * 2);
        lds_dma16_sgpr((kg & 1) ? voff_odd : voff_even, sb, lds_ring + (unsigned)(stage * kStage + plane * kPlane + panel * 8192 + kg * 1024));
      }
    } else {
      unsigned char* const sbase = lds + stage * kStage;
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        const int plane = i / (8 * P), rem = i % (8 * P), panel = rem / 8, kg = rem % 8;
        const int key = 4 * kg + lq;
        int gk = key_lo + t * 32 + key;
        gk = gk < a.T ? gk : a.T - 1;  // keys past T re-read the last row; their scores are masked
        const int chunk = (lane & 15) ^ ((key & 7) << 1);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ebase + plane * a.e_plane + (long)gk * DM + panel * 128 + chunk * 8),
            (__attribute__((address_space(3))) void*)(sbase + plane * kPlane + panel * 8192 + kg * 1024), 16, 0, 0);
      }
    }
  };

  if (wid == 4) {
    // ---- the loader: the ring's first NST - 1 tiles, then per tile "tile t has landed" (its own counted wait: the younger
    // tile stays in flight; at most 63 of the IPT = 48 + 48 outstanding instructions are counted — the issue simply blocks
    // on the 64th), the two block-wide barriers of a tile, and the next free stage's refill split around the second one
    // so that it arrives there with the computing waves.
    constexpr int kHead = IPT / 3;
    static_assert(NST <= 2 || IPT <= 63, "the counted wait leaves one whole tile in flight: it must fit the 6-bit counter");
#pragma unroll
    for (int st0 = 0; st0 < NST - 1; ++st0) dma_part(st0, 0, IPT);
    for (int t = 0; t < n_tiles; ++t) {
      if (NST > 2 && t + 1 < n_tiles) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT < 63 ? IPT : 63) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      dma_part(t + NST - 1, 0, kHead);
      __builtin_amdgcn_s_barrier();
      dma_part(t + NST - 1, kHead, IPT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- Q' planes of this lane's query column: dynamic power-of-two scale from the column's largest element
  u32x4 qh[KS], ql[KS];
  float s_inv;
  {
    const float* src = a.qp + row * (long)(a.H * DM) + qh_i * DM + 8 * lq;
    float mx = 0.0f;
    float own[KS][8];
    // all 4 KS loads of the column are issued back to back, then consumed: written as a loop with the khalf test inside,
    // hipcc made the test a branch and waited for the loads two at a time — five L2 round trips, 11-13 k cycles per block
    f32x4 qv[2 * KS][2];
#pragma unroll
    for (int c = 0; c < 2 * KS; ++c) {
      qv[c][0] = *reinterpret_cast<const f32x4*>(src + 32 * c);
      qv[c][1] = *reinterpret_cast<const f32x4*>(src + 32 * c + 4);
    }
    const float keep = q_ok ? 1.0f : 0.0f;  // columns past nq * heads contribute zeros
#pragma unroll
    for (int c = 0; c < 2 * KS; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(qv[c][0][e]), fabsf(qv[c][1][e])));
    mx *= keep;
#pragma unroll
    for (int c = 0; c < KS; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // khalf is wave-uniform: two scalar-predicated selects per element, no branch
        own[c][e] = (khalf ? qv[c + KS][0][e] : qv[c][0][e]) * keep;
        own[c][4 + e] = (khalf ? qv[c + KS][1][e] : qv[c][1][e]) * keep;
      }
    mx = xor32_max(xor16_max(mx));
    const unsigned ex = (__float_as_uint(mx) >> 23) & 0xFFu;
    const float sc = ex < 32u ? 1.0f : __uint_as_float((268u - ex) << 23);  // largest element -> [2^14, 2^15)
    const float inv = ex < 32u ? 1.0f : __uint_as_float((ex - 14u) << 23);
    if constexpr (BF) {
      s_inv = 1.0f;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        qh[c] = u32x4{pack_bf16x2(own[c][0], own[c][1]), pack_bf16x2(own[c][2], own[c][3]), pack_bf16x2(own[c][4], own[c][5]),
                      pack_bf16x2(own[c][6], own[c][7])};
        ql[c] = u32x4{0, 0, 0, 0};
      }
    } else {
      s_inv = inv / a.e_scale;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        u32x4_t pl[3];
        split8_f16x2(own[c], sc, pl);
        qh[c] = pl[0];
        ql[c] = pl[1];
      }
    }
  }

  // (round 3's computing waves issued the first NST - 1 tiles in front of these loads — which then came back behind
  // 96 KB of HBM traffic, vmcnt retiring in issue order — and spent 11-17 k cycles, 5-7 us of a block's 18-25 us, before
  // their first tile: tools/cross_abs_phase_probe.hip; now the loader wave streams while the queries are prepared)
  ABS_STAMP(1);  // queries prepared
  f32x4 cacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) cacc[t] = f32x4{0, 0, 0, 0};
  float m_run = -1e30f, l_run = 0.0f;

  // fragment address parts that do not depend on the tile
  const int skey = 16 * sub + qc;                         // score A fragment: this lane's key row inside the tile
  const int sswz = (skey & 7) << 1;
  const int tq = (lane >> 2) & 3, tp = lane & 3;          // transposed read: lane 4 q + p of its 16-lane group
  const unsigned lds0 = lds_addr(lds);

#ifdef WT_ABS_STAMPS
  tl_ = __builtin_readcyclecounter();
#endif
  for (int t = 0; t < n_tiles; ++t) {
    // tile t has landed (the loader's counted wait, then this barrier), and every wave has finished tile t - 1, whose
    // stage the loader refills now
    __builtin_amdgcn_s_barrier();  // (the loader wave waited for tile t's DMA before it)
    __builtin_amdgcn_sched_barrier(0);
#ifdef WT_ABS_STAMPS
    if (t == 0) ABS_STAMP(2);  // first tile landed
#endif
    ABS_ACC(0);  // barrier
    ABS_ACC(1);
    const unsigned char* const st = lds + (t % NST) * kStage;

    // -- partial scores of sub-tile `sub` over d-half `khalf`
    f32x4 sp = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      const int cg = khalf * KS + c;  // k-step of the whole row: columns 32 cg .. 32 cg + 31
      const int off = (cg >> 2) * 8192 + skey * 256 + ((((cg & 3) * 4 + lq) ^ sswz) << 4);
      const half8 eh = *reinterpret_cast<const half8*>(st + off);
      if constexpr (BF) {
        sp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, eh), __builtin_bit_cast(bf16x8, qh[c]), sp, 0, 0, 0);
      } else {
        const half8 el = *reinterpret_cast<const half8*>(st + kPlane + off);
        sp = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, __builtin_bit_cast(half8, ql[c]), sp, 0, 0, 0);
        sp = __builtin_amdgcn_mfma_f32_16x16x32_f16(el, __builtin_bit_cast(half8, qh[c]), sp, 0, 0, 0);
        sp = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, __builtin_bit_cast(half8, qh[c]), sp, 0, 0, 0);
      }
    }
    ABS_ACC(2);  // partial scores
    float* const xb = xch + (t & 1) * 1024;
    *reinterpret_cast<f32x4*>(xb + (wid * 64 + lane) * 4) = sp;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // raw barrier: __syncthreads() would also drain the ring's DMAs
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s0 = *reinterpret_cast<const f32x4*>(xb + (0 * 64 + lane) * 4) + *reinterpret_cast<const f32x4*>(xb + (2 * 64 + lane) * 4);
    f32x4 s1 = *reinterpret_cast<const f32x4*>(xb + (1 * 64 + lane) * 4) + *reinterpret_cast<const f32x4*>(xb + (3 * 64 + lane) * 4);
    ABS_ACC(3);  // exchange + barrier
    // accumulator register r of this lane: key 4 lq + r of its sub-tile, query column qc
    const int kbase = key_lo + t * 32 + 4 * lq;
    if (kbase + 32 > key_hi) {  // the chunk's last tile: keys past its end do not exist
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kbase + r >= key_hi) s0[r] = -1e30f;
        if (kbase + 16 + r >= key_hi) s1[r] = -1e30f;
      }
    }
    // -- online softmax per query column (a column lives on lanes qc, qc + 16, qc + 32, qc + 48), deferred maximum
    float tmax = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
    tmax = xor32_max(xor16_max(tmax));  // lanes qc, qc + 16, qc + 32, qc + 48: no LDS round trip (bf16_split.h)
    tmax *= s_inv;
    if (__any(tmax > m_run + kDeferA)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d) cacc[d] *= alpha;
      m_run = m_new;
    }
    const float shift = kPShiftA - m_run;
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], s_inv, shift));
      s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], s_inv, shift));
      psum += s0[r] + s1[r];
    }
    psum = xor32_sum(xor16_sum(psum));
    l_run += psum;
    // P^T B fragment: element j of this lane = key 4 lq + j (j < 4, sub-tile 0) / 16 + 4 lq + j - 4 (sub-tile 1)
    unsigned ph[4], pl[4] = {0, 0, 0, 0};
    if constexpr (BF) {
      ph[0] = pack_bf16x2(s0[0], s0[1]);
      ph[1] = pack_bf16x2(s0[2], s0[3]);
      ph[2] = pack_bf16x2(s1[0], s1[1]);
      ph[3] = pack_bf16x2(s1[2], s1[3]);
    } else {
      split_f16x2(s0[0], s0[1], &ph[0], &pl[0]);
      split_f16x2(s0[2], s0[3], &ph[1], &pl[1]);
      split_f16x2(s1[0], s1[1], &ph[2], &pl[2]);
      split_f16x2(s1[2], s1[3], &ph[3], &pl[3]);
    }
    const half8 pH = __builtin_bit_cast(half8, u32x4{ph[0], ph[1], ph[2], ph[3]});
    const half8 pL = __builtin_bit_cast(half8, u32x4{pl[0], pl[1], pl[2], pl[3]});
    ABS_ACC(4);  // softmax + split
    // -- context: this wave's DT d tiles; E^T fragments by transposed reads, keys in the same order as P^T
    const unsigned sa = lds0 + (unsigned)((t % NST) * kStage);
    const int k0 = 4 * lq + tq, k1 = 16 + 4 * lq + tq;  // rows this lane addresses for the two 4-key blocks
    static_assert(DT % 2 == 0, "d tiles are read in pairs");
#pragma unroll
    for (int d = 0; d < DT; d += 2) {
      unsigned ad[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int dtg = wid * DT + d + u;                  // 16-column block of the row
        const unsigned col = (unsigned)((dtg >> 3) * 8192 + tp * 8);
        ad[u][0] = sa + col + (unsigned)(k0 * 256 + (((dtg & 7) ^ (k0 & 7)) << 5));
        ad[u][1] = sa + col + (unsigned)(k1 * 256 + (((dtg & 7) ^ (k1 & 7)) << 5));
      }
      TrFrag f[2];
      if constexpr (BF) {
        ds_read_tr16_x2_one_plane(ad[0][0], ad[0][1], ad[1][0], ad[1][1], f[0], f[1]);
      } else {
        ds_read_tr16_x2(ad[0][0], ad[0][1], ad[1][0], ad[1][1], (unsigned)kPlane, f[0], f[1]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const half8 eh = __builtin_bit_cast(half8, u32x4{f[u].h0[0], f[u].h0[1], f[u].h1[0], f[u].h1[1]});
        if constexpr (BF) {
          cacc[d + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, eh), __builtin_bit_cast(bf16x8, pH),
                                                                cacc[d + u], 0, 0, 0);
        } else {
          const half8 el = __builtin_bit_cast(half8, u32x4{f[u].g0[0], f[u].g0[1], f[u].g1[0], f[u].g1[1]});
          cacc[d + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, pL, cacc[d + u], 0, 0, 0);
          cacc[d + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el, pH, cacc[d + u], 0, 0, 0);
          cacc[d + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, pH, cacc[d + u], 0, 0, 0);
        }
      }
    }
    ABS_ACC(5);  // context
  }

  ABS_STAMP(3);  // tiles done
  // ---- record of this (row, head, chunk): c = sum p e (relative to the chunk's maximum), m, l
  if (q_ok) {
    const float cs = 1.0f / (a.e_scale * 4096.0f);  // planes of E carry e_scale, probabilities 2^12
    float* rec = a.ws + ((row * a.H + qh_i) * a.chunks + ck) * (long)(DM + 4);
#pragma unroll
    for (int d = 0; d < DT; ++d) *reinterpret_cast<f32x4*>(rec + 16 * (wid * DT + d) + 4 * lq) = cacc[d] * cs;
    if (wid == 0 && lq == 0) {
      rec[DM] = m_run * 0.69314718055994530942f;  // natural-log units, like cross_attention_step's records
      rec[DM + 1] = l_run * (1.0f / 4096.0f);
    }
  }
#ifdef WT_ABS_STAMPS
  ABS_STAMP(4);
  if (tid == 0 && blockIdx.x < 4096) {
    long long* o = g_abs_stamps + blockIdx.x * 16;
    o[0] = st_[1] - st_[0], o[1] = st_[2] - st_[1], o[2] = st_[3] - st_[2], o[3] = st_[4] - st_[3], o[4] = st_[4] - st_[0];
    o[5] = (long long)__builtin_amdgcn_s_memrealtime() - rt0_;
    o[6] = n_tiles;
    for (int i = 0; i < 6; ++i) o[8 + i] = ph_[i];
  }
#endif
}

// One block per (row, head): c[d] = sum_k w_k c_k[d] / sum_k w_k l_k with w_k = exp(m_k - max m) (the key chunks'
// partials), then the head's value projection o[64 h + j] = Wv_h[j] . c + bv[64 h + j] — 64 x d multiply-adds on
// weights in cross_q_layout() order ([head][d / 4][64 outputs][4]: a wavefront reads 1 KiB contiguous per step).
// out [rows][heads * 64]: the A operand of the ordinary cross out-projection.
__global__ __launch_bounds__(256) void cross_absorbed_combine(const float* __restrict__ ws, const float* __restrict__ wv_t,
                                                              const float* __restrict__ bv, float* __restrict__ out, int H,
                                                              int chunks, int DM) {
  __shared__ __attribute__((aligned(16))) float cs[512];
  __shared__ float part[4][64];
  const long rh = blockIdx.x;
  const int h = (int)(rh % H);
  const float* rec = ws + rh * chunks * (long)(DM + 4);
  const int tid = threadIdx.x;
  float mx = -3.0e38f;
  for (int k = 0; k < chunks; ++k) mx = fmaxf(mx, rec[k * (DM + 4) + DM]);
  if (4 * tid < DM) {
    f32x4 acc = {0, 0, 0, 0};
    float l = 0.0f;
    for (int k = 0; k < chunks; ++k) {
      const float* r = rec + k * (long)(DM + 4);
      const float w = __expf(r[DM] - mx);
      acc += w * *reinterpret_cast<const f32x4*>(r + 4 * tid);
      l += w * r[DM + 1];
    }
    *reinterpret_cast<f32x4*>(&cs[4 * tid]) = acc * (1.0f / l);
  }
  __syncthreads();
  const int j = tid & 63, cq = tid >> 6, c4n = DM / 16;  // float4 steps of this k-quarter
  const float* w = wv_t + ((long)h * (DM / 4) + cq * c4n) * 256 + j * 4;
  float acc = 0.0f;
  for (int c = 0; c < c4n; ++c) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)c * 256);
    const f32x4 x = *reinterpret_cast<const f32x4*>(&cs[(cq * c4n + c) * 4]);
    acc += (wv[0] * x[0] + wv[1] * x[1]) + (wv[2] * x[2] + wv[3] * x[3]);
  }
  part[cq][j] = acc;
  __syncthreads();
  if (tid < 64) out[(rh / H) * (long)(H * 64) + h * 64 + tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + bv[h * 64 + tid];
}

template <int DM, int NST, bool BF = false>
void launch_abs(const CrossAbsDev& g, hipStream_t s) {
  constexpr size_t smem = (size_t)NST * (BF ? 1 : 2) * (DM / 128) * 8192 + 8192;
  static_assert(smem <= 160 * 1024, "ring + exchange buffer fit the CU's LDS");
  static const bool raised = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_absorbed_attention<DM, NST, BF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)raised;
  hipLaunchKernelGGL((cross_absorbed_attention<DM, NST, BF>), dim3(g.B * g.chunks), dim3(320), smem, s, g);
}

}  // namespace

int cross_absorbed_max_nq(int heads) { return heads > 0 ? 16 / heads : 0; }

void launch_cross_absorbed(const CrossAbsorbedArgs& a, hipStream_t s) {
  const int dm = a.d_model;
  if (!a.qp || !a.e || !a.ws || a.batch < 1 || a.heads < 1 || a.heads * 64 != dm || a.T < 1 || a.chunks < 1 || a.chunks > 16 ||
      a.nq < 1 || a.nq * a.heads > 16 || a.p0 < 0 || !(a.e_scale > 0.0f) ||
      (!a.bf16 && a.e_plane < (long)(a.e2 ? std::min(a.split, a.batch) : a.batch) * a.T * dm)) {
    throw Error(kErrInvalidArg, "absorbed cross-attention: shape outside the kernel contract");
  }
  const int tiles = (a.T + 31) / 32;
  const int split = a.e2 ? a.split : a.batch;
  if (split < 1 || split > a.batch) throw Error(kErrInvalidArg, "absorbed cross-attention: bad batch split");
  const int groups = (a.batch + split - 1) / split;
  const unsigned short* src[4] = {a.e, a.e2, a.e3, a.e4};
  if (groups > 4) throw Error(kErrInvalidArg, "absorbed cross-attention: at most four encoder batches per chain");
  for (int i = 0; i < groups; ++i)
    if (!src[i]) throw Error(kErrInvalidArg, "absorbed cross-attention: an encoder batch of the chain has no planes");
  CrossAbsDev g{a.qp, reinterpret_cast<const _Float16*>(a.e),
                {reinterpret_cast<const _Float16*>(a.e), reinterpret_cast<const _Float16*>(src[1] ? src[1] : a.e),
                 reinterpret_cast<const _Float16*>(src[2] ? src[2] : a.e), reinterpret_cast<const _Float16*>(src[3] ? src[3] : a.e)},
                split, a.e_plane, a.bf16 ? 1.0f : a.e_scale, a.ws, a.batch, a.heads, a.T, a.chunks, a.nq, a.p0,
                (tiles + a.chunks - 1) / a.chunks};
  switch (dm) {
    case 128: a.bf16 ? launch_abs<128, 3, true>(g, s) : launch_abs<128, 3>(g, s); break;
    case 384: a.bf16 ? launch_abs<384, 3, true>(g, s) : launch_abs<384, 3>(g, s); break;
    case 512: a.bf16 ? launch_abs<512, 3, true>(g, s) : launch_abs<512, 2>(g, s); break;
    default: throw Error(kErrFormat, "absorbed cross-attention supports d_model 128, 384 or 512");
  }
}

void launch_cross_absorbed_combine(const float* ws, const float* wv_t, const float* bv, float* out, int rows, int heads,
                                   int chunks, int d_model, hipStream_t s) {
  if (!ws || !wv_t || !bv || !out || rows < 1 || heads < 1 || chunks < 1 || d_model != heads * 64 || d_model > 512) {
    throw Error(kErrInvalidArg, "absorbed cross-attention combine: bad shape");
  }
  hipLaunchKernelGGL(cross_absorbed_combine, dim3(rows * heads), dim3(256), 0, s, ws, wv_t, bv, out, heads, chunks, d_model);
}

}  // namespace wt

// Exact three-plane bf16 split of fp32 values (device code shared by k_gemm.hip and
// k_attention.hip).  x = h1 + h2 + h3 with h1 = x rounded to bf16 (half away from zero: add
// 0x8000 to the bit pattern, keep the upper 16 bits), h2 = (x - h1) truncated to its upper 16
// bits and h3 = x - h1 - h2, which has at most 8 significant bits left and is therefore exact
// in bf16; both subtractions are exact in fp32 (2^-100 < |x| < 3.3e38; smaller values push the
// low planes out of the normal range).  With |h2| <= 2^-8 |x| and |h3| < 2^-15 |x|,
//   a.b = a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1) + dropped,
//   |dropped| = |a2b3 + a3b2 + a3b3| < 2^-22 |a||b|   (2^-25 |a||b| on average, either sign),
// six bf16 x bf16 products (each exact in fp32) for the bf16 matrix cores, accumulated in fp32.
// Measured against fp64 the resulting GEMM error equals the fp32-MFMA kernel's
// (tests/test_gpu_kernels.py::test_gemm_variants_have_fp32_error, tools/gemm_accuracy.py).
#pragma once
#include <hip/hip_runtime.h>

namespace wt {

using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;

// 8 consecutive-k values -> one 16-byte fragment (8 bf16) per plane
__device__ __forceinline__ void split8_planes(const float (&x)[8], u32x4_t (&o)[3]) {
  unsigned h[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned u = __float_as_uint(x[e]) + 0x8000u;  // the pack below keeps the upper half only
    h[0][e] = u;
    const float r1 = x[e] - __uint_as_float(u & 0xFFFF0000u);
    h[1][e] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(h[1][e] & 0xFFFF0000u);
    h[2][e] = __float_as_uint(r2);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[p][j] = __builtin_amdgcn_perm(h[p][2 * j + 1], h[p][2 * j], 0x07060302u);
}

// two values (lo -> bits 0..15, hi -> bits 16..31) -> one packed dword per plane
__device__ __forceinline__ void split2_planes(float lo, float hi, unsigned (&o)[3]) {
  const unsigned ul = __float_as_uint(lo) + 0x8000u, uh = __float_as_uint(hi) + 0x8000u;
  const float l1 = lo - __uint_as_float(ul & 0xFFFF0000u), h1 = hi - __uint_as_float(uh & 0xFFFF0000u);
  const unsigned ul1 = __float_as_uint(l1), uh1 = __float_as_uint(h1);
  const float l2 = l1 - __uint_as_float(ul1 & 0xFFFF0000u), h2 = h1 - __uint_as_float(uh1 & 0xFFFF0000u);
  o[0] = __builtin_amdgcn_perm(uh, ul, 0x07060302u);
  o[1] = __builtin_amdgcn_perm(uh1, ul1, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(__float_as_uint(h2), __float_as_uint(l2), 0x07060302u);
}

// Two-plane fp16 split: x*scale = h1 + h2 + r with h1 = fp16(x*scale) and h2 = fp16(x*scale - h1), both
// round-to-nearest; 22 significand bits are kept (|r| <= 2^-23 |x|, against 2^-25 for fp32 itself),
// and a.b = a1b1 + a1b2 + a2b1 + O(2^-22 |a||b|): three fp16 MFMA products instead of six bf16 ones.
// `scale` is a power of two chosen by the caller so that the operand sits inside fp16's normal range
// (6.1e-5 .. 65504; |x*scale| above 65504 overflows to infinity — bf16_split3 has no such limit).
__device__ __forceinline__ void split8_f16x2(const float (&x)[8], float scale, u32x4_t (&o)[3]) {
  using half2_t = __attribute__((ext_vector_type(2))) _Float16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = x[2 * j] * scale, b = x[2 * j + 1] * scale;
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    o[0][j] = __builtin_bit_cast(unsigned, half2_t{ha, hb});
    o[1][j] = __builtin_bit_cast(unsigned, half2_t{la, lb});
  }
}

__device__ __forceinline__ void split2_f16x2(float lo, float hi, float scale, unsigned (&o)[3]) {
  using half2_t = __attribute__((ext_vector_type(2))) _Float16;
  const float a = lo * scale, b = hi * scale;
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
  o[0] = __builtin_bit_cast(unsigned, half2_t{ha, hb});
  o[1] = __builtin_bit_cast(unsigned, half2_t{la, lb});
}

// One value -> (hi, lo) fp16 planes.  `a` is first pinned to ONE fp32 value: where a is itself the result of a
// multiply by something that is not a power of two (o * (1 / l) in the attention epilogue), hipcc is otherwise free to
// make hi = fp16(a) with a fused multiply-convert (one rounding from the exact product) and to subtract the
// fp32-rounded product for lo; near a rounding tie of the fp16 grid the two disagree about which way hi went and the
// pair is off by one ulp of hi (measured: 1 element in ~10^4, error 2^-11 relative — tools/attn_diag.py).
__device__ __forceinline__ void split_f16(float a, _Float16* hi, _Float16* lo) {
  asm volatile("" : "+v"(a));
  const _Float16 h = (_Float16)a;
  *hi = h;
  *lo = (_Float16)(a - (float)h);
}

// Two elements at once, three instructions: hi pair = v_cvt_pk_f16_f32(a, b) (round to nearest even), lo halves =
// v_fma_mixlo/mixhi_f16(hi, -1, x): the fp16 hi read straight as an fma operand, x - hi exact in fp32 (hi keeps the
// top 11 of x's 24 significand bits), rounded once to fp16.  Both lo's come from the SAME stored hi, so the pair
// is consistent by construction; the generic form above costs 5 instructions per element (cvt, cvt back, sub, cvt,
// pack), which made the softmax's probability split the longest VALU stretch of the encoder attention.
// Returns packed dwords: element a in bits 0-15, b in bits 16-31 (the order MFMA fragments want).
__device__ __forceinline__ void split_f16x2(float a, float b, unsigned* hi, unsigned* lo) {
  unsigned h, l;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
  *hi = h;
  *lo = l;
}

// bf16 storage mode: two fp32 values -> one packed dword of bf16 (a in bits 0-15), round to nearest even in hardware
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf16_lo(unsigned packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(unsigned packed) { return __uint_as_float(packed & 0xFFFF0000u); }

// erf for the GELU epilogues of the plane GEMM (one evaluation per output element: 73.7 M per fc1 launch).  The
// library erff costs ~37 VALU instructions and a divergent branch per element there (measured in the ISA: the GELU
// epilogue was as long as the K = 384 main loop).  Same two-interval minimax scheme, evaluated branch-free: both
// polynomials, one raw v_exp_f32, one select.  Maximum absolute error 6e-8 against fp64 erf (tests/test_gpu_kernels.py
// holds the GELU epilogue to the same bound as before), i.e. what 1 + erf(x) can resolve in fp32.
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a), s = a * a;
  // |a| > 0.927734375: erf = sign(a) (1 - exp(p(t)))
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f), a);
  // |a| <= 0.927734375: erf = a + a q(a^2)
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}

// GELU(x) = x/2 (1 + erf(x / sqrt 2)) on TWO values at once: the same polynomials as erf_fast written on 2-vectors so
// that hipcc emits packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fma per issue slot).
// The GEMM epilogues that apply it run with no MFMA beside them (all wavefronts of a block reach the epilogue
// together), which is where packed fp32 pays (MI355X_MICROARCH.md: an anti-lever only NEXT to MFMAs).
using f32x2_t = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  const f32x2_t a = x * 0.70710678118654752440f;
  const f32x2_t t = __builtin_elementwise_abs(a), s = a * a;
  f32x2_t r = __builtin_elementwise_fma(f32x2_t{-1.72853470e-5f, -1.72853470e-5f}, t, f32x2_t{3.83197126e-4f, 3.83197126e-4f});
  const f32x2_t u = __builtin_elementwise_fma(f32x2_t{-3.88396438e-3f, -3.88396438e-3f}, t, f32x2_t{2.42546219e-2f, 2.42546219e-2f});
  r = __builtin_elementwise_fma(r, s, u);
  r = __builtin_elementwise_fma(r, t, f32x2_t{-1.06777877e-1f, -1.06777877e-1f});
  r = __builtin_elementwise_fma(r, t, f32x2_t{-6.34846687e-1f, -6.34846687e-1f});
  r = __builtin_elementwise_fma(r, t, f32x2_t{-1.28717512e-1f, -1.28717512e-1f});
  r = __builtin_elementwise_fma(r, t, -t);
  r = r * 1.44269504088896340736f;
  f32x2_t q = __builtin_elementwise_fma(f32x2_t{-5.96761703e-4f, -5.96761703e-4f}, s, f32x2_t{4.99119423e-3f, 4.99119423e-3f});
  q = __builtin_elementwise_fma(q, s, f32x2_t{-2.67681349e-2f, -2.67681349e-2f});
  q = __builtin_elementwise_fma(q, s, f32x2_t{1.12819925e-1f, 1.12819925e-1f});
  q = __builtin_elementwise_fma(q, s, f32x2_t{-3.76125336e-1f, -3.76125336e-1f});
  q = __builtin_elementwise_fma(q, s, f32x2_t{1.28379166e-1f, 1.28379166e-1f});
  const f32x2_t small = __builtin_elementwise_fma(q, a, a);
  f32x2_t erf;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r[e]), a[e]);
    erf[e] = t[e] > 0.927734375f ? big : small[e];
  }
  const f32x2_t hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, erf, hx);
}

// bf16 compute mode (BASELINE configs[3]): 8 values rounded to nearest-even bf16, one fragment
__device__ __forceinline__ u32x4_t round8_bf16(const float (&x)[8]) {
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned a = __float_as_uint(x[2 * j]), b = __float_as_uint(x[2 * j + 1]);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    o[j] = __builtin_amdgcn_perm(b, a, 0x07060302u);
  }
  return o;
}

__device__ __forceinline__ unsigned round2_bf16(float lo, float hi) {
  unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
  a += 0x7FFFu + ((a >> 16) & 1u);
  b += 0x7FFFu + ((b >> 16) & 1u);
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// global -> LDS copy of 16 bytes per lane (LDS-DMA) in the instruction's SGPR-base form: address = sbase (uniform, scalar
// registers) + voff (32-bit per-lane byte offset), LDS destination = lds_dst + lane * 16 (M0).  Written out because the
// builtin __builtin_amdgcn_global_load_lds only selects the per-lane 64-bit address form, which puts a vector instruction
// in front of every issue — and on this part a wavefront's vector instruction waits for the SIMD's other wavefronts'
// MFMA bursts (tools/mfma_valu_overlap.hip), so prefetches left late.  M0 is not used by
// anything else in the kernels that call this (gfx9+ LDS instructions do not need it).
// Reductions over the lanes l, l ^ 16 (, l ^ 32, l ^ 48) without LDS: gfx950's v_permlane32_swap / v_permlane16_swap
// exchange 32- / 16-lane rows between two registers in the vector ALU.  __shfl_xor compiles to ds_bpermute_b32, an LDS
// round trip (~100+ cycles) in the middle of a softmax's dependency chain.  swap(x, x) leaves [lo, lo] and [hi, hi]
// (32: halves; 16: the even and the odd 16-lane row of each half), so op(a, b) holds the pair's result in every lane.
__device__ __forceinline__ float xor32_max(float x) {
  using u32x2_ = __attribute__((ext_vector_type(2))) unsigned;
  const u32x2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_max(float x) {
  using u32x2_ = __attribute__((ext_vector_type(2))) unsigned;
  const u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float x) {
  using u32x2_ = __attribute__((ext_vector_type(2))) unsigned;
  const u32x2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float x) {
  using u32x2_ = __attribute__((ext_vector_type(2))) unsigned;
  const u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16_sgpr(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
#pragma clang diagnostic pop

}  // namespace wt

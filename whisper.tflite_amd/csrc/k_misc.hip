// Bandwidth-bound helper kernels of the EncDec path: LayerNorm, layout changes around
// the Conv1D stem, the log-mel epilogue (reference whisper.tflite/whisper.cpp:159-213)
// and the decoder's token embedding / greedy selection (whisper.cpp:346-361, :392-399).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "bf16_split.h"
#include "kernels.h"

namespace wt {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One wavefront per row, row held in registers (d <= 64 * PER), two-pass variance.
template <int PER>
__global__ __launch_bounds__(256) void layernorm_rows(const float* __restrict__ x,
                                                      float* __restrict__ y,
                                                      const float* __restrict__ g,
                                                      const float* __restrict__ b, int M, int d,
                                                      int* __restrict__ nonfinite) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * d;
  float v[PER];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < d ? xr[c] : 0.0f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    const float t = c < d ? v[i] - mean : 0.0f;
    q += t * t;
  }
  const float var = wave_sum(q) / (float)d;
  // a row with an infinity or a NaN (an operand outside the fp16 range of the default contraction
  // kernels, or bad input) is reported, not hidden: the engine turns the flag into an error
  if (nonfinite != nullptr && lane == 0 && !(fabsf(mean) <= 3.0e38f && var <= 3.0e38f)) atomicOr(nonfinite, 1);
  const float rstd = rsqrtf(var + 1e-5f);
  float* yr = y + row * d;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    if (c < d) yr[c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
}

// LayerNorm rows written as two fp16 planes (hi, lo) of y * scale — the A operand format of the plane GEMM — and
// optionally as fp32 too (the encoder's final LayerNorm feeds both the cross-KV GEMM and the debug tap).
// 32 lanes per row (two rows per wavefront), PER float4 per lane: 16-byte loads, 8-byte plane stores.
// BF: one bf16 plane instead (bf16 storage mode; scale and the plane offset unused).
template <int PER, bool BF>
__global__ __launch_bounds__(256) void layernorm_rows_planes(const float* __restrict__ x, _Float16* __restrict__ yp,
                                                             long plane, float scale, float* __restrict__ y32,
                                                             const float* __restrict__ g, const float* __restrict__ b,
                                                             int M, int d, int* __restrict__ nonfinite) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using half4 = __attribute__((ext_vector_type(4))) _Float16;
  const int l32 = threadIdx.x & 31;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;  // whole 32-lane halves leave together; the shuffles below stay inside a half
  const float* xr = x + row * d;
  f32x4 v[PER], gg[PER], bb[PER];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = (l32 + 32 * i) * 4;
    v[i] = *reinterpret_cast<const f32x4*>(xr + c);
    gg[i] = *reinterpret_cast<const f32x4*>(g + c);  // gain / shift requested with the row, not after its statistics
    bb[i] = *reinterpret_cast<const f32x4*>(b + c);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)d;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < PER; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = v[i][e] - mean;
      q += t * t;
    }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
  const float var = q / (float)d;
  if (nonfinite != nullptr && l32 == 0 && !(fabsf(mean) <= 3.0e38f && var <= 3.0e38f)) atomicOr(nonfinite, 1);
  const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = (l32 + 32 * i) * 4;
    f32x4 y;
    half4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = (v[i][e] - mean) * rstd * gg[i][e] + bb[i][e];
      if (!BF) {
        _Float16 h, l;
        split_f16(y[e] * scale, &h, &l);
        hi[e] = h;
        lo[e] = l;
      }
    }
    if (y32 != nullptr) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(y32 + row * d + c));  // API copy of enc_out: not read on the device
    if (BF) {
      using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
      *reinterpret_cast<u32x2*>(yp + row * d + c) = u32x2{pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    } else {
      *reinterpret_cast<half4*>(yp + row * d + c) = hi;
      *reinterpret_cast<half4*>(yp + plane + row * d + c) = lo;
    }
  }
}

// fp32 rows [M][ld] -> two fp16 planes of x * scale (hi at yp, lo at yp + plane), column n scaled by
// scale[n / seg]: the hand-over from a contraction that ran on the full-range fp32-storage kernels (a load-time
// fall-back, engine.cpp) to the next one, which takes its operand as planes.  4 columns per thread.
struct SegScales {
  float s[3];
};
__global__ __launch_bounds__(256) void f32_to_planes(const float* __restrict__ x, _Float16* __restrict__ yp, long plane,
                                                     SegScales sc, int seg, int ld, long n4) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    const int col = (int)((4 * i) % ld);
    const float s = sc.s[col / seg];
    unsigned h0, l0, h1, l1;
    split_f16x2(v[0] * s, v[1] * s, &h0, &l0);
    split_f16x2(v[2] * s, v[3] * s, &h1, &l1);
    *reinterpret_cast<u32x2*>(yp + 4 * i) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(yp + plane + 4 * i) = u32x2{l0, l1};
  }
}

// PCM -> planes for the STFT-as-GEMM (Engine::logmel): the clip's samples, clamped to the bound the scale was chosen for
__global__ __launch_bounds__(256) void pcm_to_planes(const float* __restrict__ x, _Float16* __restrict__ yp, long plane, float scale,
                                                     float limit, long n4_per_clip, long out_stride, long total4) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long b = i / n4_per_clip, j = i - b * n4_per_clip;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // a NaN sample stays NaN (it poisons its frames, as in the reference); fmin / fmax would drop it
      const float c = fminf(fmaxf(v[e], -limit), limit);
      v[e] = (v[e] != v[e] ? v[e] : c) * scale;
    }
    unsigned h0, l0, h1, l1;
    split_f16x2(v[0], v[1], &h0, &l0);
    split_f16x2(v[2], v[3], &h1, &l1);
    const long o = b * out_stride + 4 * j;
    *reinterpret_cast<u32x2*>(yp + o) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(yp + plane + o) = u32x2{l0, l1};
  }
}

// mel [B][C][T] -> fp16 planes of melT * scale, [B][T + 2][ld] (rows 1..T, columns < C).  32x32 LDS tile transpose.
template <bool BF>
__global__ __launch_bounds__(256) void mel_transpose_planes(const float* __restrict__ mel, _Float16* __restrict__ out,
                                                            long plane, float scale, int C, int T, int ld) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = mel + (long)b * C * T;
  _Float16* dst = out + (long)b * (T + 2) * ld;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (c < C && t < T) ? src[(long)c * T + t] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    if (c < C && t < T) {
      if (BF) {
        const unsigned pk = pack_bf16x2(tile[tx][ty + 8 * i], 0.0f);
        reinterpret_cast<unsigned short*>(dst)[(long)(t + 1) * ld + c] = (unsigned short)(pk & 0xFFFFu);
      } else {
        _Float16 h, l;
        split_f16(tile[tx][ty + 8 * i] * scale, &h, &l);
        dst[(long)(t + 1) * ld + c] = h;
        dst[plane + (long)(t + 1) * ld + c] = l;
      }
    }
  }
}

// mel [B][C][T] -> melT [B][T + 2][C] (rows 1..T).  32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void mel_transpose(const float* __restrict__ mel,
                                                     float* __restrict__ melT, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = mel + (long)b * C * T;
  float* dst = melT + (long)b * (T + 2) * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (c < C && t < T) ? src[(long)c * T + t] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    if (c < C && t < T) dst[(long)(t + 1) * C + c] = tile[tx][ty + 8 * i];
  }
}

__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// melacc [B*T][ld] -> logmel [B][n_mel][T] (transposed through LDS), per-clip max.
__global__ __launch_bounds__(256) void log_clipmax(const float* __restrict__ melacc, int ld,
                                                   float* __restrict__ logmel,
                                                   unsigned* __restrict__ clip_max, int n_mel,
                                                   int T, int t_valid) {
  __shared__ float tile[32][33];
  __shared__ unsigned smax;
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (threadIdx.x == 0) smax = 0u;
  __syncthreads();
  unsigned lmax = 0u;
  // the four loads of a thread are issued together from clamped (always valid) addresses and masked afterwards: under
  // their bounds tests hipcc waited for each in turn — four dependent memory round trips, 98 us per batch for 80 MB
  float e4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    e4[i] = melacc[((long)b * T + (t < T ? t : T - 1)) * ld + (c < ld ? c : ld - 1)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    float v = 0.0f;
    if (t < T && c < n_mel) {
      float e = e4[i];
      e = e < 1e-10f ? 1e-10f : e;         // whisper.cpp:176-180 (a float epsilon)
      // :182 takes log10 in double and stores a float; v_log_f32 (1 ulp of log2) times log10(2) as a two-term product is
      // within 2e-6 of it in absolute terms on [-10, 6] — the bar on the normalised log-mel is 1e-4
      const float l2 = __builtin_amdgcn_logf(e);
      v = fmaf(l2, 0.30102999566f, l2 * -1.4320989e-8f);  // log10(2) = float(0.30102999566) - 1.4320989e-8
      if (e <= 1e-10f) v = -10.0f;  // the floor itself is exact: silence normalises to exactly -1.5, as in the reference
      const unsigned o = ordered_bits(v);
      if (t < t_valid) lmax = o > lmax ? o : lmax;  // the reference's maximum runs over n_len frames only
    }
    tile[ty + 8 * i][tx] = v;
  }
  // one LDS atomic per wavefront
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = __shfl_xor(lmax, off, 64);
    lmax = o > lmax ? o : lmax;
  }
  if ((threadIdx.x & 63) != 0) lmax = 0u;
  atomicMax(&smax, lmax);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    if (c < n_mel && t < T) logmel[((long)b * n_mel + c) * T + t] = tile[tx][ty + 8 * i];
  }
  // a clip's maximum in kClipMaxWays words on 128-byte lines of their own (kClipMaxStride): with the 32 clips' words in ONE
  // line all 9024 blocks' atomics serialised on it at the memory side — 98 of this kernel's 100 us
  if (threadIdx.x == 0 && smax != 0u) atomicMax(&clip_max[((long)b * kClipMaxWays + (blockIdx.x % kClipMaxWays)) * kClipMaxStride], smax);
}

__global__ __launch_bounds__(256) void mel_normalize(float* __restrict__ logmel,
                                                     const unsigned* __restrict__ clip_max,
                                                     long per_clip) {
  const int b = blockIdx.y;
  unsigned mx = 0u;
  for (int w = 0; w < kClipMaxWays; ++w) {
    const unsigned o = clip_max[((long)b * kClipMaxWays + w) * kClipMaxStride];
    mx = o > mx ? o : mx;
  }
  const double floor_v = (double)from_ordered(mx) - 8.0;  // whisper.cpp:198-205, double
  float* p = logmel + (long)b * per_clip;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (long)gridDim.x * 256) {
    float v = p[i];
    if ((double)v < floor_v) v = (float)floor_v;  // :208-210
    p[i] = (float)(((double)v + 4.0) / 4.0);       // :212
  }
}

// One block per clip: reduce the per-tile (value, column) records of the logits GEMM with the
// reference's tie rule (larger value, then larger column), then apply the greedy step.
__global__ __launch_bounds__(256) void select_token(const unsigned long long* __restrict__ best,
                                                    int n_tiles, long long* ids, int ids_stride,
                                                    int pos, int* n_ids, int* finished, long long eot,
                                                    int stop_at_eot, int keep_ids) {
  __shared__ unsigned long long wmax[4];
  const int b = blockIdx.x;
  unsigned long long p = 0ull;
  for (int t = threadIdx.x; t < n_tiles; t += 256) {
    const unsigned long long v = best[(long)b * n_tiles + t];
    p = v > p ? v : p;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(p, off, 64);
    p = o > p ? o : p;
  }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int w = 1; w < 4; ++w) p = wmax[w] > p ? wmax[w] : p;
  const long long tok = (long long)(unsigned)(p & 0xffffffffull);
  // ids always receives the token so the next position has a defined input; n_ids stops
  // growing once the clip has emitted EOT (reference loop break, whisper.cpp:397-399)
  // keep_ids (test tap wt_dbg_set_forced_ids): the id rows were filled by the host and stay as they are, so every
  // position is decoded behind a GIVEN prefix (the logits of two engine modes become comparable step by step)
  if (!keep_ids) ids[(long)b * ids_stride + pos + 1] = tok;
  if (!finished[b]) {
    n_ids[b] = pos + 2;
    if (stop_at_eot && tok == eot) finished[b] = 1;
  }
}

}  // namespace

void launch_layernorm(const float* x, float* y, const float* g, const float* b, int M, int d,
                      hipStream_t s, int* nonfinite) {
  const int blocks = (M + 3) / 4;
  if (d <= 128) {
    hipLaunchKernelGGL(layernorm_rows<2>, dim3(blocks), dim3(256), 0, s, x, y, g, b, M, d, nonfinite);
  } else if (d <= 384) {
    hipLaunchKernelGGL(layernorm_rows<6>, dim3(blocks), dim3(256), 0, s, x, y, g, b, M, d, nonfinite);
  } else if (d <= 512) {
    hipLaunchKernelGGL(layernorm_rows<8>, dim3(blocks), dim3(256), 0, s, x, y, g, b, M, d, nonfinite);
  } else {
    throw Error(kErrFormat, "LayerNorm kernel supports rows of at most 512 elements");
  }
}

template <bool BF>
static void launch_ln_planes(const float* x, _Float16* y, long plane, float scale, float* y32, const float* g, const float* b,
                             int M, int d, hipStream_t s, int* nonfinite) {
  const int blocks = (M + 7) / 8;
  if (d == 128) {
    hipLaunchKernelGGL((layernorm_rows_planes<1, BF>), dim3(blocks), dim3(256), 0, s, x, y, plane, scale, y32, g, b, M, d, nonfinite);
  } else if (d == 384) {
    hipLaunchKernelGGL((layernorm_rows_planes<3, BF>), dim3(blocks), dim3(256), 0, s, x, y, plane, scale, y32, g, b, M, d, nonfinite);
  } else if (d == 512) {
    hipLaunchKernelGGL((layernorm_rows_planes<4, BF>), dim3(blocks), dim3(256), 0, s, x, y, plane, scale, y32, g, b, M, d, nonfinite);
  } else {
    throw Error(kErrFormat, "LayerNorm plane kernel supports rows of 128, 384 or 512 elements");
  }
}

void launch_layernorm_planes(const float* x, unsigned short* yp, long plane, float scale, float* y32, const float* g,
                             const float* b, int M, int d, hipStream_t s, int* nonfinite, bool bf16) {
  _Float16* y = reinterpret_cast<_Float16*>(yp);
  if (bf16) {
    launch_ln_planes<true>(x, y, plane, scale, y32, g, b, M, d, s, nonfinite);
  } else {
    launch_ln_planes<false>(x, y, plane, scale, y32, g, b, M, d, s, nonfinite);
  }
}

void launch_f32_to_planes(const float* x, unsigned short* yp, long plane, long M, int ld, const float* scales, int seg,
                          hipStream_t s) {
  if (seg <= 0) seg = ld;
  if (M < 1 || ld % 4 != 0 || seg % 4 != 0 || (ld + seg - 1) / seg > 3) throw Error(kErrInvalidArg, "f32_to_planes: bad shape");
  SegScales sc{{scales[0], (ld + seg - 1) / seg > 1 ? scales[1] : 1.0f, (ld + seg - 1) / seg > 2 ? scales[2] : 1.0f}};
  const long n4 = M * ld / 4;
  const unsigned blocks = (unsigned)std::min<long>((n4 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(f32_to_planes, dim3(blocks), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(yp), plane, sc, seg, ld, n4);
}

void launch_pcm_to_planes(const float* pcm, unsigned short* yp, long plane, float scale, float limit, int batch, long n,
                          long out_stride, hipStream_t s) {
  if (batch < 1 || n < 4 || n % 4 != 0 || out_stride % 8 != 0 || out_stride < n || !(scale > 0.0f) || !(limit > 0.0f)) {
    throw Error(kErrInvalidArg, "pcm_to_planes: bad shape");
  }
  const long total4 = (long)batch * (n / 4);
  const unsigned blocks = (unsigned)std::min<long>((total4 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(pcm_to_planes, dim3(blocks), dim3(256), 0, s, pcm, reinterpret_cast<_Float16*>(yp), plane, scale, limit, n / 4,
                     out_stride, total4);
}

void launch_mel_transpose_planes(const float* mel, unsigned short* out, long plane, float scale, int batch, int n_mels,
                                 int T, int ld, hipStream_t s, bool bf16) {
  const dim3 grid((T + 31) / 32, (n_mels + 31) / 32, batch);
  if (bf16) {
    hipLaunchKernelGGL(mel_transpose_planes<true>, grid, dim3(256), 0, s, mel, reinterpret_cast<_Float16*>(out), plane, scale,
                       n_mels, T, ld);
  } else {
    hipLaunchKernelGGL(mel_transpose_planes<false>, grid, dim3(256), 0, s, mel, reinterpret_cast<_Float16*>(out), plane, scale,
                       n_mels, T, ld);
  }
}

__global__ void chain_probe(float* p) {
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}

void launch_chain_probe(float* p, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(chain_probe, dim3(blocks), dim3(64), 0, stream, p);
}

void launch_mel_transpose(const float* mel, float* melT, int batch, int n_mels, int T,
                          hipStream_t s) {
  hipLaunchKernelGGL(mel_transpose, dim3((T + 31) / 32, (n_mels + 31) / 32, batch), dim3(256), 0, s,
                     mel, melT, n_mels, T);
}

void launch_log_clipmax(const float* melacc, int ld, float* logmel, unsigned* clip_max, int batch,
                        int n_mel, int T, hipStream_t s, int t_valid) {
  hipLaunchKernelGGL(log_clipmax, dim3((T + 31) / 32, (n_mel + 31) / 32, batch), dim3(256), 0, s,
                     melacc, ld, logmel, clip_max, n_mel, T, t_valid < 0 ? T : t_valid);
}

void launch_mel_normalize(float* logmel, const unsigned* clip_max, int batch, int n_mel, int T,
                          hipStream_t s) {
  hipLaunchKernelGGL(mel_normalize, dim3(64, batch), dim3(256), 0, s, logmel, clip_max,
                     (long)n_mel * T);
}

void launch_select_token(const unsigned long long* best, int n_tiles, long long* ids, int ids_stride,
                         int pos, int* n_ids, int* finished, long long eot, int stop_at_eot, int batch,
                         hipStream_t s, bool keep_ids) {
  hipLaunchKernelGGL(select_token, dim3(batch), dim3(256), 0, s, best, n_tiles, ids, ids_stride, pos,
                     n_ids, finished, eot, stop_at_eot, keep_ids ? 1 : 0);
}


// One wavefront that holds a CU for `ticks` periods of the 100 MHz constant clock: the probe Engine::create_streams uses
// to find out which of its streams the runtime mapped onto the same hardware queue (two such streams run it one after
// the other, independent ones at the same time).  Bounded: the loop ends after `ticks` whatever happens.
__global__ __launch_bounds__(64) void spin_ticks(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < (1 << 22); ++i) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

void launch_spin(int microseconds, hipStream_t s) {
  hipLaunchKernelGGL(spin_ticks, dim3(1), dim3(64), 0, s, (unsigned long long)(microseconds) * 100ull);
}

}  // namespace wt

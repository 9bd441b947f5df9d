// Host-side launchers for the gfx950 kernels of the EncDec hot path.
// Every launcher enqueues on the given stream and never synchronises or
// allocates, so a caller may capture a sequence of them into a hipGraph.
// A shape outside a kernel's contract throws wt::Error (never abort()).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <vector>

#include "error.h"

namespace wt {

// Kernel-exact timing for the roofline figures (bench.py): while `start` is set, the launchers of the encoder's
// contraction kernels attach the two events to the dispatch itself (hipExtLaunchKernelGGL), so their difference is the
// kernel's own begin -> end interval, the figure rocprofv3 --kernel-trace reports.  An event pair recorded around the
// launch on the stream also counts the wait for CUs that decoder chains hold and the barrier packets of the events.
struct LaunchTimer {
  hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchTimer g_launch_timer;

#define WT_LAUNCH_TIMED(kernel, grid, block, smem, stream, ...)                                                      \
  do {                                                                                                                \
    if (::wt::g_launch_timer.start) {                                                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, ::wt::g_launch_timer.start, ::wt::g_launch_timer.stop, \
                            0, __VA_ARGS__);                                                                          \
    } else {                                                                                                          \
      hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                             \
    }                                                                                                                 \
  } while (0)

// ------------------------------------------------------------------ GEMM ---
// C = epilogue(A . W^T): A [M][K] activations (row m at
// A + (m / a_rpb) * a_bs + (m % a_rpb) * lda), W [N][K] row-major (torch Linear
// layout), fp32 in, fp32 out.  The contraction runs on the matrix cores in one of three forms selected
// by `variant` (k_gemm.hip): fp32 MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain), or fp32
// operands split into three bf16 / two fp16 planes with fp32 accumulation (fp32-level error,
// bf16_split.h).  Requires N % 128 == 0 and K % 32 == 0; M is arbitrary.
enum GemmEpi : int {
  kEpiBias = 1,      // + bias[n]
  kEpiGelu = 2,      // exact erf GELU
  kEpiResidual = 4,  // + R[row m][n]  (R addressed like C; R may alias C)
  kEpiPos = 8,       // + pos[(m % pos_period)][n]   (encoder positional embedding)
  kEpiKvLayout = 16,  // scatter into the cross-attention KV cache layout (see below)
  // plane GEMM only: columns (2 k, 2 k + 1) hold the real and imaginary part of bin k; the output is |.|^2 per pair,
  // fp32 [M][N / 2] (row stride ldc): the front end's STFT-as-GEMM writes the power spectrum directly
  kEpiPower = 32
};

struct GemmArgs {
  const float* A = nullptr;
  const float* W = nullptr;
  float* C = nullptr;
  const float* bias = nullptr;
  const float* R = nullptr;
  const float* pos = nullptr;
  int M = 0, N = 0, K = 0;
  int a_rpb = 1 << 30;  // rows per batch item for A addressing
  long a_bs = 0;        // element stride between batch items of A
  int lda = 0;
  int c_rpb = 1 << 30;
  long c_bs = 0;
  int ldc = 0;
  int pos_period = 1;
  // kEpiKvLayout: column n = (slab * d_model + head * 64 + dd), row m = (b * T + t)
  //   -> C[((slab * kv_batch + b) * kv_heads + head) * T * 64 + t * 64 + dd]
  // where slab enumerates (layer, k|v); T = c_rpb.
  int kv_batch = 0, kv_heads = 0, kv_dmodel = 0;
  int variant = -1;  // contraction form (k_gemm.hip launch_gemm_t: 0, 11, 13, 16, 17, 18); -1 = auto
  // two-plane fp16 kernels (variants 17, 18): powers of two that bring the operands into fp16's normal
  // range, |A * a_scale| and |W * w_scale| <= 32768 (f16_scale_for); the epilogue divides them out
  float a_scale = 1.0f, w_scale = 64.0f;
};
void launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
// largest power of two s with bound * s <= 16384 (a factor 4 below fp16's maximum), clamped to 2^+-24
float f16_scale_for(float bound);
// resident blocks per CU the runtime reports for a tile variant (diagnostics)
int gemm_occupancy(int variant);
// variants launch_gemm accepts: 0 fp32 MFMA, 11 bf16 operands, 13/16 three bf16 planes, 17/18 two fp16 planes
bool gemm_variant_supported(int variant);

// ------------------------------------------------------- GEMM on fp16 planes ---
// The default encoder GEMM (k_gemm_planes.hip): both operands are pairs of fp16 planes (hi = fp16(x * scale), lo =
// fp16(x * scale - hi); 4 bytes per element like fp32).  A planes: hi at A, lo at A + a_plane (element offsets, same
// row addressing as GemmArgs); W: both planes in the blocked layout of split_weight_planes().  Output: fp32 C
// (epilogues kEpiBias, | kEpiResidual, | kEpiGelu | kEpiPos, | kEpiKvLayout) or, when P != nullptr, planes of the
// result for the next contraction (kEpiBias, | kEpiGelu): column n multiplied by out_scale[n / seg] before the split.
// Requires N % 128 == 0, K % 32 == 0, lda / a_bs / ldc / c_bs multiples of 8 elements.
struct PlaneGemmArgs {
  const unsigned short* A = nullptr;
  long a_plane = 0;
  const unsigned short* W = nullptr;
  float* C = nullptr;
  unsigned short* P = nullptr;
  long p_plane = 0;
  const float* bias = nullptr;
  const float* R = nullptr;
  const float* pos = nullptr;
  int M = 0, N = 0, K = 0;
  int a_rpb = 1 << 30;
  long a_bs = 0;
  int lda = 0;
  int c_rpb = 1 << 30;
  long c_bs = 0;
  int ldc = 0;
  int pos_period = 1;
  int kv_batch = 0, kv_heads = 0, kv_dmodel = 0;
  float a_scale = 1.0f, w_scale = 1.0f;  // the powers of two baked into the A and W planes
  float out_scale[3] = {1.0f, 1.0f, 1.0f};
  int seg = 0;  // columns per out_scale segment (0 = one segment)
  int n_cu = 256;  // CUs the launching stream may use (tile choice, k_gemm_planes.hip)
  // LayerNorm fusion (fp32 output C with kEpiBias | kEpiResidual or kEpiBias | kEpiGelu | kEpiPos, N = 384, contiguous
  // [M][N] rows): the block that finishes a 384-column row also writes LayerNorm(row) * ln_g + ln_b, times ln_scale,
  // as planes (hi at ln_P, lo at ln_P + ln_plane) — the A operand of the next GEMM — and optionally as fp32 (ln_y32)
  // with the non-finite flag of launch_layernorm.  Only the 384-column tiles can do it: launch_gemm_planes returns
  // whether it did (false: the caller launches the LayerNorm kernel).
  const float *ln_g = nullptr, *ln_b = nullptr;
  unsigned short* ln_P = nullptr;
  long ln_plane = 0;
  float ln_scale = 1.0f;
  float* ln_y32 = nullptr;
  int* nonfinite = nullptr;
};
bool launch_gemm_planes(const PlaneGemmArgs& a, int epi, hipStream_t s);
// schedule of the 384-column plane tiles (0 in step, 1 ping-pong groups; k_gemm_planes.hip) — measurement knob
int plane_gemm_mode();
void set_plane_gemm_mode(int m);
// bf16 storage mode (k_gemm_bf16.hip): A, W single bf16 matrices (the plane offsets and scales of the struct are
// unused), K a multiple of 64; P set = bf16 output (row-major, or the cross-KV cache layout with kEpiKvLayout),
// else fp32 output C
// returns true when the LayerNorm fusion (a.ln_g ...: as in launch_gemm_planes, ONE bf16 plane at ln_P, ln_scale unused) was done
bool launch_gemm_bf16_planes(const PlaneGemmArgs& a, int epi, hipStream_t s);
// W [N][K] fp32 -> bf16 [N][Kpad] (round to nearest even, zero filled)
std::vector<unsigned short> round_weights_bf16(const float* W, int N, int K, int Kpad);
// W [N][K] fp32 -> both fp16 planes, scaled by `scale`, as [N / 16][Kpad / 32][hi | lo][16 rows][32 k] with swizzled
// 16-byte chunks: the LDS image of the plane GEMM's staging instructions, 1 KiB contiguous each (Kpad >= K, zero filled)
std::vector<unsigned short> split_weight_planes(const float* W, int N, int K, int Kpad, float scale);

// Decoder-step GEMM: out[M][N] = epi(pro(x)[M][K] . W[N][K]^T), M = positions x B clips <= 128 rows (row =
// p * B + b), k_decoder.hip.  Wt is W as two fp16 planes in MFMA-fragment order, made by tile_weights_f16():
// [ceil(N/32)][K/16][plane hi|lo][64 lanes][8], scaled by w_scale.
enum DecPro : int {
  kProNone = 0,    // A operand = X [M][K] (ldx) from global memory
  kProLn = 1,      // A = LayerNorm(x) * ln_g + ln_b with x = xin, or (ids != nullptr) the token +
                   // positional embedding, which block 0 also stores to xout
  kProCombine = 2  // A = combine of the cross-attention key-chunk partials cross_ws
};
enum DecEpi : int {
  kDecResid = 0,     // Y = R + bias + acc   (R may alias Y: the residual stream, in place)
  kDecBias = 1,      // Y = acc + bias
  kDecBiasGelu = 2,  // Y = gelu(acc + bias)
  kDecLogits = 3     // (optional Y = acc) + per-tile argmax records best[m][tile], reference tie rule
};
struct DecGemmArgs {
  int logits_blocks = 0;  // resident blocks of the persistent logits kernel (0 = 512, two per CU; the pipeline asks for 256)
  const unsigned short* Wt = nullptr;
  float w_scale = 1.0f;  // power of two the planes were scaled by (tile_weights_f16)
  int N = 0, K = 0, B = 0;
  int M = 0;  // rows; 0 = B (one position)
  const float* X = nullptr;
  int ldx = 0;
  const float* xin = nullptr;
  float* xout = nullptr;
  const float* ln_g = nullptr;
  const float* ln_b = nullptr;
  const long long* ids = nullptr;  // kProLn embedding rows: row p * B + b = tok_emb[ids[b][pos + p]] + pos_emb[pos + p]
  int ids_stride = 0, pos = 0;
  const float* tok_emb = nullptr;
  const float* pos_emb = nullptr;  // the whole table [n_text_ctx][K]
  int n_vocab = 0;
  const float* cross_ws = nullptr;
  int heads = 0, chunks = 0;
  const float* bias = nullptr;
  const float* R = nullptr;
  float* Y = nullptr;
  int ldy = 0;
  unsigned long long* best = nullptr;
  int best_stride = 0;  // records per row; 0 = ceil(N / 32)
  // kProNone + kDecResid with ksplit = 2: twice the blocks, each over half of K (a K = 1536 GEMM on 12 column
  // tiles is bound by what one CU can stream).  Blocks of the first half write Y = R + bias + partial, blocks
  // of the second half write their raw partial to `part` [M][ldy]; the consumer adds the two (xpart below).
  int ksplit = 1;
  float* part = nullptr;
  // kProLn: rows = xin + xpart (the pending second half of the previous residual GEMM); block 0 stores the
  // completed rows to xout
  const float* xpart = nullptr;
  // bf16 storage mode: Wt is ONE bf16 plane in fragment order (tile_weights_bf16), w_scale unused
  bool bf16 = false;
};
void launch_dec_gemm(const DecGemmArgs& a, int pro, int epi, hipStream_t s);
// y = LayerNorm(x) * g + b : input rows of the logits GEMM
void launch_dec_finalize_ln(const float* xin, const float* g, const float* b, float* y, int B, int K,
                            hipStream_t s, const float* xpart = nullptr);

// ------------------------------------------------------------- LayerNorm ---
// y[m][:] = (x[m][:] - mean) * rstd * g + b, eps 1e-5, one wavefront per row.
void launch_layernorm(const float* x, float* y, const float* g, const float* b, int M, int d,
                      hipStream_t s, int* nonfinite = nullptr);
// Same rows, written as two fp16 planes of y * scale (hi at yp, lo at yp + plane) for the plane GEMM, and
// optionally (y32 != nullptr) also as fp32.
void launch_layernorm_planes(const float* x, unsigned short* yp, long plane, float scale, float* y32, const float* g,
                             const float* b, int M, int d, hipStream_t s, int* nonfinite = nullptr, bool bf16 = false);

// fp32 rows [M][ld] (contiguous) -> planes of x * scales[n / seg] (seg = 0: one segment), hi at yp, lo at yp + plane:
// the operand hand-over between a contraction on the fp32-storage fall-back kernels and one on the plane kernels
// probe kernel of Engine::create_streams: one wavefront busy for `microseconds` (k_misc.hip)
void launch_spin(int microseconds, hipStream_t s);

void launch_f32_to_planes(const float* x, unsigned short* yp, long plane, long M, int ld, const float* scales, int seg,
                          hipStream_t s);
// PCM [batch][n] fp32 -> fp16 planes of clamp(x, -limit, limit) * scale, [batch][out_stride] (columns >= n untouched:
// the caller zeroes them once — the reference's zero fill past the last sample, whisper.cpp:149-153): hi at yp, lo at
// yp + plane.  n % 4 == 0, out_stride % 8 == 0.
void launch_pcm_to_planes(const float* pcm, unsigned short* yp, long plane, float scale, float limit, int batch, long n,
                          long out_stride, hipStream_t s);

// ------------------------------------------------------ encoder attention ---
// qkv [B*T][3*d] (q | k | v, heads of 64 inside each third) -> out [B*T][d].
// Non-causal softmax(q k^T / 8) v per (clip, head), flash-style; `variant`: 0 fp32 MFMA, 1/2 three bf16
// planes (128 / 256 queries per block), 3 bf16 operands, 4 two fp16 planes.
// q_scale, k_scale, v_scale: f16_scale_for() of the operands' bounds (variant 4 only)
void launch_encoder_attention(const float* qkv, float* out, int batch, int T, int heads, int variant,
                              hipStream_t stream, float q_scale = 1.0f, float k_scale = 1.0f, float v_scale = 1.0f);
// The same attention on pre-split operands (k_attention_planes.hip): qkv as fp16 planes [B*T][3d] (hi at qkv, lo at
// qkv + plane), q already multiplied by d_head^-1/2 * log2(e) * q_scale, k by k_scale, v by v_scale (the plane GEMM's
// out_scale); output planes of the attention result * out_scale, [B*T][d], hi at out, lo at out + out_plane.
void launch_encoder_attention_planes(const unsigned short* qkv, long plane, unsigned short* out, long out_plane,
                                     int batch, int T, int heads, float q_scale, float k_scale, float v_scale,
                                     float out_scale, hipStream_t stream);
// bf16 storage mode: qkv and out are single bf16 matrices; q arrives unscaled
void launch_encoder_attention_bf16(const unsigned short* qkv, unsigned short* out, int batch, int T, int heads,
                                   hipStream_t stream);

// ------------------------------------------------------------- front end ---
// mel [B][n_mels][T] -> melT [B][T + 2][n_mels] rows 1..T (rows 0 and T+1 stay zero).
// one dependent link of a launch-latency chain (test tap): blocks x 64 threads, p[block] += 1
void launch_chain_probe(float* p, int blocks, hipStream_t stream);
void launch_mel_transpose(const float* mel, float* melT, int batch, int n_mels, int T,
                          hipStream_t s);
// mel [B][n_mels][T] -> fp16 planes of melT * scale, [B][T + 2][ld] rows 1..T, columns [0, n_mels) (the rest and
// rows 0, T + 1 stay zero): hi at out, lo at out + plane
void launch_mel_transpose_planes(const float* mel, unsigned short* out, long plane, float scale, int batch, int n_mels,
                                 int T, int ld, hipStream_t s, bool bf16 = false);
// melacc [B*T][ld] (first n_mel columns) -> logmel [B][n_mel][T] = log10(max(x,1e-10)),
// and per-clip maximum over frames [0, t_valid) (t_valid < 0: all T) into clip_max[(b * kClipMaxWays + w) * kClipMaxStride], w < kClipMaxWays (ordered-uint
// encoding, pre-zeroed).
constexpr int kClipMaxStride = 32, kClipMaxWays = 4;  // a clip's maximum: kClipMaxWays partial maxima, one 128-byte line each
void launch_log_clipmax(const float* melacc, int ld, float* logmel, unsigned* clip_max, int batch,
                        int n_mel, int T, hipStream_t s, int t_valid = -1);
// in place: x = (max(x, clipmax[b] - 8) + 4) / 4
void launch_mel_normalize(float* logmel, const unsigned* clip_max, int batch, int n_mel, int T,
                          hipStream_t s);

// --------------------------------------------------------------- decoder ---
// Appends k, v of positions pos0 .. pos0 + npos - 1 (from qkv rows p * B + b, [.][3d]) to the self-attention
// cache [2][B][cap][d] and attends each new position's q causally over positions 0 .. pos0 + p.  out rows likewise.
// bf16: the cache holds bf16 elements (bf16 storage mode).
void launch_self_attention(const float* qkv, void* kcache, void* vcache, int cap, int pos0, int npos, float* out,
                           int batch, int heads, hipStream_t s, bool bf16 = false);
// Cross attention of nq (1..4) query rows per clip over T cached keys, the query projection included:
// q = LayerNorm(x[row]) . Wq^T + bq with x the residual stream [nq * B][d], rows p * B + b.  wq_t = Wq in the
// layout of cross_q_layout(); kc, vc [B][heads][T][64]; partial results per key chunk in ws
// [row][heads][chunks][68] (o[64], m, l, pad), combined by the out-projection's prologue (kProCombine).
struct CrossAttnArgs {
  const float* x = nullptr;
  const float *ln_g = nullptr, *ln_b = nullptr, *wq_t = nullptr, *bq = nullptr;
  const void *kc = nullptr, *vc = nullptr;  // [clip][head][T][64] fp32, or bf16 when `bf16`
  float* ws = nullptr;
  int batch = 0, heads = 0, T = 0, chunks = 1, nq = 1;
  bool bf16 = false;
};
void launch_cross_attention(const CrossAttnArgs& a, hipStream_t s);
// Cross attention with the K / V projections absorbed (k_cross_absorbed.hip): scores and context are taken against
// the encoder output E itself (planes: hi at e, lo at e + e_plane, scaled by the power of two e_scale; [clip][T][d]),
// shared by all heads and layers.  qp [rows][heads * d]: absorbed queries q'_h = c0 Wk_h^T (Wq_h LN(x) + bq_h) with
// c0 = d_head^-1/2 log2(e) (row = p * batch + b; this launch handles positions p0 .. p0 + nq - 1, nq * heads <= 16).
// ws [rows][heads][chunks][d + 4]: per key chunk the unnormalised context c[d], m (natural log), l.
struct CrossAbsorbedArgs {
  const float* qp = nullptr;
  const unsigned short* e = nullptr;
  // set: the chain decodes several encoder batches of `split` clips each (the last may be shorter): clips
  // [split, 2 split) read e2, [2 split, 3 split) e3, [3 split, batch) e4 — every one indexed from 0
  const unsigned short* e2 = nullptr;
  const unsigned short* e3 = nullptr;
  const unsigned short* e4 = nullptr;
  int split = 0;
  long e_plane = 0;
  float e_scale = 1.0f;
  float* ws = nullptr;
  int batch = 0, heads = 0, d_model = 0, T = 0, chunks = 1, nq = 1, p0 = 0;
  bool bf16 = false;  // bf16 storage mode: e (and e2 .. e4) are ONE bf16 plane [clips * T][d_model]; e_plane and e_scale unused
};
void launch_cross_absorbed(const CrossAbsorbedArgs& a, hipStream_t s);
int cross_absorbed_max_nq(int heads);  // positions one launch can take
// out [rows][heads * 64] = Wv_h (chunk-combined, normalised context of head h) + bv_h: the A operand of the ordinary
// cross out-projection.  wv_t = cross_q_layout(Wv).
void launch_cross_absorbed_combine(const float* ws, const float* wv_t, const float* bv, float* out, int rows, int heads,
                                   int chunks, int d_model, hipStream_t s);
// Greedy selection after the logits GEMM: reduces the per-tile (value, column) records
// best[B][n_tiles], appends to ids and applies the EOT stop (reference whisper.cpp:397-399).
// keep_ids: the id rows are given (test tap): the token is selected, counted and checked for EOT but not written.
void launch_select_token(const unsigned long long* best, int n_tiles, long long* ids, int ids_stride,
                         int pos, int* n_ids, int* finished, long long eot, int stop_at_eot, int batch,
                         hipStream_t s, bool keep_ids = false);

// ---- load-time re-layouts of decoder weights (host) ----
// bf16 storage mode: W [N][K] fp32 -> ONE bf16 plane (round to nearest even) in the same fragment order,
// [ceil(N/32)][K/16][64 lanes][8]
std::vector<unsigned short> tile_weights_bf16(const float* W, int N, int K);
// W [N][K] fp32 -> two fp16 planes in MFMA-fragment order [ceil(N/32)][K/16][plane][64 lanes][8] (rows past N zero):
// lane (l & 31, l >> 5) of tile t, step s holds W[32t + (l & 31)][16s + 8(l >> 5) .. +7] * scale as hi = fp16(v),
// lo = fp16(v - hi).  *scale = f16_scale_for(max |W|).  K % 16 == 0.
std::vector<unsigned short> tile_weights_f16(const float* W, int N, int K, float* scale);
// Wq [d][d] -> [head][d / 4][64 outputs][4 k] for cross_attention_step's in-kernel query projection
std::vector<float> cross_q_layout(const float* Wq, int d);

}  // namespace wt

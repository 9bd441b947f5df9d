/* wt_debug.h — kernel-level taps for parity tests (host in / host out, synchronous).
 * NOT part of the drop-in boundary (wt_capi.h); they exist so that a parity failure of
 * the whole path can be localised to one gfx950 kernel.  epi bits: 1 bias, 2 GELU(erf),
 * 4 residual (R, same shape as the output), 8 positional add. */
#ifndef WT_DEBUG_H_
#define WT_DEBUG_H_
#include "wt_capi.h"
#ifdef __cplusplus
extern "C" {
#endif
/* C[M][N] = epi(A[M][K] . W[N][K]^T); needs N % 128 == 0, K % 32 == 0 */
int wt_dbg_gemm(wt_engine* h, int M, int N, int K, const float* A, const float* W, const float* bias,
                const float* R, const float* pos, int pos_period, int epi, float* C);
/* the default encoder GEMM on fp16 planes (k_gemm_planes.hip): A and W are split on the host with scales from the data;
 * planes_out != 0 returns the plane output reconstructed as (hi + lo) / scale; iters > 0 also times the launch */
int wt_dbg_gemm_planes(wt_engine* h, int M, int N, int K, const float* A, const float* W, const float* bias,
                       const float* R, const float* pos, int pos_period, int epi, int planes_out, int iters, float* C,
                       float* avg_ms, int n_cu /* CUs the tile choice assumes; 0 = all */);
/* Teacher forcing for step-by-step comparisons: ids [clips][32] (prompt first) that every following decode of exactly
 * `clips` clips (for a pair of pipelined batches: both, in chain order) feeds to the decoder instead of its own argmax
 * choices; logits, id counts and the EOT rule are computed as always, the returned ids are the given ones.  clips = 0
 * switches it off.  Ids outside the vocabulary are refused by the decode (WT_ERR_INVALID_ARG). */
int wt_dbg_set_forced_ids(wt_engine* h, const int64_t* ids, int clips);
/* schedule of the 384-column plane-GEMM tiles for subsequent launches of this process: 0 = gemm_planes_tile (both
 * wavefronts of a SIMD in step), 1 = gemm_planes_pp (ping-pong groups, 32 x 32 x 16), 2 = gemm_planes_pp16 (ping-pong groups on
 * 16 x 16 x 32 MFMAs; its persistent form where a CU runs several plane-output tiles: the default), 4 = the same without the
 * persistent form; A/B measurements in one process */
int wt_dbg_set_plane_gemm_mode(int mode);
/* the same GEMM (N = 384, fp32 output C, epi = bias | residual (5) or bias | gelu | pos (11)) with the LayerNorm of the
 * finished rows fused into its epilogue: ln_out [M][384] = LayerNorm(C row) * ln_g + ln_b reconstructed from the planes
 * the kernel wrote, ln_y32 (optional) its fp32 copy; *fused = 1 when a 384-column tile did it, 0 when the tile choice
 * (n_cu, M) took the narrow tile and nothing was written */
int wt_dbg_gemm_planes_ln(wt_engine* h, int M, int K, const float* A, const float* W, const float* bias, const float* R,
                          const float* pos, int pos_period, int epi, const float* ln_g, const float* ln_b, int n_cu,
                          float* C, float* ln_out, float* ln_y32, int* fused);
/* encoder attention on planes (k_attention_planes.hip): qkv fp32 [B*T][3*heads*64] is split on the host the way the
 * qkv GEMM's epilogue writes it; out [B*T][heads*64] reconstructed from the output planes */
int wt_dbg_encoder_attention_planes(wt_engine* h, int batch, int T, int heads, const float* qkv, int iters, float* out,
                                    float* avg_ms);
/* times `iters` back-to-back launches of the encoder GEMM on random operands (HIP events on the
 * engine's stream); variant selects the tile shape (k_gemm.hip) */
int wt_dbg_gemm_bench(wt_engine* h, int M, int N, int K, int epi, int variant, int iters, float* avg_ms);
/* (wt_dbg_dec_gemm_bench, below) times back-to-back launches of a decoder GEMM: kind 0 residual, 1 LayerNorm-fused (3: + logits and argmax records),
 * 2 combine + residual; rows = B x positions in the pass (<= 128) */
/* Interference probe: enqueues `n_enc` encoder passes over d_mel [batch][80][3000] on the encoder
 * stream and, concurrently, a chain of `chain_len` dependent trivial launches (`blocks` x 64
 * threads) on a decoder stream; returns the device time of each. */
int wt_dbg_interference(wt_engine* h, const float* d_mel, int batch, int n_enc, int chain_len, int blocks,
                        float* enc_ms, float* chain_ms);
/* Concurrency probe with the real kernels: `n_dec` (0..4) decodes over cached cross-KV slots next to
 * `n_enc` (0..4) pipelined encoder passes over d_mel (device, [batch][80][3000]); needs six
 * earlier batches so that every slot is populated. dec_ms[n_dec], enc_ms[1] = device times. */
int wt_dbg_concurrency(wt_engine* h, const float* d_mel, int batch, int n_dec, int n_enc, float* dec_ms,
                       float* enc_ms);
int wt_dbg_dec_gemm_bench(wt_engine* h, int kind, int B, int N, int K, int rows, int iters, float* avg_us);
/* decoder-step GEMM (k_decoder.hip), plain input X[B][K] (B <= 128 rows), W[N][K] (split into fp16 planes and tiled internally):
 * mode 0: Y = X.W^T + bias   1: gelu(...)   2: Y = R + bias + X.W^T (in-place residual form)
 * mode 3: Y = X.W^T and argmax_out[B] = last maximal column (reference tie rule) */
int wt_dbg_dec_gemm(wt_engine* h, int mode, int B, int N, int K, const float* X, const float* W,
                    const float* bias, const float* R, float* Y, int64_t* argmax_out);
/* LN-fused decoder GEMM: x = xin (or, when ids != NULL, x[b] = tok_emb[ids[b]] + pos_emb[pos], also
 * returned in xout); Y = act(LayerNorm(x).W^T + bias).  K in {128, 384, 512}. */
int wt_dbg_dec_ln_gemm(wt_engine* h, int B, int N, int K, const float* xin, const int64_t* ids, int pos,
                       const float* tok_emb, const float* pos_emb, int n_vocab, int n_pos,
                       const float* ln_g, const float* ln_b, const float* W, const float* bias,
                       int gelu, float* Y, float* xout);
int wt_dbg_layernorm(wt_engine* h, int M, int d, const float* x, const float* g, const float* b, float* y);
/* qkv [B*T][3*heads*64] -> out [B*T][heads*64] */
int wt_dbg_encoder_attention(wt_engine* h, int batch, int T, int heads, const float* qkv, float* out);
/* decoder cross attention with its fused query projection: x [nq*B][d] residual rows (row = p * B + b),
 * q = LayerNorm(x; ln_g, ln_b) . wq^T + bq (wq [d][d] row-major), kc/vc [B][heads][T][64] -> out [nq*B][d] */
int wt_dbg_cross_attention(wt_engine* h, int batch, int heads, int T, int chunks, int nq, const float* x,
                           const float* ln_g, const float* ln_b, const float* wq, const float* bq, const float* kc,
                           const float* vc, float* out);
/* qkv [npos*B][3d] (row = p * B + b); caches [B][cap][d] updated in place at rows pos .. pos+npos-1; out [npos*B][d] */
/* absorbed cross-attention (k_cross_absorbed.hip) + chunk combine with the heads' value projections: qp [nq * batch]
 * [heads * d] absorbed queries (log2 domain), E [batch][T][d] encoder output (split into planes on the host), wv [d][d],
 * bv [d]; out [nq * batch][d]: out[r][64 h + j] = Wv[64 h + j] . (sum_k softmax2_k(qp_h . e_k) e_k) + bv[64 h + j];
 * d = 64 * heads.  iters > 0 also times the attention launch. */
int wt_dbg_cross_absorbed(wt_engine* h, int batch, int heads, int T, int chunks, int nq, const float* qp, const float* E,
                          const float* wv, const float* bv, float* out, int iters, float* avg_us);
/* the same with E as ONE bf16 plane (bf16 storage mode): queries and probabilities rounded to bf16 in the kernel */
int wt_dbg_cross_absorbed_bf16(wt_engine* h, int batch, int heads, int T, int chunks, int nq, const float* qp, const float* E,
                               const float* wv, const float* bv, float* out, int iters, float* avg_us);
int wt_dbg_self_attention(wt_engine* h, int batch, int heads, int cap, int pos, int npos, const float* qkv,
                          float* kcache, float* vcache, float* out);
/* bf16 storage mode kernels (option "bf16"): operands are rounded to bf16 on the host, contracted by
 * gemm_bf16_planes / encoder_attention_planes<true>; bf16_out = 1 returns the kernel's bf16 output widened to fp32 */
int wt_dbg_gemm_bf16(wt_engine* h, int M, int N, int K, const float* A, const float* W, const float* bias,
                     const float* R, const float* pos, int pos_period, int epi, int bf16_out, int iters, float* C,
                     float* avg_ms);
/* x = R + bias + A . W^T on bf16 operands with the LayerNorm of the finished rows fused into the epilogue (whole-row tiles:
 * N 128 / 384 / 512): C = x (fp32), ln_out = LayerNorm(x) * ln_g + ln_b as the bf16 plane (returned as float), ln_y32 the
 * same in fp32; *fused = 0 when N has no whole-row tile (then only C is written). */
int wt_dbg_gemm_bf16_ln(wt_engine* h, int M, int N, int K, const float* A, const float* W, const float* bias, const float* R,
                        const float* ln_g, const float* ln_b, float* C, float* ln_out, float* ln_y32, int* fused);
int wt_dbg_encoder_attention_bf16(wt_engine* h, int batch, int T, int heads, const float* qkv, int iters, float* out,
                                  float* avg_ms);
/* the decoder's kernels in the bf16 storage mode (k_decoder.hip instantiations with BF = true): weights as one bf16 plane in
 * fragment order, activations rounded to bf16 in registers, fp32 accumulation; self-attention and cross-attention on
 * bf16 caches (the host passes fp32 arrays, the taps store them as bf16 and return the updated caches widened).  Same
 * arguments as the taps without the suffix. */
int wt_dbg_dec_gemm_bf16(wt_engine* h, int mode, int B, int N, int K, const float* X, const float* W, const float* bias,
                         const float* R, float* Y, int64_t* argmax_out);
int wt_dbg_dec_ln_gemm_bf16(wt_engine* h, int B, int N, int K, const float* xin, const int64_t* ids, int pos,
                            const float* tok_emb, const float* pos_emb, int n_vocab, int n_pos, const float* ln_g,
                            const float* ln_b, const float* W, const float* bias, int gelu, float* Y, float* xout);
int wt_dbg_self_attention_bf16(wt_engine* h, int batch, int heads, int cap, int pos, int npos, const float* qkv,
                               float* kcache, float* vcache, float* out);
int wt_dbg_cross_attention_bf16(wt_engine* h, int batch, int heads, int T, int chunks, int nq, const float* x,
                                const float* ln_g, const float* ln_b, const float* wq, const float* bq, const float* kc,
                                const float* vc, float* out);
#ifdef __cplusplus
}
#endif
#endif

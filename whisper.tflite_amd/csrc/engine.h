// Engine: weights resident in HBM + the launch sequence of the EncDec hot path
// (reference EncDec::transcribe, whisper.tflite/whisper.cpp:752-769), batch-first.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "error.h"
#include "host_util.h"
#include "wtw_format.h"

namespace wt {

struct Timings {
  float logmel_ms = 0, encoder_ms = 0, cross_kv_ms = 0, decoder_ms = 0, total_ms = 0;
  int batch = 0, decoder_steps = 0;
};

// Per-kernel-class device time of the last encode(), from HIP event pairs recorded on the
// engine's stream around every launch of that class (bench.py's roofline figures).
struct KernelStat {
  const char* name;
  int launches;
  double ms;     // sum of launch durations
  double flops;  // algorithmic FLOPs of those launches
  double bytes;  // algorithmic bytes (bandwidth-bound kernels)
};
// kKcGemm / kKcEncAttn: the default plane kernels (or the bf16-storage ones); *Alt: contractions that ran on the
// fp32-storage kernels instead (a load-time fall-back of that contraction, or a forced gemm_variant / attn_variant);
// kKcConvert: the fp32 -> planes hand-over between the two kinds
enum KernelClass { kKcGemm = 0, kKcEncAttn, kKcLayerNorm, kKcTranspose, kKcGemmAlt, kKcEncAttnAlt, kKcConvert, kKcCount };

struct AttnWeights {
  const float *wqkv = nullptr, *bqkv = nullptr;  // fused [3d][d] (self attention)
  const float *wo = nullptr, *bo = nullptr;
};
struct BlockWeights {  // encoder layer: fp32 row-major [N][K]
  const float *attn_ln_g, *attn_ln_b;
  AttnWeights attn;
  const float *mlp_ln_g, *mlp_ln_b, *w1, *b1, *w2, *b2;
};
// decoder Linear weights: two fp16 planes in MFMA-fragment order (kernels.h tile_weights_f16) + their scale
struct TiledW {
  const unsigned short* w = nullptr;
  float scale = 1.0f;
};
struct DecBlockWeights {
  const float *attn_ln_g, *attn_ln_b, *bqkv, *bo;
  TiledW wqkv, wo;
  const float *cross_ln_g, *cross_ln_b, *cross_wq_t, *cross_bq, *cross_bo;  // cross_wq_t: cross_q_layout()
  TiledW cross_wo;
  // absorbed cross-attention (k_cross_absorbed.hip): the key projection folded into the query side, A_h = c0 Wk_h^T Wq_h
  // stacked over heads [H d][d] with its bias c0 Wk_h^T bq_h (upload_weights); the value projection is applied to the
  // combined context of each head by cross_absorbed_combine
  TiledW wq_abs;
  const float *bq_abs = nullptr, *cross_wv_t = nullptr, *cross_bv = nullptr;  // cross_wv_t: cross_q_layout(Wv)
  const float *mlp_ln_g, *mlp_ln_b, *b1, *b2;
  TiledW w1, w2;
};

// Opens a .wtw file and checks header, tensor table and length without touching a GPU; throws wt::Error (kErrIo /
// kErrFormat).  wt_engine_create uses it to decide whether a .wtw next to a .tflite pair must be rebuilt.
void check_wtw_file(const std::string& path);

class Engine {
 public:
  // Throws std::runtime_error with a message; the C ABI maps it to a status code.
  // monolith: the reference's other engine type (whisper.h:165-179) routed to the same encoder / decoder
  // kernels; it differs in the prompt only (prompt(), engine.cpp)
  // weights_path: the .wtw to load; empty = model_prefix + ".wtw"
  Engine(const std::string& model_prefix, const std::string& vocab_path, bool multilingual,
         int device_id, bool monolith = false, const std::string& weights_path = "");
  // front end only: log-mel kernels over `filters` (80 x 201), no weights; encode()/decode() must not be called
  Engine(const FilterBank& filters, int device_id);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  const wtw::Dims& dims() const { return dims_; }
  const VocabData& vocab() const { return vocab_; }
  const FilterBank& filters() const { return filters_; }
  const Timings& timings() const { return timings_; }
  hipStream_t stream() const { return stream_; }
  hipStream_t decoder_stream(int i) const { return dstream_[i % n_dec_streams_]; }

  // options (reference hard-codes them, see wt_capi.h)
  long language = 2;  // language_id("de")
  long max_tokens = 30;
  long stop_at_eot = 1;
  long verbose = 0;
  // key chunks per (clip, head) of the decoder cross-attention: 1, 2, 4, 8, or 0 = as many as make clips x heads x
  // chunks reach 192 blocks (32 clips x 6 heads: one chunk — in the pipeline fewer, longer blocks with one query
  // prologue per (clip, head) measured 121.2 k audio-sec/s against 119.0 k with two chunks and 113.8 k with four; a
  // single clip needs the chunks to spread its 6 heads over the chip)
  long cross_chunks = 0;
  long attn_variant = 4;  // encoder attention: 4 = two fp16 planes (default), 1 = three bf16 planes (full range), 0 = fp32 MFMA
  long fc2_ksplit = 2;  // decoder fc2 (K = 4 d_model) over twice the blocks, halves summed by the consumer
  long use_graphs = 1;  // replay the decoder's launch sequence from a captured hipGraph
  // 1 = decoder cross-attention against the encoder output itself, K / V projections absorbed into the query and
  // output sides (k_cross_absorbed.hip: half the bytes per decoder position, no cross-KV GEMM in the encoder);
  // 0 = round 2's cross-KV cache.  Both storage modes run either form (absorb_active()); a flagged ln_post operand
  // (its planes are what the absorbed form streams) always takes the cache.
  long cross_absorb = 1;
  long abs_chunks = 0;  // key chunks per clip of the absorbed form: 1..16, 0 = by batch size and mode (decode_enqueue)
  // 1 = the pipeline decodes TWO consecutive submitted batches of equal size (<= 32 clips each, absorbed form) with
  // one decoder chain: the chain's ~1000 dependent launches then serve 64 clips, and the decoder streams stop being
  // what the pipeline waits for.  Results and their order are unchanged; a batch whose partner has
  // not been submitted yet is decoded alone as soon as it is collected.
  long dec_pair = 1;
  // batches per decoder chain when dec_pair is on: 2 (default), 3 or 4 — rows of a chain = group x batch <= 128 (the
  // decoder GEMMs' row contract); the pipeline needs about 3 x group + 3 batches in flight to keep three chains fed
  long dec_group = 2;
  long last_batches = 0;  // the next N submits are the last of a job: decoded one chain per batch (latency form); counts down
  // absorbed cross-attention in effect: the default mode needs E as fp16 planes (no fall-back on that operand); bf16
  // storage mode streams E as one bf16 plane
  bool absorb_active() const { return cross_absorb != 0 && (bf16 != 0 || (gemm_variant < 0 && sc_cross_kv_.f16_ok)); }
  // ... and chosen for a call: every pipelined batch (throughput: half the decoder's stream bytes, no cross-KV GEMM);
  // synchronous calls only from 32 clips on — below that the cached form's 7 launches per decoder layer beat the absorbed
  // form's 9 (single clip 7.2 against 9.0 ms, 16 clips 11.1 against 12.1 ms, 32 clips 15.2 against 15.0 ms)
  bool absorb_for(int batch) const { return absorb_active() && (pipelined_call_ || batch >= 32); }
  long gemm_variant = -1;  // -1 = plane GEMM (per-contraction fall-back to 13/16); 0 = fp32 MFMA, 13 / 16 = three bf16 planes
  // 1 = bf16 STORAGE mode (BASELINE configs[3]): bf16 weights, activations and both KV caches, fp32 accumulation,
  // fp32 residual stream; k_gemm_bf16.hip and the BF variants of the attention / decoder kernels.  Set through
  // set_bf16(): the first switch uploads the bf16 weight copies.
  long bf16 = 0;
  void set_bf16(bool on);
  // encoder kernel-class timers (wt_last_kernel_stats): 0 = off, N = event pairs around every launch of every N-th
  // encoder pass; passes without them report zero launches
  long kernel_timers = 1;
  // non-empty: replaces the reference's hard-coded prompt (test-sized vocabularies)
  std::vector<long long> prompt_override;
  // test tap (wt_dbg_set_forced_ids): id rows [clips][32] every decode of exactly that many clips follows instead of its
  // own argmax (the argmax still runs: logits, counts and the EOT rule are unchanged); empty = off
  std::vector<long long> forced_ids;
  // ids the decoder starts from (prompt_override, or the reference's rule for the engine type)
  std::vector<long long> prompt() const;

  int mel_frames() const { return 2 * dims_.n_audio_ctx; }
  size_t mel_elems() const { return size_t(dims_.n_mels) * mel_frames(); }
  size_t pcm_elems() const { return size_t(mel_frames()) * 160; }

  // d_pcm [B][pcm_elems] -> d_mel [B][n_mels][frames]; device pointers, async on stream().  valid_frames >= 0:
  // the normalisation maximum runs over frames [0, valid_frames) only (a clip shorter than the window)
  void logmel(const float* d_pcm, int batch, float* d_mel, int valid_frames = -1);
  // d_mel -> encoder output (internal) -> cross KV cache of the next pipeline slot; async on
  // the encoder stream
  void encode(const float* d_mel, int batch);
  // greedy loop over the slot encode() just filled, on the decoder stream; synchronises and
  // returns ids [B][32], n [B].
  void decode(int batch, int64_t* ids, int32_t* n_ids, float* logits_host, int logits_steps_cap);
  // Pipeline, kSlots deep: submit() enqueues encoder (stream E) + decoder (one of the three
  // decoder streams in use, in rotation) for one batch and returns.  In steady state the MFMA-bound
  // encoder of the newest batch shares the chip with the latency-bound decoder chains of the
  // previous ones: a decoder's ~1000 tiny dependent launches leave most of the chip idle on
  // their own.  collect() waits for the OLDEST submitted batch.
  void submit(const float* d_mel, int batch);
  // same from PCM: the front end runs on the pipeline's encoder stream into the staging mel buffer
  void submit_pcm(const float* d_pcm, int batch);
  void collect(int64_t* ids, int32_t* n_ids);
  int in_flight() const { return int(inflight_.size()); }
  // contractions that were given the full-range bf16 three-plane kernels at load time (bound slack, engine.cpp)
  int f16_fallbacks() const { return n_f16_fallbacks_; }
  float f16_min_slack_bits() const { return f16_min_slack_bits_; }
  int f16_contractions() const { return int(load_ok_.size()); }
  // Test hook (option "force_fallback"): bit i treats contraction i — launch order: conv1, conv2, then per layer
  // qkv, attention, out, fc1, fc2, and last the cross-KV projection — as flagged by the load-time slack check, on top
  // of the contractions that check flagged itself.  Exercises every plane <-> fall-back hand-over on any weights.
  void set_force_fallback(long mask);
  long force_fallback() const { return force_fallback_; }
  // throws unless no submitted batch is waiting for collect(): every synchronous entry point calls it first
  void require_idle() const;
  void sync();
  // makes this engine's GPU the calling thread's current device (every C-ABI entry does it)
  void bind_device();

  const float* enc_out() const { return ws_.enc_out; }
  void ensure_batch(int batch);
  float* staging_mel(int batch);  // device buffer [B][mel_elems]
  float* staging_pcm(int batch);  // device buffer [B][pcm_elems]

  // test tap: `n_dec` decodes (slots 0..n_dec-1, whose cross-KV caches must hold a previous batch) next
  // to `n_enc` pipelined encoder passes; device time of each decode and of the encoder passes together
  void debug_concurrency(const float* d_mel, int batch, int n_dec, int n_enc, float* dec_ms, float* enc_ms);
  // raw device allocator for debug entry points
  float* dalloc(size_t n_floats);

  // Resolved by decode()/sync_stats(); index = KernelClass.
  const KernelStat* kernel_stats() const { return kstats_; }

 private:
  void open_device();     // selects the GPU, checks it is gfx950
  void create_streams();  // streams, events, pinned id buffers of the pipeline slots
  void release() noexcept;  // frees every device / host resource; idempotent (destructor and failed constructor)
  void upload_weights(const std::string& path);
  void ensure_bf16_weights();
  std::string weights_path_;
  bool bf16_ready_ = false;
  struct Bf16Encoder {  // bf16 [N][Kpad] matrices of the encoder (k_gemm_bf16.hip)
    const unsigned short *conv1 = nullptr, *conv2 = nullptr, *cross_kv = nullptr;
    int conv1_kpad = 0;
    struct Layer {
      const unsigned short *qkv = nullptr, *out = nullptr, *fc1 = nullptr, *fc2 = nullptr;
    };
    std::vector<Layer> layers;
  } bf_;
  void encode_enqueue_bf16(const float* d_mel, int batch);
  const float* dev(const std::string& name) const;
  float* upload(const std::vector<float>& host);
  TiledW upload_tiled(const float* W, int N, int K);
  void build_frontend_tables();

  // group > 1: the decoder chain takes the batches (same size) of `group` consecutive slots from `slot` together
  void decode_enqueue(int batch, int slot, float* logits_host, int logits_steps_cap, int group = 1, bool pipelined = false,
                      int stream_override = -1);
  void submit_decoder(int batch, int slot);  // decoder side of submit / submit_pcm: grouped, alone, or latency form
  int group_of(int batch) const;             // batches per decoder chain for pipelined batches of this size
  int n_spare_streams_ = 0;  // probe-selected decoder streams beyond n_dec_streams_ (latency form of the last batches)
  bool pipelined_call_ = false;  // set by select_stream: the batch being enqueued belongs to submit(), not to a synchronous call
  std::vector<int> pending_;  // consecutive slots submitted, encoder enqueued, decoder waiting for the rest of their group
  void flush_pending();
  void decode_collect(int slot, int64_t* ids, int32_t* n_ids);

  int enc_cus_masked_ = 0;  // CUs the pipelined encoder stream may use
 public:
  int pipelined_encoder_cus() const { return enc_cus_masked_ > 0 ? enc_cus_masked_ : n_cu_; }
 private:
  int device_ = 0, n_cu_ = 0, reserve_ = 0;  // reserve_: CUs per XCD the pipelined encoder stream leaves free
  bool monolith_ = false, multilingual_ = true;
  hipStream_t stream_ = nullptr;   // encoder + front end: stream_full_ or stream_masked_ (select_stream)
  hipStream_t stream_full_ = nullptr, stream_masked_ = nullptr;
  hipEvent_t ev_switch_ = nullptr;
  void select_stream(bool pipelined);
  void pick_decoder_streams();  // decoder streams that do not share a hardware queue with the encoder stream
  // default encoder GEMM: the k16 split kernel in its two-plane fp16 form, at 2 blocks per CU when decoders
  // share the chip (pipelined), at 3 blocks per CU otherwise
  void encode_enqueue(const float* d_mel, int batch);
  // slots (WT_PIPELINE_DEPTH): a multiple of 1, 2, 3 and 4 times the 3 decoder streams in use, so single batches AND
  // group leaders (every second / third / fourth slot) rotate evenly over them; a group's chain holds its slots for
  // several pipeline periods, so the encoder needs that many slots ahead of it to keep running
  static constexpr int kDecStreams = 8, kSlots = 24;
  int n_dec_streams_ = 3;  // decoder streams in use: one hardware queue each (the runtime multiplexes
                           // streams onto 4 queues per priority; two decoders sharing one serialise)
  hipStream_t dstream_[kDecStreams] = {};  // decoders (batches rotate over them)
  static constexpr int kEncAsDec = kDecStreams;  // decoder-stream index meaning "the pipelined encoder stream" (submit_decoder)
  hipStream_t dec_stream_at(int i) const { return i == kEncAsDec ? (stream_masked_ ? stream_masked_ : stream_full_) : dstream_[i]; }
  hipEvent_t ev_[2] = {nullptr, nullptr};  // front end begin / end
  struct Slot {
    float* cross_kv = nullptr;  // [layer][k|v][clip][head][t][64]
    unsigned short* e_planes = nullptr;  // absorbed cross-attention: planes of the encoder output [clip][T][d]
    bool absorbed = false;               // which of the two this slot's encoder pass filled
    int pair_leader = -1;  // decoded by another slot's chain (which copies this batch's ids into THIS slot's buffers)
    hipEvent_t enc_begin = nullptr, enc_mid = nullptr, enc_done = nullptr;
    hipEvent_t dec_begin = nullptr, dec_done = nullptr;
    long long* h_ids = nullptr;  // pinned [4096][32]
    int* h_n = nullptr;          // pinned [4096]
    int *h_flag = nullptr, *d_flag = nullptr;  // non-finite encoder output (pinned copy / device word)
    int batch = 0, steps = 0, dec = 0;  // dec: decoder stream / workspace of this batch
    bool used = false;
    std::vector<hipEvent_t> kt_events;
    std::vector<hipEvent_t> dt_events;  // WT_DEC_KERNEL_TIMERS diagnostics
    std::vector<int> dt_cls;
    std::vector<int> kt_cls;
    std::vector<double> kt_flops, kt_bytes;
  } slots_[kSlots];
  struct GraphEntry {
    hipGraphExec_t exec;
    int steps;
  };
  void need_cross_kv(Slot& slot);  // the cached decoder form's cross K/V cache of a slot, on first use
  std::map<std::vector<long long>, GraphEntry> graphs_;
  hipEvent_t trace_base_ = nullptr;  // WT_TRACE_PIPELINE=1: origin of the per-batch device timeline
  int enc_slot_ = 0;        // slot the next encode() fills
  int last_enc_slot_ = 0;   // slot the last encode() filled
  std::vector<int> inflight_;
  wtw::Dims dims_{};
  VocabData vocab_;
  FilterBank filters_;
  Timings timings_;
  bool have_logmel_ = false;

  std::vector<void*> allocations_;
  std::map<std::string, const float*> tensors_;  // raw tensors by .wtw name

  // derived, re-laid-out weights
  const float *conv1_w = nullptr, *conv1_b = nullptr, *conv2_w = nullptr, *conv2_b = nullptr;
  int conv1_kpad = 0;
  // encoder weights as fp16 planes for the plane GEMM (hi plane, then lo plane at + N * K), scaled by GemmScale::w
  struct PlaneW {
    const unsigned short* w = nullptr;  // both fp16 planes, blocked (split_weight_planes)
  };
  PlaneW conv1_p_, conv2_p_, cross_kv_p_;
  struct EncLayerPlanes {
    PlaneW qkv, out, fc1, fc2;
  };
  std::vector<EncLayerPlanes> enc_planes_;
  PlaneW upload_planes(const float* W, int N, int K, int Kpad, float scale);
  int conv1_kpad_p_ = 0;  // conv1's K padded to the plane GEMM's k-tile
  const float* enc_pos = nullptr;
  std::vector<BlockWeights> enc_blocks_;
  std::vector<DecBlockWeights> dec_blocks_;
  std::vector<DecBlockWeights> dec_blocks_bf_;  // bf16 storage mode: TiledW members are single bf16 planes
  TiledW tok_emb_tiled_bf_;
  // Operand scales of the two-plane fp16 encoder kernels (powers of two, f16_scale_for): derived at load
  // time from weight-only upper bounds of every contraction operand (LayerNorm output <= |g| sqrt(d-1) + |b|,
  // Linear output <= sum |W| * input bound + |bias|, GELU(x) <= max(x, 0.17), attention output <= V bound),
  // so no operand can overflow fp16 for an input mel inside kMelBound
  struct GemmScale {
    float a = 1.0f, w = 64.0f;
    bool f16_ok = true;  // false: the operand's bound is too far above its typical magnitude (upload_weights)
  };
  // Per contraction: the plane kernels unless the load-time slack check gave that contraction the full-range form
  // (GemmScale::f16_ok) or an explicit gemm_variant / attn_variant selects the fp32-storage kernels for all of them
  bool gemm_on_planes(const GemmScale& sc) const { return gemm_variant < 0 && sc.f16_ok; }
  struct EncLayerScales {
    GemmScale qkv, out, fc1, fc2;
    float q = 1.0f, k = 1.0f, v = 1.0f;
    bool attn_f16_ok = true;
  };
  static constexpr float kMelBound = 8.0f;
  static constexpr float kF16Slack = 4096.0f;  // largest bound / typical ratio the two-plane fp16 form is used for
  int n_f16_fallbacks_ = 0;
  float f16_min_slack_bits_ = 0.0f;  // smallest log2(kF16Slack * typical / bound) over the checked operands (load time)
  long force_fallback_ = 0;
  std::vector<bool> load_ok_;  // the slack check's own verdicts, in contraction order (set_force_fallback)
  std::vector<bool*> ok_flags();  // the f16_ok / attn_f16_ok members in contraction order
  // the fp32-storage GEMM a contraction falls back to: the forced variant, else three bf16 planes (13; 16 = the same
  // at two blocks per CU beside the decoders of the pipeline)
  int alt_gemm_variant() const {
    if (gemm_variant >= 0) return int(gemm_variant);
    return (stream_ == stream_masked_ && stream_masked_) ? 16 : 13;
  }
  GemmScale sc_conv1_, sc_conv2_, sc_cross_kv_;
  std::vector<EncLayerScales> sc_layers_;
  const float *enc_ln_post_g = nullptr, *enc_ln_post_b = nullptr;
  const float *cross_kv_w = nullptr, *cross_kv_b = nullptr;  // [L*2*d][d], [L*2*d]
  const float *tok_emb = nullptr, *dec_pos = nullptr, *dec_ln_g = nullptr, *dec_ln_b = nullptr;
  TiledW tok_emb_tiled;  // logits GEMM against the tied embedding
  // front end
  PlaneW dft_basis_p_;               // [dft_n][dft_k] windowed basis, rows (2 k, 2 k + 1) = (re, im) of bin k, as fp16 planes
  const float* dft_zero_bias_ = nullptr;
  float dft_w_scale_ = 1.0f;
  int pw_ld_ = 0;                    // row stride of the power spectrum (dft_n / 2)
  long pcm_plane_ = 0;               // elements between the hi and the lo plane of the PCM workspace
  static constexpr float kPcmBound = 32.0f;  // PCM is clamped to +-32 for the fp16 planes (audio lives in +-1)
  const float* mel_w = nullptr;      // [mel_n][mel_k]
  int dft_n = 0, dft_k = 0, mel_n = 0, mel_k = 0;

  struct DecWorkspace {  // one per decoder stream
    float *xb = nullptr, *xpart = nullptr;  // fc2's K-split: first-half result / second-half partial
    float *xd = nullptr, *qkvd = nullptr, *attd = nullptr,
          *hd = nullptr, *cross_ws = nullptr, *self_kv = nullptr, *logits = nullptr;
    float *qp = nullptr, *abs_ws = nullptr, *cabs = nullptr;  // absorbed queries, per-chunk records, combined contexts
    unsigned long long* best = nullptr;
    long long* ids = nullptr;
    int *n_ids = nullptr, *finished = nullptr;
  } dws_[kDecStreams + 1];  // one per decoder stream, + one for a chain on the encoder stream (kEncAsDec)
  struct Workspace {
    int batch = 0;
    float *melT = nullptr, *h1p = nullptr, *x = nullptr, *ln = nullptr, *qkv = nullptr,
          *att = nullptr, *hid = nullptr, *enc_out = nullptr;
    // plane path (fp16 hi | lo planes; ln / qkv / att / hid reuse the fp32 buffers above, same bytes): the two
    // zero-padded convolution inputs need buffers of their own (their pad rows sit elsewhere in the plane layout)
    unsigned short *melTp = nullptr, *h1pp = nullptr;
    // planes made by launch_f32_to_planes from the fp32 output of a fall-back contraction ([B T][4 d] elements)
    unsigned short* cvt = nullptr;
    // front end
    unsigned short* pcm_planes = nullptr;
    float *pw = nullptr, *melacc = nullptr;
    unsigned* clip_max = nullptr;
    float *mel_stage = nullptr, *pcm_stage = nullptr;
    std::vector<void*> owned;
  } ws_;
  // kernel-class timer (events live in the pipeline slot being encoded)
  bool kt_on_ = true;
  long enc_count_ = 0;
  void kt_begin(int cls, double flops, double bytes);
  void kt_end();
  void resolve_kernel_stats(int slot);
  KernelStat kstats_[kKcCount] = {{"gemm_planes_tile", 0, 0, 0, 0},     {"encoder_attention_planes", 0, 0, 0, 0},
                                  {"layernorm_rows", 0, 0, 0, 0},       {"mel_transpose", 0, 0, 0, 0},
                                  {"gemm_split16_tile", 0, 0, 0, 0},    {"encoder_attention_split", 0, 0, 0, 0},
                                  {"f32_to_planes", 0, 0, 0, 0}};
  int self_cap_ = 32;
  static constexpr int kDecRowsMax = 128;  // rows of one decoder pass (k_decoder.hip: up to four 32-row tiles)
};

}  // namespace wt

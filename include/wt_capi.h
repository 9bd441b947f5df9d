/* wt_capi.h — C ABI of the MI355X (gfx950) Whisper EncDec engine.
 *
 * Drop-in boundary for the reference's engine surface (jerinphilip/whisper.tflite @ v2):
 * every entry point below is what a binding for that path would call instead of the
 * TFLite-interpreter-backed implementation.  Plain pointers and sizes only; nothing here
 * throws, exits or aborts: a file, shape or option the kernels do not support comes back as an int
 * status code plus wt_last_error().
 *
 * Threading contract = the reference's (an engine is NOT re-entrant; callers serialise,
 * cf. io/github/jerinphilip/whisper/Whisper.java:109-113): one in-flight call per handle,
 * one handle per GPU (device_id at create), one HIP stream per handle.
 */
#ifndef WT_CAPI_H_
#define WT_CAPI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wt_engine wt_engine;

enum wt_status {
  WT_OK = 0,
  WT_ERR_INVALID_ARG = 1,
  WT_ERR_IO = 2,          /* file missing / unreadable (reference: MmapFile throws, mmap_file.cpp:16-29) */
  WT_ERR_FORMAT = 3,      /* malformed weight or vocab file */
  WT_ERR_UNSUPPORTED = 4, /* e.g. an audio geometry other than the reference's fixed one */
  WT_ERR_DEVICE = 5,      /* no usable gfx950 device / HIP failure: the product has NO CPU fallback */
  WT_ERR_BUFFER = 6       /* caller buffer too small; *len still reports the needed size */
};

/* reference whisper.h:199-204 enum class EngineType */
enum wt_engine_type { WT_ENGINE_MONOLITH = 0, WT_ENGINE_ENCDEC = 1 };

/* Fixed audio geometry of the path (reference whisper.h:34-39). */
#define WT_SAMPLE_RATE 16000
#define WT_N_FFT 400
#define WT_HOP 160
#define WT_CHUNK_SAMPLES 480000 /* kSampleRate * kChunkSize */
#define WT_MAX_IDS 32           /* ids per clip: 4 prompt + <=27 generated (whisper.cpp:364-367), padded */

typedef struct wt_dims {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wt_dims;

/* Per-stage device time of the last batch call, from HIP events on the engine's stream. */
typedef struct wt_timings {
  float logmel_ms, encoder_ms, cross_kv_ms, decoder_ms, total_ms;
  int32_t batch, decoder_steps;
} wt_timings;

/* ---- lifecycle -------------------------------------------------------------------------
 * Replaces whisper::create_engine (whisper.h:259-260, whisper.cpp:778-790) and
 * EncDec::EncDec (whisper.cpp:740-750).  `model_prefix` resolves to "<prefix>.wtw" (this
 * build's weight file, standing in for "<prefix>.encoder.tflite"/"<prefix>.decoder.tflite",
 * whisper.cpp:743-744); `vocab_path` is the reference's filters+vocab .bin, byte-compatible.
 * When "<prefix>.wtw" does not exist but the reference's "<prefix>.encoder.tflite" and
 * "<prefix>.decoder.tflite" do, their weights are extracted (wt_convert_tflite below) into
 * "<prefix>.wtw" first.  WT_ENGINE_MONOLITH (reference Monolith, whisper.cpp:667-738: one graph with HF
 * generate() inside) runs the same encoder / decoder kernels from "<prefix>.wtw" with the prompt that
 * graph forces: [sot, notimestamps] for an English-only model (the head of kGoldenGeneratedIDs,
 * whisper.h:27-32), [sot, <|en|>, transcribe, notimestamps] for a multilingual one; ids are capped at the
 * engine's 31 positions where HF generate() allows 448.
 * On failure *out is NULL and wt_last_error(NULL) describes why. */
int wt_engine_create(int engine_type, const char* model_prefix, const char* vocab_path,
                     int multilingual, int device_id, wt_engine** out);
/* Replaces `delete engine` (bindings/java/whisper.tflite.cpp:36-42). NULL is a no-op. */
void wt_engine_destroy(wt_engine* h);
/* Message of the last failing call on `h` (or of the last failing create when h == NULL). */
const char* wt_last_error(const wt_engine* h);
int wt_engine_dims(const wt_engine* h, wt_dims* out);

/* Options (reference hard-codes them): "language" (prompt language id, whisper.cpp:327,
 * default language_id("de") = 2), "max_tokens" (max decoder positions, whisper.cpp:364,
 * default 30), "stop_at_eot" (whisper.cpp:397-399, default 1), "verbose" (default 0),
 * "cross_chunks" (key chunks per (clip, head) in the decoder cross attention: 1, 2, 4, 8, or 0 = by batch size, the default).
 * Kernel selection (results stay within the fp32 error budget for every value): "gemm_variant"
 * (-1 = default: every encoder GEMM on the plane kernel, operands as two fp16 planes with power-of-two scales
 * from weight-derived bounds, csrc/bf16_split.h — except the contractions the load-time slack check flagged,
 * which run the full-range form by themselves; 13, 16 = all of them on three bf16 planes split in the loop, full
 * fp32 operand range, at 3 / 2 blocks per CU; 0 = fp32 MFMA), "attn_variant" (4 = default, two fp16 planes;
 * 1 = three bf16 planes; 0 = fp32 MFMA),
 * "fc2_ksplit" (2 = default: the decoder's fc2 GEMM over twice the blocks, halves added by the
 * consumer; 1 = one block per column tile), "use_graphs" (1 = default: the decoder's launch sequence is replayed from a hipGraph).
 * Decoder form: "cross_absorb" (1 = default: cross-attention scores the decoder's queries, pre-multiplied by Wk, directly
 * against the encoder output planes and applies Wv after the softmax, so no cross K/V cache is projected or streamed;
 * 0 = the cross-KV cache of whisper.cpp's graph; both the fp32-accurate and the bf16 storage mode run either form —
 * pipelined batches and synchronous calls of 32 clips or more the absorbed one, smaller synchronous calls the cached
 * one; "cross_absorb_active" reads what is in effect), "abs_chunks" (key chunks per clip of that form, 0 = by batch size), "dec_pair" (1 = default: two consecutive
 * pipelined batches of equal size <= 32 share one decoder chain; change only with nothing in flight), "dec_group" (2 =
 * default, 3 or 4: that many consecutive batches per chain, rows = group x batch <= 128; three or four want 3 x group + 4
 * or more batches in flight and pay off on long jobs only: +1 % at four),
 * "last_batches" (N = the next N pipelined submits are the last of a job: they are decoded one chain per batch, the very
 * last on the encoder's stream, so the pipeline drains sooner; counts down to 0 by itself, may be set with batches in
 * flight), "force_fallback" (test hook: bit mask of contractions sent to the full-range kernels, nothing in flight).
 * Read-only (wt_engine_get_option): "f16_fallbacks" = contractions that were given the full-range bf16
 * three-plane kernels at load time because an operand's weight-derived bound lies more than 2^12 above its
 * typical magnitude (csrc/engine.cpp, upload_weights) — only those leave the plane kernels, the others keep them; "in_flight" = submitted, uncollected batches; "pipelined_encoder_cus" = CUs the CU-masked encoder stream of the pipeline may use.
 * Environment, read at wt_engine_create: WT_ENC_CU_RESERVE (CUs per XCD the pipelined encoder
 * stream leaves to the decoders, default 8, 0 = none), WT_DEC_STREAMS (decoder streams, default 3),
 * WT_TRACE_PIPELINE (per-batch device timeline on stderr), WT_NO_STREAM_PROBE (skip the ~10 ms probe that picks decoder
 * streams which do not share a hardware queue with the encoder stream or with each other). */
int wt_engine_set_option(wt_engine* h, const char* key, long value);
int wt_engine_get_option(const wt_engine* h, const char* key, long* value);
/* Replaces the reference's hard-coded prompt [sot, 50259+language, transcribe, notimestamps]
 * (whisper.cpp:327-339) with n (1..8) caller ids; n = 0 restores the default.  Needed for
 * test-sized vocabularies that do not contain the multilingual special ids. */
int wt_engine_set_prompt(wt_engine* h, const int64_t* ids, int n);

/* ---- single-clip entry points (the reference's two virtuals) ----------------------------
 * Replace Engine::transcribe(std::vector<float>&) (whisper.h:160, whisper.cpp:752-769; JNI
 * transcribeBuffer, bindings/java/whisper.tflite.cpp:45-58) and
 * Engine::transcribe(const char*) (whisper.h:161, whisper.cpp:771-776; JNI transcribeFile
 * :61-71).  pcm is padded with zeros / truncated to 480000 samples like the reference
 * (whisper.cpp:753) but the caller's buffer is not modified.  Text is written without a
 * terminating NUL guarantee beyond min(len, cap-1); *len gets the full byte length. */
int wt_transcribe_pcm(wt_engine* h, const float* pcm, size_t n_samples, char* out, size_t cap,
                      size_t* len);
int wt_transcribe_file(wt_engine* h, const char* wav_path, char* out, size_t cap, size_t* len);

/* Long audio (the reference truncates to one 30 s window, whisper.cpp:753,773):
 * pcm is cut into consecutive 30 s windows (the last one zero-padded), the windows are
 * transcribed as batches of up to 32 clips, and the per-window texts — each exactly what
 * wt_transcribe_pcm returns for that window — are joined with '\n'. */
int wt_transcribe_long_pcm(wt_engine* h, const float* pcm, size_t n_samples, char* out, size_t cap,
                           size_t* len);

/* ---- batch entry points (the reference is batch 1; clips are independent) ---------------
 * Host-pointer forms copy over PCIe; *_dev forms take device pointers already in HBM (the
 * form bench.py times).  All are synchronous on return.
 *   pcm  [B][480000] fp32           mel [B][n_mels][2*n_audio_ctx] fp32 (reference Mel
 *   layout, whisper.cpp:184: [mel][frame])     ids [B][WT_MAX_IDS] int64, n_ids [B] int32.
 * ids rows hold prompt + generated ids exactly as Decoder::forward returns them
 * (whisper.cpp:402), zero-padded. */
/* Device buffers for the *_dev forms, allocated on the engine's GPU by the engine's own HIP runtime: a host program
 * (bench.py, the tests, a binding) needs no HIP of its own to keep its inputs resident in HBM.  upload / download are
 * synchronous copies (offset in bytes into the buffer); wt_device_synchronize returns once the device is idle. */
int wt_device_alloc(wt_engine* h, size_t bytes, void** d_ptr);
int wt_device_free(wt_engine* h, void* d_ptr);
int wt_device_upload(wt_engine* h, void* d_dst, size_t offset, const void* src, size_t bytes);
int wt_device_download(wt_engine* h, void* dst, const void* d_src, size_t offset, size_t bytes);
int wt_device_synchronize(wt_engine* h);
int wt_logmel_batch(wt_engine* h, const float* pcm, int batch, float* mel);
int wt_logmel_batch_dev(wt_engine* h, const float* d_pcm, int batch, float* d_mel);
int wt_encdec_tokens_batch(wt_engine* h, const float* mel, int batch, int64_t* ids, int32_t* n_ids);
int wt_encdec_tokens_batch_dev(wt_engine* h, const float* d_mel, int batch, int64_t* ids,
                               int32_t* n_ids);
/* PCM -> ids in one call (front end + encoder + decoder), device-resident input. */
int wt_transcribe_tokens_batch_dev(wt_engine* h, const float* d_pcm, int batch, int64_t* ids,
                                   int32_t* n_ids);

/* Pipelined form of the same path: submit enqueues encoder (one HIP stream) and decoder (one
 * of three further streams, in rotation) for one device-resident batch and returns at once;
 * collect blocks until the OLDEST submitted batch has its ids on the host.  In steady state
 * the MFMA-bound encoder of the newest batch shares the chip with the latency/HBM-bound
 * decoder chains of the previous ones.  d_mel must stay valid until that batch is collected.
 * At most WT_PIPELINE_DEPTH uncollected submits (ten keep the default pipeline full); batch <= 64. */
#define WT_PIPELINE_DEPTH 24
int wt_pipeline_submit_dev(wt_engine* h, const float* d_mel, int batch);
/* Same from device-resident PCM [batch][480000]: the log-mel front end (whisper.cpp:109-216) runs
 * on the pipeline's encoder stream ahead of the encoder; d_pcm may be reused once the call returns
 * only after the batch is collected. */
int wt_pipeline_submit_pcm_dev(wt_engine* h, const float* d_pcm, int batch);
int wt_pipeline_collect(wt_engine* h, int64_t* ids, int32_t* n_ids);

/* Stage taps for parity tests: encoder output [B][n_audio_ctx][n_audio_state] and the
 * last-position logits of every argmax step [B][steps][n_vocab] (either may be NULL). */
int wt_encdec_debug_batch(wt_engine* h, const float* mel, int batch, int64_t* ids, int32_t* n_ids,
                          float* enc_out, float* logits, int logits_steps_cap);

int wt_last_timings(const wt_engine* h, wt_timings* out);

/* Per-kernel-class device time of the encoder phase of the last batch call: HIP event pairs
 * recorded on the engine's stream around every launch of the class.  flops / bytes are the
 * ALGORITHMIC work of those launches (2*M*N*K per GEMM with the true K, 4*B*H*T*T*64 per
 * attention, read+write bytes for bandwidth-bound kernels).  Returns the number of classes
 * (<= cap entries written). */
typedef struct wt_kernel_stat {
  char name[48];
  int32_t launches;
  int32_t reserved;
  double ms, flops, bytes;
} wt_kernel_stat;
int wt_last_kernel_stats(const wt_engine* h, wt_kernel_stat* out, int cap);

/* ---- host-side helpers of the path ------------------------------------------------------ */
/* whisper.cpp:634-665 decode() over the engine's vocab. */
int wt_decode_text(wt_engine* h, const int64_t* ids, int n, int omit_special_tokens, char* out,
                   size_t cap, size_t* len);
/* whisper.cpp:510-515 / :517 language table (returns the table size, 100, when absent). */
int wt_language_id(const char* code);
const char* wt_lang_code(int id);
/* wav_util.cpp:18-87 wav_read_legacy: *n gets the sample count (WT_ERR_IO when the reference
 * would return an empty vector); at most cap samples are written. */
int wt_wav_read_legacy(const char* path, float* out, size_t cap, size_t* n);
/* token ids the engine uses: out[0..8] = n_vocab, eot, sot, translate, transcribe, prev,
 * solm, not, beg (whisper.h:69-91 after whisper.cpp:218-226). */
int wt_vocab_info(const wt_engine* h, int32_t out[9]);
/* mel filter bank as loaded from the vocab file: [n_mel][n_fft]; returns element count. */
int wt_filters(const wt_engine* h, float* out, size_t cap, int32_t* n_mel, int32_t* n_fft);

/* ---- vocab / filter file on the host (no GPU needed) --------------------------------------
 * Replace MmapFile + Reader::read (mmap_file.cpp:13-31, whisper.cpp:519-611, :746-749), Vocab
 * (whisper.h:44-94) and decode() (whisper.cpp:634-665) for callers that only need the tables. */
typedef struct wt_vocab wt_vocab;
int wt_vocab_open(const char* vocab_path, int multilingual, wt_vocab** out);
void wt_vocab_close(wt_vocab* v);
/* out[0..8] as wt_vocab_info */
int wt_vocab_get_info(const wt_vocab* v, int32_t out[9]);
/* as wt_filters */
int wt_vocab_get_filters(const wt_vocab* v, float* out, size_t cap, int32_t* n_mel, int32_t* n_fft);
/* number of id -> token entries (file tokens + synthesised specials) */
int wt_vocab_size(const wt_vocab* v);
/* bytes of one token (WT_ERR_INVALID_ARG when the id has no entry) */
int wt_vocab_token(const wt_vocab* v, int id, char* out, size_t cap, size_t* len);
int wt_vocab_decode(const wt_vocab* v, const int64_t* ids, int n, int omit_special_tokens, char* out,
                    size_t cap, size_t* len);

/* ---- the log-mel front end as a free function ---------------------------------------------
 * Replaces whisper::log_mel_spectrogram (whisper.h:123, whisper.cpp:109-216) for callers that hold a
 * Filters table but no engine: the same gfx950 kernels the engine uses, on `device_id`, through a
 * process-wide front-end context created at the first call.  Only the reference's fixed geometry is
 * provided (16 kHz, fft 400, hop 160, 80 x 201 filters, n_samples <= 480000), anything else is
 * WT_ERR_UNSUPPORTED.  mel_out [n_mel][n_len] with n_len = n_samples / 160, the layout of Mel::data.
 * At most four contexts (device x filter table) are kept, least recently used evicted; wt_shutdown() releases
 * them (call it before unloading the library / at exit, ahead of the HIP runtime's own teardown). */
void wt_shutdown(void);
int wt_log_mel_spectrogram(const float* samples, int n_samples, const float* filters, int n_mel,
                           int n_fft_bins, int device_id, float* mel_out, size_t cap, int* n_len);

/* ---- asset tooling (stand-ins for the reference's offline export) -------- */
/* Weight extractor: reads the reference's model pair "<prefix>.encoder.tflite" + "<prefix>.decoder.tflite"
 * (TFLite FlatBuffers, parsed by hand; float32 / float16 constants and dynamic-range int8 weights with
 * per-tensor or per-axis scales are de-quantised) and writes "<out_path>" in .wtw format.  Tensors are
 * identified by shape and by their position in each graph's operator order (csrc/tflite_extract.cpp). */
int wt_convert_tflite(const char* model_prefix, const char* out_path);
/* Deterministic random-init weights of a named architecture ("tiny", "tiny.en", "base",
 * "micro") -> "<path>" in .wtw format. */
int wt_write_synthetic_weights(const char* path, const char* arch, uint64_t seed);
/* filters+vocab .bin in the reference layout with a Slaney 80x201 bank and n_tokens
 * synthetic tokens. */
int wt_write_synthetic_vocab(const char* path, int n_tokens);

#ifdef __cplusplus
}
#endif
#endif /* WT_CAPI_H_ */

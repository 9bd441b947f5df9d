/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's EncDec::transcribe hot path
 * (reference: whisper.tflite/whisper.cpp:752-769).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (whisper.tflite_amd/) never links, imports or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - front end (log-mel, WAV, vocab reader, text decode, language table,
 *     argmax): bit-exact against the reference's own functions compiled from
 *     /root/reference in the build container (oracle/build_ref.sh ->
 *     oracle/_ref/libwt_ref_frontend.so) and against golden vectors produced by
 *     them (tests/golden/frontend_*.npz).
 *   - model arithmetic (encoder/decoder): the reference delegates it to the
 *     TensorFlow Lite runtime + a .tflite graph, neither of which is present
 *     (deps/tensorflow is an empty submodule; no model file exists).  PARITY
 *     UNPINNED against the reference itself; pinned instead against an
 *     independent implementation of the same architecture (HuggingFace
 *     transformers' WhisperForConditionalGeneration, CPU fp32) through
 *     tests/golden/model_*.npz.
 */
#ifndef WT_ORACLE_H_
#define WT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- front end (frontend.cpp) ------------------------------------------- */

/* whisper.cpp:109-216 log_mel_spectrogram (+ fft :58-106, dft :37-54).
 * mel_out is [n_mel][n_samples / fft_step], row-major.  Returns 0. */
int wto_logmel(const float* samples, int n_samples, int fft_size, int fft_step, int n_mel,
               int n_threads, const float* filters /*[n_mel][1+fft_size/2]*/, float* mel_out);

/* wav_util.cpp:18-87 wav_read_legacy.  Returns the sample count (may exceed cap;
 * only min(count, cap) samples are written), or -1 when the reference returns an
 * empty vector (open failure / bad magic). */
long wto_wav_read_legacy(const char* path, float* out, long cap);

/* whisper.cpp:346-361 argmax lambda: `>=` keeps the LAST index among equal maxima. */
int64_t wto_argmax_last(const float* begin, int64_t n);

/* whisper.cpp:405-517 language table. language_id returns the table size (100) when absent. */
int wto_language_id(const char* code);
const char* wto_lang_code(int id);
int wto_language_count(void);

/* whisper.cpp:519-611 Reader + :218-226 transform_vocab_multilingual, over the file
 * layout EncDec::EncDec consumes (:746-749: skip the leading u64). */
typedef struct wto_vocab wto_vocab;
wto_vocab* wto_vocab_open(const char* path, int multilingual);
void wto_vocab_close(wto_vocab* v);
/* out[0..8] = n_vocab, eot, sot, translate, transcribe, prev, solm, not, beg */
void wto_vocab_info(const wto_vocab* v, int32_t out[9]);
void wto_vocab_filters_shape(const wto_vocab* v, int32_t* n_mel, int32_t* n_fft);
const float* wto_vocab_filters(const wto_vocab* v);
int wto_vocab_size(const wto_vocab* v); /* number of id_to_token entries */
/* returns token byte length, or -1 if id has no entry; copies min(len, cap) bytes */
int wto_vocab_token(const wto_vocab* v, int id, char* out, int cap);
/* whisper.cpp:634-665 decode(); returns byte length (copies min(len, cap)); -1 on a missing id */
long wto_decode_text(const wto_vocab* v, const int64_t* ids, int n, int omit_special, char* out,
                     long cap);
/* whisper.cpp:613-631 */
long wto_remove_extra_spaces(const char* in, char* out, long cap);

/* ---- model (model.cpp) --------------------------------------------------- */

typedef struct wto_model wto_model;
wto_model* wto_model_open(const char* wtw_path);
void wto_model_close(wto_model* m);
/* n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer,
 * n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer */
void wto_model_dims(const wto_model* m, int32_t out[10]);

/* Encoder graph (exporter export/generate_onnx.py:85-93).
 * mel [n_mels][2*n_audio_ctx] -> enc_out [n_audio_ctx][n_audio_state]. */
int wto_encode(const wto_model* m, const float* mel, float* enc_out, int n_threads);

/* Greedy loop of whisper.cpp:327-402 over the decoder graph, with the intended
 * encoder->decoder hand-off (export/generate_onnx.py:207,236-237).
 *   prompt/n_prompt : initial ids (reference: [sot, 50259+lang, transcribe, not])
 *   max_positions   : reference max_decoder_tokens = 30
 *   eot             : stop id; stop_at_eot=0 keeps going (throughput runs)
 *   use_cache       : 1 = KV-cached; 0 = reference-faithful, the whole prefix and the
 *                     cross K/V are recomputed every step (whisper.cpp:367-375)
 *   ids_out         : capacity max_positions + 1; receives prompt + generated ids
 *   logits_out      : optional [n_steps][n_vocab], the last-position logits per step
 * Returns the number of argmax steps taken. */
int wto_decode_greedy(const wto_model* m, const float* enc_out, const int64_t* prompt,
                      int n_prompt, int max_positions, int64_t eot, int stop_at_eot,
                      int use_cache, int n_threads, int64_t* ids_out, int* n_ids_out,
                      float* logits_out);

/* B clips spread over n_threads threads (one clip per thread at a time):
 * mel [B][n_mels][2*ctx] -> ids [B][max_positions+1], n_ids [B]. */
int wto_encdec_batch(const wto_model* m, const float* mel, int batch, const int64_t* prompt,
                     int n_prompt, int max_positions, int64_t eot, int stop_at_eot,
                     int use_cache, int n_threads, int64_t* ids_out, int* n_ids_out);

#ifdef __cplusplus
}
#endif
#endif /* WT_ORACLE_H_ */

// whisper.tflite/whisper.h — source-compatible C++ surface of the reference engine API
// (jerinphilip/whisper.tflite @ v2, whisper.tflite/whisper.h:1-261) for the MI355X build.
//
// Same namespace, type names, member names and function signatures as the reference header, minus the
// TensorFlow Lite includes (reference whisper.h:10-11) and the three TfLiteTensor-typed internals
// (Atom / Encoder / Decoder, :128-157, and tf_type_to_name / inspect_tflite_tensor, :126,206), whose role —
// mmap a .tflite graph, build an interpreter, Invoke() — is taken by an opaque engine handle on the C ABI
// (include/wt_capi.h).  An application written against the reference header recompiles unchanged
// against this one: tests/test_reference_apps.py builds the reference's own app/encdec.cpp and
// app/minimal.cpp with -Iinclude and links them against libwhisper-tflite.so.
//
// Every declaration cites the reference line it stands for.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

struct wt_engine;  // C ABI handle (wt_capi.h)

// reference whisper.h:13-17 (kept for source compatibility; nothing in this build uses it: the C ABI
// reports errors as status codes instead of exiting)
#define TFLITE_MINIMAL_CHECK(x)                              \
  if (!(x)) {                                                \
    fprintf(stderr, "Error at %s:%d\n", __FILE__, __LINE__); \
    exit(1);                                                 \
  }

// reference whisper.h:19-22
#define TIME_DIFF_MS(start, end)                  \
  (((((end).tv_sec - (start).tv_sec) * 1000000) + \
    ((end).tv_usec - (start).tv_usec)) /          \
   1000)

namespace whisper {

// Known-answer ids of the reference (whisper.h:27-32; English vocab, Monolith path).
static constexpr int kNumGoldenGeneratedIDs = 21;
static constexpr int kGoldenGeneratedIDs[kNumGoldenGeneratedIDs] = {
    50257, 50362, 1770, 13,   2264, 346, 353, 318,  262, 46329, 286,
    262,   3504,  6097, 11,   290,  356, 389, 9675, 284, 7062};

static constexpr int kSampleRate = 16000;  // reference whisper.h:34-39
static constexpr int kNFFT = 400;
static constexpr int kNMEL = 80;
static constexpr int kHopLength = 160;
static constexpr int kChunkSize = 30;
static constexpr int kMelLen = 3000;
static constexpr int kVocabEnSize = 51864;            // :41
static constexpr int kVocabMultilingualSize = 51865;  // :42

// reference whisper.h:44-94: token table + the ids of the special tokens (English defaults;
// transform_vocab_multilingual moves six of them up by one).
struct Vocab {
  std::map<int, std::string> id_to_token;
  int n_vocab = 51864;
  int token_eot = 50256;         // end of transcript
  int token_sot = 50257;         // start of transcript
  int token_translate = 50358;   // translate
  int token_transcribe = 50359;  // transcribe
  int token_prev = 50360;        // start of prev
  int token_solm = 50361;        // start of LM
  int token_not = 50362;         // no timestamps
  int token_beg = 50363;         // timestamp begin <|0.00|>
};

struct Filters {  // reference whisper.h:96-101: [n_mel][n_fft] triangular bank, n_fft = 201 bins
  int n_mel;
  int n_fft;
  std::vector<float> data;
};

struct Mel {  // reference whisper.h:103-107: data[mel * n_len + frame]
  int n_len;
  int n_mel;
  std::vector<float> data;
};

// reference whisper.h:109-116.  Host-side helpers kept for source compatibility; the front end itself
// (log_mel_spectrogram below) does not call them: it runs on the GPU as a GEMM against the effective
// transform of this very fft (csrc/engine.cpp, build_frontend_tables).
void print(const std::vector<float>& a);                          // "[a, b, c]\n" on stdout
void dft(const std::vector<float>& in, std::vector<float>& out);  // N real -> 2N interleaved re/im
void fft(const std::vector<float>& in, std::vector<float>& out);  // radix-2 recursion over dft

// reference whisper.h:121-125 (whisper.cpp:109-216): samples -> normalised log-mel, on the gfx950 front end
// (wt_log_mel_spectrogram; device 0).  Only the reference's fixed geometry is provided: sample_rate
// 16000, fft_size 400, fft_step 160, n_mel 80 with 80 x 201 filters, n_samples <= 480000; anything else
// returns false with a message on stderr (the reference computes on the CPU for any geometry).
// n_threads is accepted and ignored.
bool log_mel_spectrogram(const float* samples, int n_samples, int sample_rate, int fft_size, int fft_step,
                         int n_mel, int n_threads, Filters& filters, Mel& mel);

void transform_vocab_multilingual(Vocab& vocab);  // reference whisper.h:127, whisper.cpp:218-226

struct Engine {  // reference whisper.h:159-163
  virtual std::string transcribe(std::vector<float>& samples) = 0;
  virtual std::string transcribe(const char* waveFile) = 0;
  virtual ~Engine() = default;
};

// reference whisper.h:165-179, whisper.cpp:667-738: one .tflite graph with HuggingFace generate() inside.
// Here the same HIP encoder / decoder kernels run from "<prefix>.wtw" with the prompt that graph forces
// ([sot, notimestamps] for an English-only model, [sot, <|en|>, transcribe, notimestamps] for a
// multilingual one) and greedy selection up to the engine's 31 positions.
struct Monolith : public Engine {
 public:
  Monolith(const std::string& model_prefix, const std::string& vocab_path, bool multilingual);
  ~Monolith() override;
  Monolith(const Monolith&) = delete;
  Monolith& operator=(const Monolith&) = delete;
  std::string transcribe(std::vector<float>& samples) final;
  std::string transcribe(const char* waveFile) final;
  wt_engine* handle() const { return handle_; }

 private:
  wt_engine* handle_ = nullptr;
};

// reference whisper.h:181-197; whisper.cpp:740-776.  `model_prefix` resolves to "<prefix>.wtw" (or is
// extracted from "<prefix>.encoder.tflite" / "<prefix>.decoder.tflite" when only those exist).  Throws
// std::runtime_error when a file cannot be opened (as the reference's MmapFile does) or no gfx950 device
// is usable.
struct EncDec : public Engine {
 public:
  EncDec(const std::string& model_prefix, const std::string& vocab_path, bool multilingual);
  ~EncDec() override;
  EncDec(const EncDec&) = delete;
  EncDec& operator=(const EncDec&) = delete;
  // Pads/truncates the CALLER's vector to 480000 samples like the reference (whisper.cpp:753).
  std::string transcribe(std::vector<float>& samples) final;
  std::string transcribe(const char* waveFile) final;
  wt_engine* handle() const { return handle_; }  // for the batch entry points of wt_capi.h

 private:
  wt_engine* handle_ = nullptr;
};

enum class EngineType { Monolith = 0, EncDec = 1 };  // reference whisper.h:199-204

// reference whisper.h:208-212, whisper.cpp:405-517
using LangKey = std::pair<std::string, std::string>;
extern std::vector<LangKey> language_meta;  // {code, name} in tokenizer order
int language_id(const std::string& code);
const std::string& lang_code(size_t id);

// reference whisper.h:236-248, whisper.cpp:519-611.  `head` points at the u32 magic, i.e. 8 bytes into the
// vocab file (EncDec skips the leading u64 payload size, whisper.cpp:746-747).  `size` (an addition, defaulted)
// bounds the parse when the caller knows how many bytes follow `head`; the reference trusts the buffer.
struct Reader {
 public:
  explicit Reader(const char* head, bool multilingual, size_t size = static_cast<size_t>(-1))
      : head_(head), multilingual_(multilingual), size_(size) {}
  void read(Filters& filters, Vocab& vocab);

 private:
  const char* head_;
  bool multilingual_;
  size_t size_;
};

std::string remove_extra_spaces(const std::string& input);  // reference whisper.h:250

// reference whisper.h:252-257, whisper.cpp:634-665 (instantiated for int and int64_t, as the reference)
template <class Int>
std::string decode(const Vocab& vocab, const Int* begin, const Int* end, bool omit_special_tokens);
std::string decode(const Vocab& vocab, const std::vector<int64_t>& generated, bool omit_special_tokens);

// reference whisper.h:259-260 / whisper.cpp:778-790: caller owns (delete) the result; nullptr + a message
// on stderr for an unknown type.
Engine* create_engine(EngineType type, const char* model_prefix, const char* vocab_path,
                      bool multilingual);

// reference wav_util.h:23 (the reference header reaches it through whisper.cpp; declared here because
// EncDec::transcribe(const char*) is specified in terms of it)
std::vector<float> wav_read_legacy(const char* filename);

}  // namespace whisper

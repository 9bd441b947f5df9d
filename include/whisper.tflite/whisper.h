// whisper.tflite/whisper.h — source-compatible C++ surface of the reference engine API
// (jerinphilip/whisper.tflite @ v2, whisper.tflite/whisper.h:24-261) for the MI355X build.
//
// Same namespace, type names, constructor and method signatures as the reference, minus the
// TensorFlow Lite includes (reference whisper.h:10-11) and the TfLiteTensor-typed internals
// (Atom/Encoder/Decoder, :128-157), which are replaced by an opaque handle on the C ABI
// (include/wt_capi.h).  An application written against the reference header — e.g.
// app/encdec.cpp — recompiles unchanged against this one.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

struct wt_engine;  // C ABI handle (wt_capi.h)

namespace whisper {

static constexpr int kSampleRate = 16000;  // reference whisper.h:34-39
static constexpr int kNFFT = 400;
static constexpr int kNMEL = 80;
static constexpr int kHopLength = 160;
static constexpr int kChunkSize = 30;
static constexpr int kMelLen = 3000;
static constexpr int kVocabEnSize = 51864;            // :41
static constexpr int kVocabMultilingualSize = 51865;  // :42

// Known-answer ids of the reference (whisper.h:27-32; English vocab, Monolith path).
static constexpr int kNumGoldenGeneratedIDs = 21;
static constexpr int kGoldenGeneratedIDs[kNumGoldenGeneratedIDs] = {
    50257, 50362, 1770, 13,   2264, 346, 353, 318,  262, 46329, 286,
    262,   3504,  6097, 11,   290,  356, 389, 9675, 284, 7062};

struct Engine {  // reference whisper.h:159-163
  virtual std::string transcribe(std::vector<float>& samples) = 0;
  virtual std::string transcribe(const char* waveFile) = 0;
  virtual ~Engine() = default;
};

// reference whisper.h:181-197; whisper.cpp:740-776.  `model_prefix` resolves to
// "<prefix>.wtw".  Throws std::runtime_error when a file cannot be opened (as the
// reference's MmapFile does) or no gfx950 device is usable.
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

// reference whisper.h:165-179.  The single-graph HF-generate engine is outside the scope of
// this build: construction throws std::runtime_error("unsupported").
struct Monolith : public Engine {
 public:
  Monolith(const std::string& model_prefix, const std::string& vocab_path, bool multilingual);
  std::string transcribe(std::vector<float>& samples) final;
  std::string transcribe(const char* waveFile) final;
};

enum class EngineType { Monolith = 0, EncDec = 1 };  // reference whisper.h:199-204

// reference whisper.h:259-260 / whisper.cpp:778-790: caller owns (delete) the result;
// nullptr + a message on stderr for an unknown or unsupported type.
Engine* create_engine(EngineType type, const char* model_prefix, const char* vocab_path,
                      bool multilingual);

// reference whisper.h:208-212
int language_id(const std::string& code);
const std::string& lang_code(size_t id);
// reference whisper.h:250
std::string remove_extra_spaces(const std::string& input);
// reference wav_util.h:23
std::vector<float> wav_read_legacy(const char* filename);

}  // namespace whisper

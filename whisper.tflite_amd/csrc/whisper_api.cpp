// C++ engine surface (include/whisper.tflite/whisper.h) as a thin wrapper over the C ABI,
// re-creating the reference's observable behaviour: exceptions for missing files
// (mmap_file.cpp:16-29), "" on soft failure (whisper.cpp:760), nullptr + stderr for an
// unknown engine type (whisper.cpp:785-789).
#include "whisper.tflite/whisper.h"

#include <cstdio>
#include <stdexcept>

#include "host_util.h"
#include "wt_capi.h"

namespace whisper {

EncDec::EncDec(const std::string& model_prefix, const std::string& vocab_path, bool multilingual) {
  const int rc = wt_engine_create(WT_ENGINE_ENCDEC, model_prefix.c_str(), vocab_path.c_str(),
                                  multilingual ? 1 : 0, 0, &handle_);
  if (rc != WT_OK) throw std::runtime_error(wt_last_error(nullptr));
}

EncDec::~EncDec() { wt_engine_destroy(handle_); }

std::string EncDec::transcribe(std::vector<float>& samples) {
  samples.resize(size_t(kSampleRate) * kChunkSize, 0);  // whisper.cpp:753 mutates the caller's vector
  std::string text(8192, '\0');
  size_t len = 0;
  int rc = wt_transcribe_pcm(handle_, samples.data(), samples.size(), &text[0], text.size(), &len);
  if (rc == WT_ERR_BUFFER) {
    text.assign(len + 1, '\0');
    rc = wt_transcribe_pcm(handle_, samples.data(), samples.size(), &text[0], text.size(), &len);
  }
  if (rc != WT_OK) {
    std::fprintf(stderr, "transcribe failed: %s\n", wt_last_error(handle_));
    return "";
  }
  text.resize(len);
  return text;
}

std::string EncDec::transcribe(const char* waveFile) {
  std::vector<float> pcmf32 = wav_read_legacy(waveFile);
  pcmf32.resize(size_t(kSampleRate) * kChunkSize, 0);  // whisper.cpp:773
  return transcribe(pcmf32);
}

Monolith::Monolith(const std::string&, const std::string&, bool) {
  throw std::runtime_error("unsupported: EngineType::Monolith is not provided by the MI355X build");
}
std::string Monolith::transcribe(std::vector<float>&) { return ""; }
std::string Monolith::transcribe(const char*) { return ""; }

Engine* create_engine(EngineType type, const char* model_prefix, const char* vocab_path,
                      bool multilingual) {
  switch (type) {
    case EngineType::EncDec:
      return new EncDec(model_prefix, vocab_path, multilingual);
    case EngineType::Monolith:
      std::fprintf(stderr, "EngineType::Monolith is not provided by the MI355X build\n");
      return nullptr;
    default:
      std::fprintf(stderr, "Unknown engine-type\n");
      break;
  }
  return nullptr;
}

int language_id(const std::string& code) { return wt::language_id(code); }
const std::string& lang_code(size_t id) { return wt::lang_code(id); }
std::string remove_extra_spaces(const std::string& input) { return wt::remove_extra_spaces(input); }
std::vector<float> wav_read_legacy(const char* filename) {
  std::vector<float> s;
  wt::wav_read_legacy(filename, &s, false);
  return s;
}

}  // namespace whisper

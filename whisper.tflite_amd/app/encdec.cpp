// encdec — same command line as the reference's app/encdec.cpp:30-36:
//   encdec --model-prefix P --vocab V --input WAV
// prints the transcript followed by '\n' as the last line of stdout.  The reference pulls
// in the 11 kLoC CLI11 header for three required options; a minimal parser keeps the same
// flags (and --flag=value spelling) and exit status on a usage error.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>

#include <vector>

#include "whisper.tflite/whisper.h"
#include "wt_capi.h"

namespace {
void usage(const char* argv0) {
  std::cerr << "Usage: " << argv0 << " --model-prefix <prefix> --vocab <vocab.bin> --input <wav>\n"
            << "  --model-prefix  Model prefix (loads <prefix>.wtw)   REQUIRED\n"
            << "  --vocab         Path to vocabulary                 REQUIRED\n"
            << "  --input         Path to the 16 kHz mono WAV        REQUIRED\n"
            << "  --lang          language code of the prompt (default de, as the reference hard-codes)\n"
            << "  --english       English-only vocabulary ids (multilingual = false; the reference hard-codes true)\n"
            << "  --long          transcribe every 30 s window of the file, not only the first\n";
}
}  // namespace

int main(int argc, char* argv[]) {
  std::string model_prefix, vocab, input, lang;
  bool long_audio = false, english = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], v;
    if (a == "-h" || a == "--help") {
      usage(argv[0]);
      return 0;
    }
    if (a == "--long") {
      long_audio = true;
      continue;
    }
    if (a == "--english") {
      english = true;
      continue;
    }
    const size_t eq = a.find('=');
    if (eq != std::string::npos) {
      v = a.substr(eq + 1);
      a = a.substr(0, eq);
    } else if (i + 1 < argc) {
      v = argv[++i];
    } else {
      std::cerr << a << ": 1 required TEXT missing\n";
      return 106;
    }
    if (a == "--model-prefix") model_prefix = v;
    else if (a == "--vocab") vocab = v;
    else if (a == "--input") input = v;
    else if (a == "--lang") lang = v;
    else {
      std::cerr << "The following argument was not expected: " << a << "\n";
      usage(argv[0]);
      return 109;
    }
  }
  if (model_prefix.empty() || vocab.empty() || input.empty()) {
    std::cerr << (model_prefix.empty() ? "--model-prefix" : vocab.empty() ? "--vocab" : "--input")
              << " is required\n";
    usage(argv[0]);
    return 106;
  }
  using namespace whisper;  // NOLINT
  const bool multilingual = !english;  // true is hard-coded in the reference (app/encdec.cpp:47)
  std::unique_ptr<EncDec> engine;
  try {
    engine.reset(new EncDec(model_prefix, vocab, multilingual));
  } catch (const std::exception& e) {  // the reference lets it terminate the process
    std::cerr << "encdec: " << e.what() << "\n";
    return 1;
  }
  EncDec& encdec = *engine;
  if (!lang.empty()) {
    const int id = language_id(lang);
    if (wt_engine_set_option(encdec.handle(), "language", id) != WT_OK) {
      std::cerr << "--lang: unknown language code " << lang << "\n";
      return 105;
    }
  }
  std::string text;
  if (long_audio) {
    std::vector<float> pcm = wav_read_legacy(input.c_str());
    std::vector<char> buf(1 << 20);
    size_t len = 0;
    if (wt_transcribe_long_pcm(encdec.handle(), pcm.data(), pcm.size(), buf.data(), buf.size(), &len) != WT_OK) {
      std::cerr << "transcribe failed: " << wt_last_error(encdec.handle()) << "\n";
      return 1;
    }
    text.assign(buf.data(), len);
  } else {
    text = encdec.transcribe(input.c_str());
    // Engine::transcribe returns "" on failure like the reference (whisper.cpp:760): a device error must not
    // look like an empty transcript
    const char* err = wt_last_error(encdec.handle());
    if (err && *err) return 2;
  }
  std::cout << text << "\n";
  return 0;
}

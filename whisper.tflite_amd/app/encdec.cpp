// encdec — same command line as the reference's app/encdec.cpp:30-36:
//   encdec --model-prefix P --vocab V --input WAV
// prints the transcript followed by '\n' as the last line of stdout.  The reference pulls
// in the 11 kLoC CLI11 header for three required options; a minimal parser keeps the same
// flags (and --flag=value spelling) and exit status on a usage error.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "whisper.tflite/whisper.h"

namespace {
void usage(const char* argv0) {
  std::cerr << "Usage: " << argv0 << " --model-prefix <prefix> --vocab <vocab.bin> --input <wav>\n"
            << "  --model-prefix  Model prefix (loads <prefix>.wtw)   REQUIRED\n"
            << "  --vocab         Path to vocabulary                 REQUIRED\n"
            << "  --input         Path to the 16 kHz mono WAV        REQUIRED\n";
}
}  // namespace

int main(int argc, char* argv[]) {
  std::string model_prefix, vocab, input;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], v;
    if (a == "-h" || a == "--help") {
      usage(argv[0]);
      return 0;
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
  const bool multilingual = true;  // hard-coded in the reference (app/encdec.cpp:47)
  EncDec encdec(model_prefix, vocab, multilingual);
  const std::string text = encdec.transcribe(input.c_str());
  std::cout << text << "\n";
  return 0;
}

// minimal — same command line as the reference's app/minimal.cpp:21-46:
//   minimal <model prefix> <vocab file> <wav file>
// English-only Monolith engine (multilingual = false, :34), transcript with runs of spaces collapsed,
// printed between blank lines.  Exits non-zero when the engine reports a failure.
#include <cstdio>
#include <exception>
#include <string>

#include "whisper.tflite/whisper.h"
#include "wt_capi.h"

int main(int argc, char* argv[]) {
  if (argc != 4) {
    std::fprintf(stderr, "Usage: minimal <model prefix> <vocab file> <pcm_file name>\n");
    return 1;
  }
  try {
    whisper::Monolith monolith(argv[1], argv[2], /*multilingual=*/false);
    std::string text = monolith.transcribe(argv[3]);
    const char* err = wt_last_error(monolith.handle());
    if (err && *err) return 2;  // message already on stderr
    text = whisper::remove_extra_spaces(text);
    std::printf("\n%s\n\n", text.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "minimal: %s\n", e.what());
    return 1;
  }
  return 0;
}

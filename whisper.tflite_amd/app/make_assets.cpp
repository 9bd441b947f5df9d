// wt-make-assets — writes the two on-disk assets of the path when the upstream ones are
// absent (no network in this environment):
//   wt-make-assets weights <out.wtw> <tiny|tiny.en|base|micro> [seed]
//   wt-make-assets vocab   <out.bin> [n_tokens]
// and converts the reference's model files into the engine's weight file:
//   wt-make-assets convert <prefix> <out.wtw>     (<prefix>.encoder.tflite + <prefix>.decoder.tflite -> .wtw)
#include <cstdio>
#include <cstdlib>
#include <string>

#include "wt_capi.h"

int main(int argc, char** argv) {
  if (argc >= 4 && std::string(argv[1]) == "weights") {
    const uint64_t seed = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0;
    const int rc = wt_write_synthetic_weights(argv[2], argv[3], seed);
    if (rc != WT_OK) std::fprintf(stderr, "error: %s\n", wt_last_error(nullptr));
    return rc;
  }
  if (argc >= 3 && std::string(argv[1]) == "vocab") {
    const int n = argc > 3 ? std::atoi(argv[3]) : 50257;
    const int rc = wt_write_synthetic_vocab(argv[2], n);
    if (rc != WT_OK) std::fprintf(stderr, "error: %s\n", wt_last_error(nullptr));
    return rc;
  }
  if (argc >= 4 && std::string(argv[1]) == "convert") {
    const int rc = wt_convert_tflite(argv[2], argv[3]);
    if (rc != WT_OK) std::fprintf(stderr, "error: %s\n", wt_last_error(nullptr));
    return rc;
  }
  std::fprintf(stderr,
               "usage: %s weights <out.wtw> <tiny|tiny.en|base|micro> [seed]\n"
               "       %s vocab <out.bin> [n_tokens]\n"
               "       %s convert <prefix> <out.wtw>\n",
               argv[0], argv[0], argv[0]);
  return 2;
}

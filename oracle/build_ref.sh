#!/bin/bash
# Builds oracle/_ref/libwt_ref_frontend.so: the reference's OWN host-side functions,
# compiled from the sources where they lie under /root/reference.
#
# The reference's translation units do not compile as files (whisper.h:10-11 includes
# TensorFlow Lite headers and wav_util.cpp:14 includes dr_libs, neither present), and
# no stand-in headers are written for them.  The functions on the hot path that are
# pure standard C++ are instead streamed, by line range, straight from the reference
# files into the compiler's stdin together with a small extern "C" shim (below, ours).
# No reference source text is ever written to disk: only the .so lands in
# oracle/_ref/ (git-ignored; it travels to the GPU box like any other built .so).
#
#   whisper.h   24-107   constants, Vocab, Filters, Mel
#               208-212  LangKey / language_id / lang_code declarations
#               236-257  Reader, remove_extra_spaces, decode declarations
#   whisper.cpp 36-226   dft, fft, log_mel_spectrogram, transform_vocab_multilingual
#               346-361  the argmax lambda of Decoder::forward
#               405-665  language table, Reader, remove_extra_spaces, decode
#   wav_util.h  (whole)  WAVHeader
#   wav_util.cpp 18-87   wav_read_legacy
#
# Skipped silently (exit 0) when /root/reference is absent (the GPU box).
set -euo pipefail
REF=${WT_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ ! -f "$REF/whisper.tflite/whisper.cpp" ]; then
  echo "[build_ref] $REF not present; keeping any prebuilt oracle/_ref/" >&2
  exit 0
fi
mkdir -p "$HERE/_ref"
H="$REF/whisper.tflite/whisper.h"
C="$REF/whisper.tflite/whisper.cpp"
WH="$REF/whisper.tflite/wav_util.h"
WC="$REF/whisper.tflite/wav_util.cpp"
{
  echo '#include <math.h>'
  echo '#include <algorithm>'
  echo '#include <cassert>'
  echo '#include <cmath>'
  echo '#include <cstddef>'
  echo '#include <cstdint>'
  echo '#include <cstdio>'
  echo '#include <cstdlib>'
  echo '#include <cstring>'
  echo '#include <fstream>'
  echo '#include <iostream>'
  echo '#include <iterator>'
  echo '#include <map>'
  echo '#include <string>'
  echo '#include <thread>'
  echo '#include <utility>'
  echo '#include <vector>'
  sed -n '24,107p' "$H"
  echo 'void transform_vocab_multilingual(Vocab& vocab);'
  echo 'bool log_mel_spectrogram(const float*, int, int, int, int, int, int, Filters&, Mel&);'
  sed -n '208,212p;236,257p' "$H"
  echo '}  // namespace whisper'
  echo 'namespace whisper {'
  sed -n '36,226p' "$C"
  sed -n '346,361p' "$C"
  sed -n '405,665p' "$C"
  echo '}  // namespace whisper'
  sed -n '2,$p' "$WH"
  echo 'namespace whisper {'
  sed -n '18,87p' "$WC"
  echo '}  // namespace whisper'
  cat <<'SHIM'
// ---- extern "C" shim (ours): same signatures as oracle/wt_oracle.h with a ref_ prefix ----
struct ref_vocab { whisper::Vocab vocab; whisper::Filters filters; std::vector<char> bytes; };
extern "C" {
int ref_logmel(const float* samples, int n_samples, int fft_size, int fft_step, int n_mel,
               int n_threads, const float* filters, float* mel_out) {
  whisper::Filters f; f.n_mel = n_mel; f.n_fft = 1 + fft_size / 2;
  f.data.assign(filters, filters + size_t(f.n_mel) * f.n_fft);
  whisper::Mel mel;
  whisper::log_mel_spectrogram(samples, n_samples, 16000, fft_size, fft_step, n_mel, n_threads, f, mel);
  std::copy(mel.data.begin(), mel.data.end(), mel_out);
  return 0;
}
long ref_wav_read_legacy(const char* path, float* out, long cap) {
  std::vector<float> v = whisper::wav_read_legacy(path);
  if (v.empty()) return -1;
  for (long i = 0; i < (long)v.size() && i < cap; ++i) out[i] = v[i];
  return (long)v.size();
}
int64_t ref_argmax_last(const float* begin, int64_t n) { return whisper::argmax(begin, begin + n).first; }
int ref_language_id(const char* code) { return whisper::language_id(code); }
const char* ref_lang_code(int id) { return whisper::lang_code(id).c_str(); }
int ref_language_count(void) { return (int)whisper::language_meta.size(); }
ref_vocab* ref_vocab_open(const char* path, int multilingual) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return nullptr;
  auto* v = new ref_vocab;
  v->bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  const char* ptr = v->bytes.data() + sizeof(int64_t);  // as EncDec::EncDec does
  whisper::Reader reader(ptr, multilingual != 0);
  reader.read(v->filters, v->vocab);
  return v;
}
void ref_vocab_close(ref_vocab* v) { delete v; }
void ref_vocab_info(const ref_vocab* v, int32_t out[9]) {
  const whisper::Vocab& k = v->vocab;
  const int32_t vals[9] = {k.n_vocab, k.token_eot, k.token_sot, k.token_translate, k.token_transcribe,
                           k.token_prev, k.token_solm, k.token_not, k.token_beg};
  memcpy(out, vals, sizeof(vals));
}
void ref_vocab_filters_shape(const ref_vocab* v, int32_t* n_mel, int32_t* n_fft) {
  *n_mel = v->filters.n_mel; *n_fft = v->filters.n_fft;
}
const float* ref_vocab_filters(const ref_vocab* v) { return v->filters.data.data(); }
int ref_vocab_size(const ref_vocab* v) { return (int)v->vocab.id_to_token.size(); }
int ref_vocab_token(const ref_vocab* v, int id, char* out, int cap) {
  auto it = v->vocab.id_to_token.find(id);
  if (it == v->vocab.id_to_token.end()) return -1;
  int len = (int)it->second.size();
  memcpy(out, it->second.data(), std::min(len, cap));
  return len;
}
long ref_decode_text(const ref_vocab* v, const int64_t* ids, int n, int omit_special, char* out, long cap) {
  for (int i = 0; i < n; ++i)  // the reference asserts (UB under NDEBUG) on a missing id
    if (v->vocab.id_to_token.find((int)ids[i]) == v->vocab.id_to_token.end()) return -1;
  std::string s = whisper::decode(v->vocab, ids, ids + n, omit_special != 0);
  long len = (long)s.size();
  memcpy(out, s.data(), std::min(len, cap));
  return len;
}
long ref_remove_extra_spaces(const char* in, char* out, long cap) {
  std::string s = whisper::remove_extra_spaces(in);
  long len = (long)s.size();
  memcpy(out, s.data(), std::min(len, cap));
  return len;
}
}
SHIM
} | g++ -std=c++17 -O3 -DNDEBUG -fPIC -shared -pthread -w -x c++ - -o "$HERE/_ref/libwt_ref_frontend.so"
echo "[build_ref] built $HERE/_ref/libwt_ref_frontend.so" >&2

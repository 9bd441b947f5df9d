// C++ engine surface (include/whisper.tflite/whisper.h) as a thin wrapper over the C ABI,
// re-creating the reference's observable behaviour: exceptions for missing files
// (mmap_file.cpp:16-29), "" on soft failure (whisper.cpp:760), nullptr + stderr for an
// unknown engine type (whisper.cpp:785-789).
#include "whisper.tflite/whisper.h"

#include <cmath>
#include <cstdio>
#include <iostream>
#include <stdexcept>

#include "host_util.h"
#include "wt_capi.h"

namespace whisper {

namespace {
wt_engine* open_engine(int type, const std::string& model_prefix, const std::string& vocab_path, bool multilingual) {
  wt_engine* h = nullptr;
  const int rc = wt_engine_create(type, model_prefix.c_str(), vocab_path.c_str(), multilingual ? 1 : 0, 0, &h);
  if (rc != WT_OK) throw std::runtime_error(wt_last_error(nullptr));
  return h;
}

// Engine::transcribe(std::vector<float>&) of both engine types (whisper.cpp:752-769, :692-738)
std::string transcribe_samples(wt_engine* h, std::vector<float>& samples) {
  samples.resize(size_t(kSampleRate) * kChunkSize, 0);  // whisper.cpp:753 mutates the caller's vector
  std::string text(8192, '\0');
  size_t len = 0;
  int rc = wt_transcribe_pcm(h, samples.data(), samples.size(), &text[0], text.size(), &len);
  if (rc == WT_ERR_BUFFER) {
    text.assign(len + 1, '\0');
    rc = wt_transcribe_pcm(h, samples.data(), samples.size(), &text[0], text.size(), &len);
  }
  if (rc != WT_OK) {
    std::fprintf(stderr, "transcribe failed: %s\n", wt_last_error(h));
    return "";
  }
  text.resize(len);
  return text;
}

std::string transcribe_wav(wt_engine* h, const char* wave_file) {
  std::vector<float> pcmf32 = wav_read_legacy(wave_file);
  pcmf32.resize(size_t(kSampleRate) * kChunkSize, 0);  // whisper.cpp:773, :687
  return transcribe_samples(h, pcmf32);
}

wt::VocabData to_data(const Vocab& v) {
  wt::VocabData d;
  d.id_to_token = v.id_to_token;
  d.n_vocab = v.n_vocab, d.token_eot = v.token_eot, d.token_sot = v.token_sot;
  d.token_translate = v.token_translate, d.token_transcribe = v.token_transcribe;
  d.token_prev = v.token_prev, d.token_solm = v.token_solm, d.token_not = v.token_not, d.token_beg = v.token_beg;
  return d;
}

void from_data(const wt::VocabData& d, Vocab* v) {
  v->id_to_token = d.id_to_token;
  v->n_vocab = d.n_vocab, v->token_eot = d.token_eot, v->token_sot = d.token_sot;
  v->token_translate = d.token_translate, v->token_transcribe = d.token_transcribe;
  v->token_prev = d.token_prev, v->token_solm = d.token_solm, v->token_not = d.token_not, v->token_beg = d.token_beg;
}
}  // namespace

EncDec::EncDec(const std::string& model_prefix, const std::string& vocab_path, bool multilingual)
    : handle_(open_engine(WT_ENGINE_ENCDEC, model_prefix, vocab_path, multilingual)) {}
EncDec::~EncDec() { wt_engine_destroy(handle_); }
std::string EncDec::transcribe(std::vector<float>& samples) { return transcribe_samples(handle_, samples); }
std::string EncDec::transcribe(const char* waveFile) { return transcribe_wav(handle_, waveFile); }

Monolith::Monolith(const std::string& model_prefix, const std::string& vocab_path, bool multilingual)
    : handle_(open_engine(WT_ENGINE_MONOLITH, model_prefix, vocab_path, multilingual)) {}
Monolith::~Monolith() { wt_engine_destroy(handle_); }
std::string Monolith::transcribe(std::vector<float>& samples) { return transcribe_samples(handle_, samples); }
std::string Monolith::transcribe(const char* waveFile) { return transcribe_wav(handle_, waveFile); }

Engine* create_engine(EngineType type, const char* model_prefix, const char* vocab_path,
                      bool multilingual) {
  switch (type) {
    case EngineType::Monolith:
      return new Monolith(model_prefix, vocab_path, multilingual);
    case EngineType::EncDec:
      return new EncDec(model_prefix, vocab_path, multilingual);
    default:
      std::fprintf(stderr, "Unknown engine-type\n");
      break;
  }
  return nullptr;
}

// ----------------------------------------------------------- language table ---
std::vector<LangKey> language_meta = [] {
  std::vector<LangKey> v;
  for (int i = 0; i < wt::language_count(); ++i) v.emplace_back(wt::lang_code(size_t(i)), wt::lang_name(size_t(i)));
  return v;
}();
int language_id(const std::string& code) { return wt::language_id(code); }
const std::string& lang_code(size_t id) { return wt::lang_code(id); }

// ------------------------------------------------------------ vocab / text ---
void transform_vocab_multilingual(Vocab& vocab) {
  wt::VocabData d = to_data(vocab);
  wt::transform_vocab_multilingual(&d);
  d.id_to_token.clear();
  vocab.n_vocab = d.n_vocab, vocab.token_eot = d.token_eot, vocab.token_sot = d.token_sot;
  vocab.token_prev = d.token_prev, vocab.token_solm = d.token_solm, vocab.token_not = d.token_not;
  vocab.token_beg = d.token_beg;
}

void Reader::read(Filters& filters, Vocab& vocab) {
  wt::FilterBank fb;
  wt::VocabData d;
  // unbounded (the reference trusts the buffer): any end far beyond a real mapping
  const char* end = head_ + (size_ == static_cast<size_t>(-1) ? (size_t(1) << 46) : size_);
  wt::parse_vocab(head_, end, multilingual_, &fb, &d);
  filters.n_mel = fb.n_mel;
  filters.n_fft = fb.n_fft;
  filters.data = std::move(fb.data);
  from_data(d, &vocab);
}

std::string remove_extra_spaces(const std::string& input) { return wt::remove_extra_spaces(input); }

template <class Int>
std::string decode(const Vocab& vocab, const Int* begin, const Int* end, bool omit_special_tokens) {
  std::vector<int64_t> ids(begin, end);
  // An id without a vocabulary entry: the reference asserts (whisper.cpp:642; undefined behaviour in a release
  // build).  Here it is an exception, in line with the C ABI, where wt_vocab_decode returns WT_ERR_INVALID_ARG.
  bool missing = false;
  std::string text = wt::decode_tokens(to_data(vocab), ids.data(), int(ids.size()), omit_special_tokens, &missing);
  if (missing) throw std::out_of_range("whisper::decode: token id without a vocabulary entry");
  return text;
}
template std::string decode(const Vocab& vocab, const int* begin, const int* end, bool omit_special_tokens);
template std::string decode(const Vocab& vocab, const int64_t* begin, const int64_t* end, bool omit_special_tokens);

std::string decode(const Vocab& vocab, const std::vector<int64_t>& generated, bool omit_special_tokens) {
  return decode(vocab, generated.data(), generated.data() + generated.size(), omit_special_tokens);
}

std::vector<float> wav_read_legacy(const char* filename) {
  std::vector<float> s;
  wt::wav_read_legacy(filename, &s, false);
  return s;
}

// --------------------------------------------------------------- front end ---
bool log_mel_spectrogram(const float* samples, int n_samples, int sample_rate, int fft_size, int fft_step,
                         int n_mel, int /*n_threads*/, Filters& filters, Mel& mel) {
  (void)sample_rate;  // the reference never reads it either (whisper.cpp:109-216)
  if (fft_size != kNFFT || fft_step != kHopLength || n_mel != filters.n_mel ||
      size_t(filters.n_mel) * size_t(filters.n_fft) != filters.data.size()) {
    std::cerr << "log_mel_spectrogram: only the reference's fixed geometry (fft 400, hop 160, 80 x 201 filters) runs on the GPU front end\n";
    return false;
  }
  mel.n_mel = n_mel;
  mel.n_len = n_samples / fft_step;  // whisper.cpp:122-124
  mel.data.assign(size_t(mel.n_mel) * size_t(mel.n_len), 0.0f);
  int n_len = 0;
  const int rc = wt_log_mel_spectrogram(samples, n_samples, filters.data.data(), filters.n_mel, filters.n_fft, 0,
                                        mel.data.data(), mel.data.size(), &n_len);
  if (rc != WT_OK) {
    std::cerr << "log_mel_spectrogram: " << wt_last_error(nullptr) << '\n';
    return false;
  }
  return true;
}

// Host-side transforms of the reference header, kept for source compatibility (nothing on the hot path calls
// them).  dft: direct evaluation; fft: even/odd recursion down to an odd length, then dft — the structure
// whisper.cpp:58-106 describes, written iteratively over index strides.
void print(const std::vector<float>& a) {
  std::cout << "The vector elements are: ";
  for (float x : a) std::cout << x << ' ';
}

void dft(const std::vector<float>& in, std::vector<float>& out) {
  const int n = static_cast<int>(in.size());
  out.assign(size_t(n) * 2, 0.0f);
  for (int k = 0; k < n; ++k) {
    float re = 0.0f, im = 0.0f;
    for (int t = 0; t < n; ++t) {
      const float angle = static_cast<float>(2 * M_PI * k * t / n);
      re += in[t] * std::cos(angle);
      im -= in[t] * std::sin(angle);
    }
    out[2 * k] = re;
    out[2 * k + 1] = im;
  }
}

void fft(const std::vector<float>& in, std::vector<float>& out) {
  const int n = static_cast<int>(in.size());
  out.assign(size_t(n) * 2, 0.0f);
  if (n == 0) return;
  if (n == 1) {
    out[0] = in[0];
    return;
  }
  if (n % 2 == 1) {
    dft(in, out);
    return;
  }
  std::vector<float> half[2], spec[2];
  for (int i = 0; i < n; ++i) half[i & 1].push_back(in[i]);
  fft(half[0], spec[0]);
  fft(half[1], spec[1]);
  for (int k = 0; k < n / 2; ++k) {
    const float theta = static_cast<float>(2 * M_PI * k / n);
    const float c = std::cos(theta), s = -std::sin(theta);
    const float er = spec[0][2 * k], ei = spec[0][2 * k + 1], orr = spec[1][2 * k], oi = spec[1][2 * k + 1];
    const float tr = c * orr - s * oi, ti = c * oi + s * orr;
    out[2 * k] = er + tr;
    out[2 * k + 1] = ei + ti;
    out[2 * (k + n / 2)] = er - tr;
    out[2 * (k + n / 2) + 1] = ei - ti;
  }
}

}  // namespace whisper

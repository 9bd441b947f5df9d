#include "host_util.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

namespace wt {

// ------------------------------------------------------- language table ---
namespace {
struct Lang {
  std::string code, name;
};
// OpenAI tokenizer LANGUAGES order; the position is the language id the reference adds
// to 50259 for the prompt's language token (whisper.cpp:327, table :405-508).
const std::vector<Lang>& languages() {
  static const std::vector<Lang> table = [] {
    static const char* const kPairs =
        "en english|zh chinese|de german|es spanish|ru russian|ko korean|fr french|ja japanese|"
        "pt portuguese|tr turkish|pl polish|ca catalan|nl dutch|ar arabic|sv swedish|it italian|"
        "id indonesian|hi hindi|fi finnish|vi vietnamese|he hebrew|uk ukrainian|el greek|"
        "ms malay|cs czech|ro romanian|da danish|hu hungarian|ta tamil|no norwegian|th thai|"
        "ur urdu|hr croatian|bg bulgarian|lt lithuanian|la latin|mi maori|ml malayalam|cy welsh|"
        "sk slovak|te telugu|fa persian|lv latvian|bn bengali|sr serbian|az azerbaijani|"
        "sl slovenian|kn kannada|et estonian|mk macedonian|br breton|eu basque|is icelandic|"
        "hy armenian|ne nepali|mn mongolian|bs bosnian|kk kazakh|sq albanian|sw swahili|"
        "gl galician|mr marathi|pa punjabi|si sinhala|km khmer|sn shona|yo yoruba|so somali|"
        "af afrikaans|oc occitan|ka georgian|be belarusian|tg tajik|sd sindhi|gu gujarati|"
        "am amharic|yi yiddish|lo lao|uz uzbek|fo faroese|ht haitian creole|ps pashto|"
        "tk turkmen|nn nynorsk|mt maltese|sa sanskrit|lb luxembourgish|my myanmar|bo tibetan|"
        "tl tagalog|mg malagasy|as assamese|tt tatar|haw hawaiian|ln lingala|ha hausa|"
        "ba bashkir|jw javanese|su sundanese|yue cantonese";
    std::vector<Lang> v;
    std::string all(kPairs);
    size_t start = 0;
    while (start <= all.size()) {
      size_t bar = all.find('|', start);
      if (bar == std::string::npos) bar = all.size();
      const std::string item = all.substr(start, bar - start);
      const size_t sp = item.find(' ');
      v.push_back({item.substr(0, sp), item.substr(sp + 1)});
      start = bar + 1;
    }
    return v;
  }();
  return table;
}
}  // namespace

int language_count() { return static_cast<int>(languages().size()); }
int language_id(const std::string& code) {
  const auto& t = languages();
  for (size_t i = 0; i < t.size(); ++i)
    if (t[i].code == code) return static_cast<int>(i);
  return static_cast<int>(t.size());
}
const std::string& lang_code(size_t id) { return languages().at(id).code; }
const std::string& lang_name(size_t id) { return languages().at(id).name; }

// ---------------------------------------------------- vocab/filter file ---
namespace {
template <class T>
T take(const char*& p, const char* end) {
  if (size_t(end - p) < sizeof(T)) throw std::runtime_error("vocab file truncated");
  T v;
  std::memcpy(&v, p, sizeof(T));
  p += sizeof(T);
  return v;
}
}  // namespace

void transform_vocab_multilingual(VocabData* vocab) {  // six ids move up by one; translate/transcribe stay
  vocab->n_vocab = 51865;
  vocab->token_eot += 1;
  vocab->token_sot += 1;
  vocab->token_prev += 1;
  vocab->token_solm += 1;
  vocab->token_not += 1;
  vocab->token_beg += 1;
}

void read_vocab_file(const std::string& path, bool multilingual, FilterBank* filters,
                     VocabData* vocab) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) throw std::runtime_error("Failed to open file: " + path);
  std::vector<char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const char* p = bytes.data();
  const char* const end = p + bytes.size();
  (void)take<uint64_t>(p, end);  // payload size written by the asset dumper; unused
  parse_vocab(p, end, multilingual, filters, vocab);
}

void parse_vocab(const char* p, const char* end, bool multilingual, FilterBank* filters, VocabData* vocab) {
  (void)take<uint32_t>(p, end);  // magic: the reference reads it and checks nothing
  filters->n_mel = take<int32_t>(p, end);
  filters->n_fft = take<int32_t>(p, end);
  if (filters->n_mel <= 0 || filters->n_fft <= 0 || filters->n_mel > 1024 || filters->n_fft > 65536)
    throw std::runtime_error("vocab file: implausible filter shape");
  const size_t nf = size_t(filters->n_mel) * size_t(filters->n_fft);
  if (size_t(end - p) < nf * sizeof(float)) throw std::runtime_error("vocab file truncated");
  filters->data.resize(nf);
  std::memcpy(filters->data.data(), p, nf * sizeof(float));
  p += nf * sizeof(float);

  *vocab = VocabData();
  const int32_t n_file = take<int32_t>(p, end);
  if (n_file < 0) throw std::runtime_error("vocab file: negative token count");
  vocab->n_vocab = n_file;
  if (multilingual) transform_vocab_multilingual(vocab);
  for (int i = 0; i < n_file; ++i) {
    const uint32_t len = take<uint32_t>(p, end);
    if (len > 255) throw std::runtime_error("vocab file: token longer than 255 bytes");
    if (size_t(end - p) < len) throw std::runtime_error("vocab file truncated");
    // the reference round-trips through a C string, so an embedded NUL ends the token
    vocab->id_to_token[i] = std::string(std::string(p, len).c_str());
    p += len;
  }
  const int n_expected = 51864 + (multilingual ? 1 : 0);
  for (int i = n_file; i < n_expected; ++i) {
    std::string w;
    if (i > vocab->token_beg) {
      w = "<|TT" + std::to_string(i - vocab->token_beg) + "|>";
    } else if (i == vocab->token_eot) {
      w = "<|endoftranscript|>";
    } else if (i == vocab->token_sot) {
      w = "<|startoftranscript_|>";
    } else if (i == vocab->token_prev) {
      w = "<|PREV|>";
    } else if (i == vocab->token_not) {
      w = "<|notimestamps|>";
    } else if (i == vocab->token_beg) {
      w = "<|timestampbegin|>";
    } else if (i == vocab->token_translate) {
      w = "<|translate|>";
    } else if (i == vocab->token_transcribe) {
      w = "<|transcribe|>";
    } else if (i > vocab->token_sot && i < vocab->token_translate) {
      const size_t lang = size_t(i - (vocab->token_sot + 1));
      w = "<|lang-" + (lang < size_t(language_count()) ? lang_code(lang) : std::string("??")) + "|>";
    } else {
      w = "<|e" + std::to_string(i) + "|>";
    }
    vocab->id_to_token[i] = w;
  }
}

void write_vocab_file(const std::string& path, const FilterBank& filters,
                      const std::vector<std::string>& tokens) {
  std::vector<char> payload;
  auto put = [&payload](const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    payload.insert(payload.end(), c, c + n);
  };
  const uint32_t magic = 0x5553454e;  // "USEN", the value the Java twin expects
  put(&magic, 4);
  const int32_t n_mel = filters.n_mel, n_fft = filters.n_fft;
  put(&n_mel, 4);
  put(&n_fft, 4);
  put(filters.data.data(), filters.data.size() * sizeof(float));
  const int32_t n_vocab = static_cast<int32_t>(tokens.size());
  put(&n_vocab, 4);
  for (const std::string& t : tokens) {
    const uint32_t len = static_cast<uint32_t>(std::min<size_t>(t.size(), 255));
    put(&len, 4);
    put(t.data(), len);
  }
  std::ofstream f(path, std::ios::binary);
  if (!f.is_open()) throw std::runtime_error("Failed to open file for writing: " + path);
  const uint64_t size = payload.size();
  f.write(reinterpret_cast<const char*>(&size), 8);
  f.write(payload.data(), static_cast<std::streamsize>(payload.size()));
}

FilterBank make_slaney_filterbank(int n_mel, int n_fft_size, int sample_rate) {
  const int n_bins = 1 + n_fft_size / 2;
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  auto hz_to_mel = [&](double f) {
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
  };
  auto mel_to_hz = [&](double m) {
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
  };
  const double mel_lo = hz_to_mel(0.0), mel_hi = hz_to_mel(sample_rate / 2.0);
  std::vector<double> pts(n_mel + 2);
  for (int i = 0; i < n_mel + 2; ++i) pts[i] = mel_to_hz(mel_lo + (mel_hi - mel_lo) * i / (n_mel + 1));
  FilterBank fb;
  fb.n_mel = n_mel;
  fb.n_fft = n_bins;
  fb.data.assign(size_t(n_mel) * n_bins, 0.0f);
  for (int i = 0; i < n_mel; ++i) {
    const double enorm = 2.0 / (pts[i + 2] - pts[i]);
    for (int k = 0; k < n_bins; ++k) {
      const double f = (sample_rate / 2.0) * k / (n_bins - 1);
      const double lower = (f - pts[i]) / (pts[i + 1] - pts[i]);
      const double upper = (pts[i + 2] - f) / (pts[i + 2] - pts[i + 1]);
      const double w = std::max(0.0, std::min(lower, upper));
      fb.data[size_t(i) * n_bins + k] = static_cast<float>(w * enorm);
    }
  }
  return fb;
}

std::vector<std::string> make_synthetic_tokens(int n_tokens) {
  std::vector<std::string> t;
  t.reserve(n_tokens);
  for (int i = 0; i < n_tokens; ++i) {
    if (i >= 33 && i < 127) {
      t.push_back(std::string(1, static_cast<char>(i)));
    } else {
      t.push_back(" t" + std::to_string(i));
    }
  }
  return t;
}

// ------------------------------------------------------------------ WAV ---
bool wav_read_legacy(const std::string& path, std::vector<float>* samples, bool verbose) {
  samples->clear();
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) {
    std::cerr << "Failed to open file: " << path << '\n';
    return false;
  }
  unsigned char h[36];
  std::memset(h, 0, sizeof(h));
  f.read(reinterpret_cast<char*>(h), sizeof(h));
  if (std::memcmp(h + 0, "RIFF", 4) != 0 || std::memcmp(h + 8, "WAVE", 4) != 0 ||
      std::memcmp(h + 12, "fmt ", 4) != 0) {
    std::cerr << "Not a valid WAV file: " << path << '\n';
    return false;
  }
  auto u16 = [&h](int o) { return uint32_t(h[o]) | (uint32_t(h[o + 1]) << 8); };
  auto u32 = [&h](int o) {
    return uint32_t(h[o]) | (uint32_t(h[o + 1]) << 8) | (uint32_t(h[o + 2]) << 16) | (uint32_t(h[o + 3]) << 24);
  };
  const uint32_t riff_size = u32(4), format = u16(20), channels = u16(22), rate = u32(24);
  const uint32_t block_align = u16(32), bits = u16(34);
  if (verbose) {  // the reference prints these four lines to stdout unconditionally
    std::cerr << "Audio Format: " << (format == 1 ? "PCM" : format == 3 ? "IEEE Float" : "Unknown")
              << "\nNum Channels: " << channels << "\nSample Rate: " << rate
              << "\nBits Per Sample: " << bits << '\n';
  }
  if (block_align == 0) {
    std::cerr << "Not a valid WAV file (block_align 0): " << path << '\n';
    return false;
  }
  // Reference quirks kept on purpose (wav_util.cpp:61-80): the sample count comes from
  // the RIFF size field, and the read starts right after the 36-byte header, so the 8-byte
  // "data" chunk header is decoded as the first four PCM16 samples and the tail past EOF
  // stays zero.
  const uint32_t n = riff_size / block_align;
  samples->assign(n, 0.0f);
  if (format == 1) {
    std::vector<int16_t> pcm(n, 0);
    f.read(reinterpret_cast<char*>(pcm.data()),
           static_cast<std::streamsize>(std::min<uint64_t>(riff_size, uint64_t(n) * 2)));
    for (uint32_t i = 0; i < n; ++i) (*samples)[i] = static_cast<float>(pcm[i]) / 32767.0f;
  } else {
    f.read(reinterpret_cast<char*>(samples->data()),
           static_cast<std::streamsize>(std::min<uint64_t>(riff_size, uint64_t(n) * 4)));
  }
  return true;
}

bool wav_write_pcm16(const std::string& path, const std::vector<float>& samples, int sample_rate) {
  std::ofstream f(path, std::ios::binary);
  if (!f.is_open()) return false;
  const uint32_t data_bytes = static_cast<uint32_t>(samples.size() * 2);
  auto w32 = [&f](uint32_t v) { f.write(reinterpret_cast<const char*>(&v), 4); };
  auto w16 = [&f](uint16_t v) { f.write(reinterpret_cast<const char*>(&v), 2); };
  f.write("RIFF", 4);
  w32(36 + data_bytes);
  f.write("WAVE", 4);
  f.write("fmt ", 4);
  w32(16);
  w16(1);
  w16(1);
  w32(static_cast<uint32_t>(sample_rate));
  w32(static_cast<uint32_t>(sample_rate) * 2);
  w16(2);
  w16(16);
  f.write("data", 4);
  w32(data_bytes);
  for (float s : samples) {
    const float c = std::max(-1.0f, std::min(1.0f, s));
    w16(static_cast<uint16_t>(static_cast<int16_t>(std::lrintf(c * 32767.0f))));
  }
  return static_cast<bool>(f);
}

// ----------------------------------------------------------------- text ---
std::string decode_tokens(const VocabData& vocab, const int64_t* ids, int n,
                          bool omit_special_tokens, bool* missing) {
  std::string surface;
  if (missing) *missing = false;
  for (int i = 0; i < n; ++i) {
    const int id = static_cast<int>(ids[i]);
    if (!omit_special_tokens || id < vocab.token_eot) {
      auto it = vocab.id_to_token.find(id);
      if (it == vocab.id_to_token.end()) {  // the reference asserts; report instead
        if (missing) *missing = true;
      } else {
        surface += it->second;
      }
    }
    if (id == vocab.token_eot) break;  // EOT is appended, then decoding stops
  }
  return surface;
}

std::string remove_extra_spaces(const std::string& in) {
  std::string out;
  out.reserve(in.size());
  bool prev_space = false;
  for (char c : in) {
    if (c != ' ' || !prev_space) out += c;
    prev_space = (c == ' ');
  }
  return out;
}

}  // namespace wt

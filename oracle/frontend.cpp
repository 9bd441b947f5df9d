// ORACLE — TEST INFRASTRUCTURE ONLY (see wt_oracle.h).
//
// CPU restatement of the reference's host-side pipeline stages.  This TU is
// compiled with -ffp-contract=off and without -ffast-math so that every float
// operation rounds exactly where the reference's does (the reference builds with
// default flags: no FMA contraction on baseline x86-64).
//
//   log-mel            whisper.tflite/whisper.cpp:109-216 (fft :58-106, dft :37-54)
//   WAV reader         whisper.tflite/wav_util.cpp:18-87, header wav_util.h:9-21
//   vocab/filter file  whisper.tflite/whisper.cpp:519-611, :218-226, :746-749
//   text decode        whisper.tflite/whisper.cpp:634-665, :613-631
//   language table     whisper.tflite/whisper.cpp:405-517
//   argmax             whisper.tflite/whisper.cpp:346-361
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "wt_oracle.h"

namespace {

// ---------------------------------------------------------------- log-mel ---

// Twiddle tables for one transform size, evaluated exactly as the reference
// evaluates them per call: the angle is formed in double, narrowed to float,
// and cosf/sinf are taken of the float (std::cos(float) is the float overload).
struct FftPlan {
  int n_top = 0;
  // level[l] holds N_l/2 (cos, -sin) pairs for N_l = n_top >> l while N_l is even
  std::vector<std::vector<float>> tw_re, tw_im;
  int leaf = 1;                         // odd size at which the recursion bottoms out
  std::vector<float> leaf_cos, leaf_sin;  // [k][n] for the naive DFT leaf

  explicit FftPlan(int n) : n_top(n) {
    int cur = n;
    while (cur > 1 && cur % 2 == 0) {
      std::vector<float> re(cur / 2), im(cur / 2);
      for (int k = 0; k < cur / 2; ++k) {
        double a = 2 * M_PI;  // whisper.cpp:92 `2 * M_PI * k / N`, left to right in double
        a = a * k;
        a = a / cur;
        const float theta = static_cast<float>(a);
        re[k] = cosf(theta);
        im[k] = -sinf(theta);
      }
      tw_re.push_back(re);
      tw_im.push_back(im);
      cur /= 2;
    }
    leaf = cur;
    if (leaf > 1) {
      leaf_cos.resize(size_t(leaf) * leaf);
      leaf_sin.resize(size_t(leaf) * leaf);
      for (int k = 0; k < leaf; ++k) {
        for (int j = 0; j < leaf; ++j) {
          double a = 2 * M_PI;  // whisper.cpp:46 `2 * M_PI * k * n / N`
          a = a * k;
          a = a * j;
          a = a / leaf;
          const float angle = static_cast<float>(a);
          leaf_cos[size_t(k) * leaf + j] = cosf(angle);
          leaf_sin[size_t(k) * leaf + j] = sinf(angle);
        }
      }
    }
  }
};

// Decimation in time on a strided view: in[0], in[stride], ... (n of them) plays the
// role of the reference's freshly copied even/odd vectors.  out gets 2*n floats
// (re, im interleaved); scratch must offer 4*n floats below the top level.
void fft_strided(const FftPlan& plan, int level, const float* in, int stride, int n, float* out,
                 float* scratch) {
  if (n == 1) {  // whisper.cpp:63-67
    out[0] = in[0];
    out[1] = 0;
    return;
  }
  if (n % 2 == 1) {  // whisper.cpp:69-72 -> dft()
    for (int k = 0; k < n; ++k) {
      float re = 0;
      float im = 0;
      const float* c = &plan.leaf_cos[size_t(k) * n];
      const float* s = &plan.leaf_sin[size_t(k) * n];
      for (int j = 0; j < n; ++j) {
        const float x = in[size_t(j) * stride];
        re += x * c[j];
        im -= x * s[j];
      }
      out[2 * k + 0] = re;
      out[2 * k + 1] = im;
    }
    return;
  }
  const int h = n / 2;
  float* ev = scratch;          // 2*h floats
  float* od = scratch + 2 * h;  // 2*h floats
  float* deeper = scratch + 4 * h;
  fft_strided(plan, level + 1, in, stride * 2, h, ev, deeper);
  fft_strided(plan, level + 1, in + stride, stride * 2, h, od, deeper);
  const float* twr = plan.tw_re[level].data();
  const float* twi = plan.tw_im[level].data();
  for (int k = 0; k < h; ++k) {  // whisper.cpp:91-105
    const float re = twr[k];
    const float im = twi[k];
    const float re_odd = od[2 * k + 0];
    const float im_odd = od[2 * k + 1];
    out[2 * k + 0] = ev[2 * k + 0] + re * re_odd - im * im_odd;
    out[2 * k + 1] = ev[2 * k + 1] + re * im_odd + im * re_odd;
    out[2 * (k + h) + 0] = ev[2 * k + 0] - re * re_odd + im * im_odd;
    out[2 * (k + h) + 1] = ev[2 * k + 1] - re * im_odd - im * re_odd;
  }
}

}  // namespace

extern "C" int wto_logmel(const float* samples, int n_samples, int fft_size, int fft_step,
                          int n_mel, int n_threads, const float* filters, float* mel_out) {
  std::vector<float> hann(fft_size);
  for (int i = 0; i < fft_size; ++i) {  // whisper.cpp:117-120: double cos, float store
    hann[i] = static_cast<float>(0.5 * (1.0 - cos((2.0 * M_PI * i) / fft_size)));
  }
  const int n_len = n_samples / fft_step;  // :123
  const int n_fft = 1 + fft_size / 2;      // :129
  const FftPlan plan(fft_size);
  if (n_threads < 1) n_threads = 1;

  auto work = [&](int ith) {
    std::vector<float> frame(fft_size, 0.0f);
    std::vector<float> spec(2 * size_t(fft_size));
    std::vector<float> scratch(8 * size_t(fft_size));
    for (int i = ith; i < n_len; i += n_threads) {  // :144 frame striding
      const int offset = i * fft_step;
      for (int j = 0; j < fft_size; ++j) {  // :148-154 window, zero past the end
        frame[j] = (offset + j < n_samples) ? hann[j] * samples[offset + j] : 0.0f;
      }
      fft_strided(plan, 0, frame.data(), 1, fft_size, spec.data(), scratch.data());
      for (int j = 0; j < fft_size; ++j) {  // :159-162 power, in place over the low half
        spec[j] = spec[2 * j + 0] * spec[2 * j + 0] + spec[2 * j + 1] * spec[2 * j + 1];
      }
      for (int j = 1; j < fft_size / 2; ++j) {  // :164-166 fold the mirror bins in
        spec[j] += spec[fft_size - j];
      }
      for (int j = 0; j < n_mel; ++j) {  // :169-185
        double sum = 0.0;
        const float* f = filters + size_t(j) * n_fft;
        for (int k = 0; k < n_fft; ++k) {
          const float prod = spec[k] * f[k];  // float product, double accumulate
          sum += prod;
        }
        const float eps = 1e-10f;  // :176 a float constant compared in double
        if (sum < eps) sum = eps;
        sum = log10(sum);
        mel_out[size_t(j) * n_len + i] = static_cast<float>(sum);
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();

  // :198-213 global max, clamp to max-8, (x+4)/4 — comparisons and arithmetic in double
  const size_t total = size_t(n_mel) * n_len;
  double mmax = -1e20;
  for (size_t i = 0; i < total; ++i) {
    if (mel_out[i] > mmax) mmax = mel_out[i];
  }
  mmax -= 8.0;
  for (size_t i = 0; i < total; ++i) {
    if (mel_out[i] < mmax) mel_out[i] = static_cast<float>(mmax);
    mel_out[i] = static_cast<float>((mel_out[i] + 4.0) / 4.0);
  }
  return 0;
}

// -------------------------------------------------------------------- WAV ---

extern "C" long wto_wav_read_legacy(const char* path, float* out, long cap) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return -1;  // wav_util.cpp:22-25
#pragma pack(push, 1)
  struct Header {  // wav_util.h:9-21, 36 bytes packed
    char riff[4];
    uint32_t wav_size;
    char wave[4];
    char fmt[4];
    uint32_t fmt_chunk_size;
    uint16_t audio_format, num_channels;
    uint32_t sample_rate, byte_rate;
    uint16_t block_align, bits_per_sample;
  } h;
#pragma pack(pop)
  static_assert(sizeof(Header) == 36, "packed header");
  std::memset(&h, 0, sizeof(h));
  size_t got = std::fread(&h, 1, sizeof(h), f);
  (void)got;
  if (std::strncmp(h.riff, "RIFF", 4) != 0 || std::strncmp(h.wave, "WAVE", 4) != 0 ||
      std::strncmp(h.fmt, "fmt ", 4) != 0) {
    std::fclose(f);
    return -1;  // :32-37
  }
  if (h.block_align == 0) {  // the reference divides by zero here; report as unreadable
    std::fclose(f);
    return -1;
  }
  // :61 sample count from the RIFF size field, not from the data chunk
  const uint32_t num_samples = h.wav_size / h.block_align;
  std::vector<float> samples(num_samples, 0.0f);
  if (h.audio_format == 1) {  // :64-75 PCM16 read starts right after the 36-byte header
    std::vector<int16_t> pcm(num_samples, 0);
    // the reference asks for wav_size bytes into a num_samples*2-byte buffer; the
    // stream ends first for any well-formed file, leaving the tail zero
    size_t want = std::min<size_t>(h.wav_size, size_t(num_samples) * sizeof(int16_t));
    got = std::fread(pcm.data(), 1, want, f);
    for (uint32_t i = 0; i < num_samples; ++i) {
      samples[i] = static_cast<float>(pcm[i]) / static_cast<float>(INT16_MAX);
    }
  } else {  // :76-80 raw float copy
    size_t want = std::min<size_t>(h.wav_size, size_t(num_samples) * sizeof(float));
    got = std::fread(samples.data(), 1, want, f);
  }
  std::fclose(f);
  const long n = static_cast<long>(num_samples);
  for (long i = 0; i < n && i < cap; ++i) out[i] = samples[i];
  return n;
}

// ----------------------------------------------------------------- argmax ---

extern "C" int64_t wto_argmax_last(const float* begin, int64_t n) {
  float best = begin[0];
  int64_t best_i = 0;
  for (int64_t i = 1; i < n; ++i) {
    if (begin[i] >= best) {  // ties move forward: last maximal index wins
      best = begin[i];
      best_i = i;
    }
  }
  return best_i;
}

// --------------------------------------------------------- language table ---

namespace {
// code:name pairs in OpenAI tokenizer order (whisper.cpp:405-508).
const char kLanguages[] =
    "en:english,zh:chinese,de:german,es:spanish,ru:russian,ko:korean,fr:french,ja:japanese,"
    "pt:portuguese,tr:turkish,pl:polish,ca:catalan,nl:dutch,ar:arabic,sv:swedish,it:italian,"
    "id:indonesian,hi:hindi,fi:finnish,vi:vietnamese,he:hebrew,uk:ukrainian,el:greek,ms:malay,"
    "cs:czech,ro:romanian,da:danish,hu:hungarian,ta:tamil,no:norwegian,th:thai,ur:urdu,"
    "hr:croatian,bg:bulgarian,lt:lithuanian,la:latin,mi:maori,ml:malayalam,cy:welsh,sk:slovak,"
    "te:telugu,fa:persian,lv:latvian,bn:bengali,sr:serbian,az:azerbaijani,sl:slovenian,"
    "kn:kannada,et:estonian,mk:macedonian,br:breton,eu:basque,is:icelandic,hy:armenian,"
    "ne:nepali,mn:mongolian,bs:bosnian,kk:kazakh,sq:albanian,sw:swahili,gl:galician,mr:marathi,"
    "pa:punjabi,si:sinhala,km:khmer,sn:shona,yo:yoruba,so:somali,af:afrikaans,oc:occitan,"
    "ka:georgian,be:belarusian,tg:tajik,sd:sindhi,gu:gujarati,am:amharic,yi:yiddish,lo:lao,"
    "uz:uzbek,fo:faroese,ht:haitian creole,ps:pashto,tk:turkmen,nn:nynorsk,mt:maltese,"
    "sa:sanskrit,lb:luxembourgish,my:myanmar,bo:tibetan,tl:tagalog,mg:malagasy,as:assamese,"
    "tt:tatar,haw:hawaiian,ln:lingala,ha:hausa,ba:bashkir,jw:javanese,su:sundanese,"
    "yue:cantonese";

const std::vector<std::string>& lang_codes() {
  static const std::vector<std::string> codes = [] {
    std::vector<std::string> v;
    const char* p = kLanguages;
    while (*p) {
      const char* colon = std::strchr(p, ':');
      v.emplace_back(p, colon);
      const char* comma = std::strchr(colon, ',');
      if (!comma) break;
      p = comma + 1;
    }
    return v;
  }();
  return codes;
}
}  // namespace

extern "C" int wto_language_count(void) { return static_cast<int>(lang_codes().size()); }
extern "C" int wto_language_id(const char* code) {  // :510-515, == size when absent
  const auto& c = lang_codes();
  for (size_t i = 0; i < c.size(); ++i) {
    if (c[i] == code) return static_cast<int>(i);
  }
  return static_cast<int>(c.size());
}
extern "C" const char* wto_lang_code(int id) {
  const auto& c = lang_codes();
  return (id >= 0 && id < static_cast<int>(c.size())) ? c[id].c_str() : "";
}

// ------------------------------------------------------ vocab/filter file ---

struct wto_vocab {
  std::map<int, std::string> id_to_token;
  int n_vocab = 51864;  // whisper.h:69-91 English defaults
  int eot = 50256, sot = 50257, translate = 50358, transcribe = 50359;
  int prev = 50360, solm = 50361, tnot = 50362, beg = 50363;
  int n_mel = 0, n_fft = 0;
  std::vector<float> filters;
};

extern "C" wto_vocab* wto_vocab_open(const char* path, int multilingual) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return nullptr;
  std::vector<char> bytes;
  char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
  std::fclose(f);
  if (bytes.size() < 8 + 4 + 8) return nullptr;
  const char* p = bytes.data() + sizeof(int64_t);  // whisper.cpp:746-747
  const char* const end = bytes.data() + bytes.size();
  auto* v = new wto_vocab;
  p += sizeof(uint32_t);  // magic: read, never checked (:521-529)
  std::memcpy(&v->n_mel, p, 4);
  p += 4;
  std::memcpy(&v->n_fft, p, 4);
  p += 4;
  const size_t nf = size_t(v->n_mel) * size_t(v->n_fft);
  if (v->n_mel <= 0 || v->n_fft <= 0 || p + nf * 4 + 4 > end) {
    delete v;
    return nullptr;
  }
  v->filters.resize(nf);
  std::memcpy(v->filters.data(), p, nf * 4);
  p += nf * 4;
  int32_t n_vocab = 0;
  std::memcpy(&n_vocab, p, 4);
  p += 4;
  v->n_vocab = n_vocab;  // :554
  if (multilingual) {    // :218-226
    v->n_vocab = 51865;
    v->eot++, v->sot++, v->prev++, v->solm++, v->tnot++, v->beg++;
  }
  for (int i = 0; i < n_vocab; ++i) {  // :565-575
    uint32_t len = 0;
    if (p + 4 > end) break;
    std::memcpy(&len, p, 4);
    p += 4;
    if (len > 255 || p + len > end) break;
    // the reference copies into a char[256] and builds a C string: stops at a NUL
    v->id_to_token[i] = std::string(std::string(p, len).c_str());
    p += len;
  }
  const int expected = 51864 + (multilingual ? 1 : 0);  // :577
  for (int i = n_vocab; i < expected; ++i) {            // :578-603
    std::string w;
    if (i > v->beg) {
      w = "<|TT" + std::to_string(i - v->beg) + "|>";
    } else if (i == v->eot) {
      w = "<|endoftranscript|>";
    } else if (i == v->sot) {
      w = "<|startoftranscript_|>";
    } else if (i == v->prev) {
      w = "<|PREV|>";
    } else if (i == v->tnot) {
      w = "<|notimestamps|>";
    } else if (i == v->beg) {
      w = "<|timestampbegin|>";
    } else if (i == v->translate) {
      w = "<|translate|>";
    } else if (i == v->transcribe) {
      w = "<|transcribe|>";
    } else if (i > v->sot && i < v->translate) {
      w = std::string("<|lang-") + wto_lang_code(i - (v->sot + 1)) + "|>";
    } else {
      w = "<|e" + std::to_string(i) + "|>";
    }
    v->id_to_token[i] = w;
  }
  return v;
}

extern "C" void wto_vocab_close(wto_vocab* v) { delete v; }
extern "C" void wto_vocab_info(const wto_vocab* v, int32_t out[9]) {
  const int32_t vals[9] = {v->n_vocab, v->eot,  v->sot,  v->translate, v->transcribe,
                           v->prev,    v->solm, v->tnot, v->beg};
  std::memcpy(out, vals, sizeof(vals));
}
extern "C" void wto_vocab_filters_shape(const wto_vocab* v, int32_t* n_mel, int32_t* n_fft) {
  *n_mel = v->n_mel;
  *n_fft = v->n_fft;
}
extern "C" const float* wto_vocab_filters(const wto_vocab* v) { return v->filters.data(); }
extern "C" int wto_vocab_size(const wto_vocab* v) { return static_cast<int>(v->id_to_token.size()); }
extern "C" int wto_vocab_token(const wto_vocab* v, int id, char* out, int cap) {
  auto it = v->id_to_token.find(id);
  if (it == v->id_to_token.end()) return -1;
  const int len = static_cast<int>(it->second.size());
  std::memcpy(out, it->second.data(), std::min(len, cap));
  return len;
}

extern "C" long wto_decode_text(const wto_vocab* v, const int64_t* ids, int n, int omit_special,
                                char* out, long cap) {
  std::string surface;
  for (int i = 0; i < n; ++i) {  // whisper.cpp:638-649
    const int id = static_cast<int>(ids[i]);
    if (!omit_special || id < v->eot) {
      auto it = v->id_to_token.find(id);
      if (it == v->id_to_token.end()) return -1;  // the reference asserts here
      surface += it->second;
    }
    if (id == v->eot) break;
  }
  const long len = static_cast<long>(surface.size());
  std::memcpy(out, surface.data(), std::min(len, cap));
  return len;
}

extern "C" long wto_remove_extra_spaces(const char* in, char* out, long cap) {
  std::string r;
  bool space = false;
  for (const char* p = in; *p; ++p) {
    if (*p == ' ') {
      if (!space) r += *p;
      space = true;
    } else {
      r += *p;
      space = false;
    }
  }
  const long len = static_cast<long>(r.size());
  std::memcpy(out, r.data(), std::min(len, cap));
  return len;
}

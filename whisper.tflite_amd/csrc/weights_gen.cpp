// Deterministic synthetic Whisper weights -> .wtw file.
//
// Stands in for the reference's offline export step (export/generate_onnx.py:80-163
// downloads OpenAI "tiny" and converts it to a .tflite pair): there is no network
// and no checkpoint in this environment, so both the test container and the GPU
// box materialise bit-identical random-init weights of the right architecture
// from (dims, seed) alone.  Only integer hashing and one double multiply per
// element are used, so the bytes do not depend on libm or the host CPU.
#include "weights_gen.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace wtw {
namespace {

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

inline uint64_t fnv1a(const char* s) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (; *s; ++s) {
    h ^= static_cast<unsigned char>(*s);
    h *= 0x100000001b3ull;
  }
  return h;
}

// Irwin-Hall(4) on 16-bit uniforms: bell-shaped, unit variance after scaling,
// pure integer arithmetic.
inline float unit_normal(uint64_t key, uint64_t i) {
  uint64_t h = splitmix64(key ^ splitmix64(i));
  int64_t s = static_cast<int64_t>((h & 0xFFFF) + ((h >> 16) & 0xFFFF) +
                                   ((h >> 32) & 0xFFFF) + ((h >> 48) & 0xFFFF)) -
              2 * 65535;
  // var = 4 * (65536^2 - 1) / 12
  constexpr double kInvStd = 1.0 / 37837.22704534247;
  return static_cast<float>(static_cast<double>(s) * kInvStd);
}

enum class Init { Normal, OnePlusNormal, Sinusoid };

struct Spec {
  std::string name;
  std::vector<uint32_t> shape;
  Init init;
  double std;
};

void add_attn(std::vector<Spec>& specs, const std::string& p, uint32_t d) {
  const double s = 1.0 / std::sqrt(static_cast<double>(d));
  specs.push_back({p + ".query.weight", {d, d}, Init::Normal, s});
  specs.push_back({p + ".query.bias", {d}, Init::Normal, 0.02});
  specs.push_back({p + ".key.weight", {d, d}, Init::Normal, s});
  specs.push_back({p + ".value.weight", {d, d}, Init::Normal, s});
  specs.push_back({p + ".value.bias", {d}, Init::Normal, 0.02});
  specs.push_back({p + ".out.weight", {d, d}, Init::Normal, s});
  specs.push_back({p + ".out.bias", {d}, Init::Normal, 0.02});
}

void add_ln(std::vector<Spec>& specs, const std::string& p, uint32_t d) {
  specs.push_back({p + ".weight", {d}, Init::OnePlusNormal, 0.1});
  specs.push_back({p + ".bias", {d}, Init::Normal, 0.1});
}

void add_mlp(std::vector<Spec>& specs, const std::string& p, uint32_t d) {
  specs.push_back({p + ".0.weight", {4 * d, d}, Init::Normal, 1.0 / std::sqrt(double(d))});
  specs.push_back({p + ".0.bias", {4 * d}, Init::Normal, 0.02});
  specs.push_back({p + ".2.weight", {d, 4 * d}, Init::Normal, 1.0 / std::sqrt(4.0 * d)});
  specs.push_back({p + ".2.bias", {d}, Init::Normal, 0.02});
}

std::vector<Spec> build_specs(const Dims& c) {
  std::vector<Spec> specs;
  const uint32_t da = c.n_audio_state, dt = c.n_text_state;
  specs.push_back({"encoder.conv1.weight", {da, uint32_t(c.n_mels), 3}, Init::Normal,
                   1.0 / std::sqrt(3.0 * c.n_mels)});
  specs.push_back({"encoder.conv1.bias", {da}, Init::Normal, 0.02});
  specs.push_back({"encoder.conv2.weight", {da, da, 3}, Init::Normal, 1.0 / std::sqrt(3.0 * da)});
  specs.push_back({"encoder.conv2.bias", {da}, Init::Normal, 0.02});
  specs.push_back({"encoder.positional_embedding", {uint32_t(c.n_audio_ctx), da}, Init::Sinusoid, 0});
  for (int i = 0; i < c.n_audio_layer; ++i) {
    const std::string b = "encoder.blocks." + std::to_string(i);
    add_ln(specs, b + ".attn_ln", da);
    add_attn(specs, b + ".attn", da);
    add_ln(specs, b + ".mlp_ln", da);
    add_mlp(specs, b + ".mlp", da);
  }
  add_ln(specs, "encoder.ln_post", da);
  specs.push_back({"decoder.token_embedding.weight", {uint32_t(c.n_vocab), dt}, Init::Normal, 0.02});
  specs.push_back({"decoder.positional_embedding", {uint32_t(c.n_text_ctx), dt}, Init::Normal, 0.02});
  for (int i = 0; i < c.n_text_layer; ++i) {
    const std::string b = "decoder.blocks." + std::to_string(i);
    add_ln(specs, b + ".attn_ln", dt);
    add_attn(specs, b + ".attn", dt);
    add_ln(specs, b + ".cross_attn_ln", dt);
    add_attn(specs, b + ".cross_attn", dt);
    add_ln(specs, b + ".mlp_ln", dt);
    add_mlp(specs, b + ".mlp", dt);
  }
  add_ln(specs, "decoder.ln", dt);
  return specs;
}

uint64_t numel(const std::vector<uint32_t>& s) {
  uint64_t n = 1;
  for (uint32_t v : s) n *= v;
  return n;
}

// OpenAI whisper/model.py sinusoids(): [sin | cos] halves over log-spaced timescales.
// Evaluated in double; the table is written to the file, so readers never recompute it.
void fill_sinusoid(float* out, uint32_t length, uint32_t channels) {
  const uint32_t half = channels / 2;
  const double inc = std::log(10000.0) / (half > 1 ? (half - 1) : 1);
  for (uint32_t t = 0; t < length; ++t) {
    for (uint32_t j = 0; j < half; ++j) {
      const double st = double(t) * std::exp(-inc * double(j));
      out[size_t(t) * channels + j] = static_cast<float>(std::sin(st));
      out[size_t(t) * channels + half + j] = static_cast<float>(std::cos(st));
    }
  }
}

}  // namespace

bool dims_by_name(const char* name, Dims* out) {
  const std::string n(name ? name : "");
  if (n == "tiny") {
    *out = Dims{80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4};
  } else if (n == "tiny.en") {
    *out = Dims{80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4};
  } else if (n == "base") {
    *out = Dims{80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6};
  } else if (n == "micro") {  // test-sized: same graph, seconds on a CPU
    *out = Dims{80, 100, 128, 2, 2, 1024, 64, 128, 2, 2};
  } else {
    return false;
  }
  return true;
}

std::vector<NamedTensor> tensor_specs(const Dims& dims) {
  std::vector<NamedTensor> out;
  for (const Spec& s : build_specs(dims)) out.push_back(NamedTensor{s.name, s.shape, {}});
  return out;
}

int write_tensors(const char* path, const Dims& dims, const std::vector<NamedTensor>& tensors, uint64_t seed,
                  std::string* err) {
  for (const NamedTensor& t : tensors) {
    if (t.shape.empty() || t.shape.size() > 4 || t.name.size() >= sizeof(WtwTensor::name) || numel(t.shape) != t.data.size()) {
      if (err) *err = "bad tensor " + t.name;
      return 1;
    }
  }
  WtwHeader hdr;
  std::memset(&hdr, 0, sizeof(hdr));
  hdr.magic = kMagic;
  hdr.version = kVersion;
  hdr.n_tensors = static_cast<uint32_t>(tensors.size());
  hdr.table_offset = sizeof(WtwHeader);
  hdr.dims = dims;
  hdr.seed = seed;
  std::vector<WtwTensor> table(tensors.size());
  uint64_t off = sizeof(WtwHeader) + sizeof(WtwTensor) * tensors.size();
  off = (off + kAlign - 1) / kAlign * kAlign;
  hdr.payload_offset = off;
  for (size_t i = 0; i < tensors.size(); ++i) {
    WtwTensor& t = table[i];
    std::memset(&t, 0, sizeof(t));
    std::snprintf(t.name, sizeof(t.name), "%s", tensors[i].name.c_str());
    t.dtype = 0;
    t.ndim = static_cast<uint32_t>(tensors[i].shape.size());
    for (size_t k = 0; k < tensors[i].shape.size(); ++k) t.shape[k] = tensors[i].shape[k];
    t.offset = off;
    t.nbytes = tensors[i].data.size() * sizeof(float);
    off += (t.nbytes + kAlign - 1) / kAlign * kAlign;
  }
  hdr.file_bytes = off;

  // Written next to the target under a name of its own and renamed into place: a reader (another rank converting
  // the same .tflite pair, a later wt_engine_create) sees either no file or a complete one, never a short write.
  // The temporary is created exclusively and without following links (O_EXCL | O_NOFOLLOW, mode 0600): in a shared
  // directory nobody can pre-plant the name — as a file or as a symlink to one of the user's own files.
  const std::string tmp = std::string(path) + ".tmp." + std::to_string(static_cast<long>(::getpid()));
  ::unlink(tmp.c_str());  // a leftover of a killed run of this very pid
  const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  FILE* f = fd >= 0 ? ::fdopen(fd, "wb") : nullptr;
  if (!f) {
    if (fd >= 0) ::close(fd);
    if (err) *err = std::string("cannot open for writing: ") + tmp;
    return 2;
  }
  bool ok = std::fwrite(&hdr, sizeof(hdr), 1, f) == 1;
  ok = ok && (table.empty() || std::fwrite(table.data(), sizeof(WtwTensor), table.size(), f) == table.size());
  uint64_t pos = sizeof(WtwHeader) + sizeof(WtwTensor) * tensors.size();
  static const char zeros[kAlign] = {0};
  auto pad_to = [&](uint64_t target) {
    while (ok && pos < target) {
      const uint64_t n = std::min<uint64_t>(kAlign, target - pos);
      ok = ok && std::fwrite(zeros, 1, n, f) == n;
      pos += n;
    }
  };
  for (size_t i = 0; ok && i < tensors.size(); ++i) {
    pad_to(table[i].offset);
    const size_t n = tensors[i].data.size();
    ok = ok && std::fwrite(tensors[i].data.data(), sizeof(float), n, f) == n;
    pos += n * sizeof(float);
  }
  pad_to(hdr.file_bytes);
  ok = ok && std::fflush(f) == 0 && ::fsync(::fileno(f)) == 0;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) {
    ::unlink(tmp.c_str());
    if (err) *err = std::string("short write: ") + tmp;
    return 2;
  }
  if (::rename(tmp.c_str(), path) != 0) {  // two writers racing for the same name both succeed: same bytes
    ::unlink(tmp.c_str());
    if (err) *err = std::string("cannot rename into place: ") + path;
    return 2;
  }
  return 0;
}

int write_synthetic(const char* path, const Dims& dims, uint64_t seed, std::string* err) {
  if (dims.n_audio_state % dims.n_audio_head != 0 || dims.n_text_state % dims.n_text_head != 0 ||
      dims.n_audio_state != dims.n_text_state) {
    if (err) *err = "inconsistent dims";
    return 1;
  }
  const std::vector<Spec> specs = build_specs(dims);
  std::vector<NamedTensor> tensors;
  tensors.reserve(specs.size());
  for (const Spec& sp : specs) {
    NamedTensor t{sp.name, sp.shape, {}};
    const uint64_t n = numel(sp.shape);
    t.data.resize(n);
    if (sp.init == Init::Sinusoid) {
      fill_sinusoid(t.data.data(), sp.shape[0], sp.shape[1]);
    } else {
      const uint64_t key = splitmix64(seed ^ fnv1a(sp.name.c_str()));
      const float sd = static_cast<float>(sp.std);
      const float base = sp.init == Init::OnePlusNormal ? 1.0f : 0.0f;
      for (uint64_t e = 0; e < n; ++e) t.data[e] = base + sd * unit_normal(key, e);
    }
    tensors.push_back(std::move(t));
  }
  return write_tensors(path, dims, tensors, seed, err);
}

}  // namespace wtw

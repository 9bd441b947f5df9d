#include "engine.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "kernels.h"

namespace wt {

namespace {
void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    throw Error(5, std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
  }
}
#define HIPCHK(x) hip_check((x), #x)

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
}  // namespace

// ------------------------------------------------------------ weights ---

// W [N][K] -> two fp16 planes in MFMA-fragment order [ceil(N/32)][K/16][plane][64 lanes][8] (rows past N zero):
// lane (l & 31, l >> 5) of tile t, k-step s holds W[32t + (l & 31)][16s + 8(l >> 5) .. +7], so one wave-instruction
// of the decoder GEMM reads 1 KiB contiguous.  The planes are hi = fp16(w * scale), lo = fp16(w * scale - hi) with
// scale the power of two that puts max |W| into (8192, 16384]: 22 significand bits, same bytes as fp32.
std::vector<unsigned short> tile_weights_f16(const float* W, int N, int K, float* scale) {
  float mx = 0.0f;
  for (size_t i = 0; i < size_t(N) * K; ++i) mx = std::max(mx, std::fabs(W[i]));
  const float sc = f16_scale_for(mx);
  *scale = sc;
  const int n_tiles = (N + 31) / 32, steps = K / 16;
  std::vector<unsigned short> out(size_t(n_tiles) * steps * 1024, 0);
  for (int t = 0; t < n_tiles; ++t)
    for (int s = 0; s < steps; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = t * 32 + (lane & 31);
        if (n >= N) continue;
        const float* src = W + size_t(n) * K + 16 * s + 8 * (lane >> 5);
        unsigned short* hi = out.data() + (size_t(t) * steps + s) * 1024 + lane * 8;
        unsigned short* lo = hi + 512;
        for (int e = 0; e < 8; ++e) {
          const float v = src[e] * sc;
          const _Float16 h = static_cast<_Float16>(v);
          const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
          std::memcpy(hi + e, &h, 2);
          std::memcpy(lo + e, &l, 2);
        }
      }
  return out;
}

namespace {
inline unsigned short bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return static_cast<unsigned short>((u >> 16) | 0x40u);  // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return static_cast<unsigned short>(u >> 16);
}
}  // namespace

// bf16 storage mode: one bf16 plane in the decoder GEMM's fragment order [ceil(N/32)][K/16][64 lanes][8]
std::vector<unsigned short> tile_weights_bf16(const float* W, int N, int K) {
  const int n_tiles = (N + 31) / 32, steps = K / 16;
  std::vector<unsigned short> out(size_t(n_tiles) * steps * 512, 0);
  for (int t = 0; t < n_tiles; ++t)
    for (int s = 0; s < steps; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = t * 32 + (lane & 31);
        if (n >= N) continue;
        const float* src = W + size_t(n) * K + 16 * s + 8 * (lane >> 5);
        unsigned short* dst = out.data() + (size_t(t) * steps + s) * 512 + lane * 8;
        for (int e = 0; e < 8; ++e) dst[e] = bf16_rne(src[e]);
      }
  return out;
}

// bf16 storage mode: W [N][K] fp32 -> bf16 [N][Kpad], zero filled
std::vector<unsigned short> round_weights_bf16(const float* W, int N, int K, int Kpad) {
  std::vector<unsigned short> out(size_t(N) * Kpad, 0);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) out[size_t(n) * Kpad + k] = bf16_rne(W[size_t(n) * K + k]);
  return out;
}

// Wq [d][d] (row = output) -> [head][d / 4][64 outputs of the head][4 k]: thread (output j, k-quarter) of
// cross_attention_step reads 16 contiguous bytes per step and a wavefront 1 KiB
std::vector<float> cross_q_layout(const float* Wq, int d) {
  const int heads = d / 64, c4 = d / 4;
  std::vector<float> out(size_t(d) * d);
  for (int h = 0; h < heads; ++h)
    for (int c = 0; c < c4; ++c)
      for (int j = 0; j < 64; ++j)
        for (int e = 0; e < 4; ++e)
          out[((size_t(h) * c4 + c) * 64 + j) * 4 + e] = Wq[size_t(h * 64 + j) * d + 4 * c + e];
  return out;
}

Engine::PlaneW Engine::upload_planes(const float* W, int N, int K, int Kpad, float scale) {
  const std::vector<unsigned short> planes = split_weight_planes(W, N, K, Kpad, scale);
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, planes.size() * sizeof(unsigned short) + 256));
  allocations_.push_back(p);
  HIPCHK(hipMemcpy(p, planes.data(), planes.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  return PlaneW{static_cast<const unsigned short*>(p)};
}

TiledW Engine::upload_tiled(const float* W, int N, int K) {
  TiledW t;
  const std::vector<unsigned short> planes = tile_weights_f16(W, N, K, &t.scale);
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, std::max<size_t>(planes.size(), 1) * sizeof(unsigned short)));
  allocations_.push_back(p);
  HIPCHK(hipMemcpy(p, planes.data(), planes.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  t.w = static_cast<const unsigned short*>(p);
  return t;
}

float* Engine::upload(const std::vector<float>& host) {
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, std::max<size_t>(host.size(), 1) * sizeof(float)));
  allocations_.push_back(p);
  if (!host.empty()) HIPCHK(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return static_cast<float*>(p);
}

float* Engine::dalloc(size_t n_floats) {
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, std::max<size_t>(n_floats, 1) * sizeof(float)));
  return static_cast<float*>(p);
}

const float* Engine::dev(const std::string& name) const {
  auto it = tensors_.find(name);
  if (it == tensors_.end()) throw Error(3, "weight file: missing tensor " + name);
  return it->second;
}

namespace {
// A .wtw file mapped read-only: header checked, tensor table parsed into name -> (fp32 payload, element count).
struct WtwFile {
  void* map = MAP_FAILED;
  size_t bytes = 0;
  wtw::WtwHeader hdr{};
  std::map<std::string, std::pair<const float*, size_t>> host;

  explicit WtwFile(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(2, "Failed to open file: " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || size_t(st.st_size) < sizeof(wtw::WtwHeader)) {
      ::close(fd);
      throw Error(3, "weight file too small: " + path);
    }
    bytes = size_t(st.st_size);
    map = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) throw Error(2, "Failed to mmap file: " + path);
    try {
      const char* base = static_cast<const char*>(map);
      std::memcpy(&hdr, base, sizeof(hdr));
      if (hdr.magic != wtw::kMagic || hdr.version != wtw::kVersion || hdr.n_tensors > (1u << 20) ||
          size_t(hdr.table_offset) + size_t(hdr.n_tensors) * sizeof(wtw::WtwTensor) > bytes) {
        throw Error(3, "not a .wtw weight file: " + path);
      }
      const uint64_t file_bytes = uint64_t(bytes);
      for (uint32_t i = 0; i < hdr.n_tensors; ++i) {
        wtw::WtwTensor t;
        std::memcpy(&t, base + hdr.table_offset + size_t(i) * sizeof(t), sizeof(t));
        // overflow-safe range check; fp32 payloads are read in place, so they must be 4-byte aligned
        if (t.dtype != 0 || t.offset > file_bytes || t.nbytes > file_bytes - t.offset || t.offset % 4 != 0 || t.nbytes % 4 != 0) {
          throw Error(kErrFormat, "corrupt tensor table in " + path);
        }
        t.name[sizeof(t.name) - 1] = 0;
        host[t.name] = {reinterpret_cast<const float*>(base + t.offset), t.nbytes / sizeof(float)};
      }
    } catch (...) {
      munmap(map, bytes);
      throw;
    }
  }
  ~WtwFile() {
    if (map != MAP_FAILED) munmap(map, bytes);
  }
  WtwFile(const WtwFile&) = delete;
  WtwFile& operator=(const WtwFile&) = delete;
  const float* get(const std::string& n, size_t expect) const {
    auto it = host.find(n);
    if (it == host.end()) throw Error(3, "weight file: missing tensor " + n);
    if (it->second.second != expect) throw Error(3, "weight file: bad shape for " + n);
    return it->second.first;
  }
};
}  // namespace

void check_wtw_file(const std::string& path) {
  const WtwFile file(path);  // throws kErrIo / kErrFormat
  if (file.hdr.file_bytes != 0 && file.hdr.file_bytes != file.bytes) {
    throw Error(kErrFormat, "truncated or padded .wtw weight file (header says " + std::to_string(file.hdr.file_bytes) +
                                " bytes, file has " + std::to_string(file.bytes) + "): " + path);
  }
}

// The fused query matrix of the absorbed cross-attention, folded once, in double: A_h = c0 Wk_h^T Wq_h stacked over the
// heads ([heads * d][d]: row (h, c) gives column c of head h's d-wide query) and a_h = c0 Wk_h^T bq_h, with
// c0 = d_head^-1/2 log2 e (the kernel's softmax is an exp2).
static void absorbed_query_matrix(const float* wq, const float* bq, const float* wk, int heads, int d, std::vector<float>* A,
                                  std::vector<float>* av) {
  const double c0 = 0.125 * 1.44269504088896340736;
  A->assign(size_t(heads) * d * d, 0.0f);
  av->assign(size_t(heads) * d, 0.0f);
  std::vector<double> acc(d);
  for (int h = 0; h < heads; ++h) {
    for (int cc = 0; cc < d; ++cc) {
      std::fill(acc.begin(), acc.end(), 0.0);
      double ab = 0.0;
      for (int i = 0; i < 64; ++i) {
        const double kv = wk[size_t(h * 64 + i) * d + cc];
        const float* qrow = wq + size_t(h * 64 + i) * d;
        for (int j = 0; j < d; ++j) acc[j] += kv * qrow[j];
        ab += kv * bq[h * 64 + i];
      }
      float* arow = A->data() + (size_t(h) * d + cc) * d;
      for (int j = 0; j < d; ++j) arow[j] = float(c0 * acc[j]);
      (*av)[size_t(h) * d + cc] = float(c0 * ab);
    }
  }
}

void Engine::upload_weights(const std::string& path) {
  // Replaces Atom::Atom (whisper.cpp:261-271): instead of mmapping a .tflite FlatBuffer and
  // building an interpreter, the flat .wtw payload is mapped, re-laid-out for the kernels
  // and copied to HBM once.
  const WtwFile file(path);
  weights_path_ = path;
  const wtw::WtwHeader& hdr = file.hdr;
  dims_ = hdr.dims;
  const wtw::Dims& c = dims_;
  // Everything the kernels index with comes from this header: reject what they do not support here, as a
  // format error, instead of faulting (or dividing by zero) later.
  const auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  const int dm = c.n_audio_state;
  if (!(dm == 128 || dm == 384 || dm == 512) || c.n_text_state != dm) {
    throw Error(kErrFormat, "unsupported model dims: d_model must be 128, 384 or 512 (encoder == decoder)");
  }
  if (!in(c.n_audio_head, 1, 64) || !in(c.n_text_head, 1, 64) || dm != 64 * c.n_audio_head || dm != 64 * c.n_text_head) {
    throw Error(kErrFormat, "unsupported model dims: heads must be d_model / 64");
  }
  if (!in(c.n_mels, 1, 128) || !in(c.n_audio_ctx, 32, 4096) || !in(c.n_text_ctx, 32, 4096) ||
      !in(c.n_audio_layer, 1, 64) || !in(c.n_text_layer, 1, 64) || !in(c.n_vocab, 1, 1 << 20)) {
    throw Error(kErrFormat, "unsupported model dims: n_mels 1..128, n_audio_ctx / n_text_ctx 32..4096, layers 1..64, n_vocab 1..2^20");
  }
  auto H = [&](const std::string& n, size_t expect) -> const float* { return file.get(n, expect); };
  auto up = [&](const std::string& n, size_t expect) -> const float* {
    const float* p = H(n, expect);
    const float* d = upload(std::vector<float>(p, p + expect));
    tensors_[n] = d;
    return d;
  };

  const int d = c.n_audio_state, nm = c.n_mels;
  // conv1 [d][n_mels][3] -> [d][kpad], k = kk * n_mels + ci  (implicit-GEMM order: the three
  // taps of output frame t are three consecutive rows of the time-major padded input)
  conv1_kpad = int(round_up(size_t(3) * nm, 32));
  {
    const float* w = H("encoder.conv1.weight", size_t(d) * nm * 3);
    std::vector<float> r(size_t(d) * conv1_kpad, 0.0f);
    for (int co = 0; co < d; ++co)
      for (int ci = 0; ci < nm; ++ci)
        for (int kk = 0; kk < 3; ++kk)
          r[size_t(co) * conv1_kpad + kk * nm + ci] = w[(size_t(co) * nm + ci) * 3 + kk];
    conv1_w = upload(r);
    conv1_b = up("encoder.conv1.bias", d);
  }
  {
    const float* w = H("encoder.conv2.weight", size_t(d) * d * 3);
    std::vector<float> r(size_t(d) * 3 * d);
    for (int co = 0; co < d; ++co)
      for (int ci = 0; ci < d; ++ci)
        for (int kk = 0; kk < 3; ++kk)
          r[size_t(co) * 3 * d + kk * d + ci] = w[(size_t(co) * d + ci) * 3 + kk];
    conv2_w = upload(r);
    conv2_b = up("encoder.conv2.bias", d);
  }
  enc_pos = up("encoder.positional_embedding", size_t(c.n_audio_ctx) * d);

  auto fused_qkv = [&](const std::string& p, std::vector<float>* w, std::vector<float>* b) {
    const size_t dd = size_t(d) * d;
    w->assign(3 * dd, 0.0f);
    b->assign(3 * size_t(d), 0.0f);
    std::memcpy(w->data(), H(p + ".query.weight", dd), dd * 4);
    std::memcpy(w->data() + dd, H(p + ".key.weight", dd), dd * 4);
    std::memcpy(w->data() + 2 * dd, H(p + ".value.weight", dd), dd * 4);
    std::memcpy(b->data(), H(p + ".query.bias", d), size_t(d) * 4);
    std::memcpy(b->data() + 2 * size_t(d), H(p + ".value.bias", d), size_t(d) * 4);  // key has no bias
  };
  enc_blocks_.resize(c.n_audio_layer);
  for (int l = 0; l < c.n_audio_layer; ++l) {
    const std::string blk = "encoder.blocks." + std::to_string(l);
    BlockWeights& bw = enc_blocks_[l];
    bw.attn_ln_g = up(blk + ".attn_ln.weight", d);
    bw.attn_ln_b = up(blk + ".attn_ln.bias", d);
    std::vector<float> wqkv, bqkv;
    fused_qkv(blk + ".attn", &wqkv, &bqkv);
    bw.attn.wqkv = upload(wqkv);
    bw.attn.bqkv = upload(bqkv);
    bw.attn.wo = up(blk + ".attn.out.weight", size_t(d) * d);
    bw.attn.bo = up(blk + ".attn.out.bias", d);
    bw.mlp_ln_g = up(blk + ".mlp_ln.weight", d);
    bw.mlp_ln_b = up(blk + ".mlp_ln.bias", d);
    bw.w1 = up(blk + ".mlp.0.weight", size_t(4) * d * d);
    bw.b1 = up(blk + ".mlp.0.bias", size_t(4) * d);
    bw.w2 = up(blk + ".mlp.2.weight", size_t(4) * d * d);
    bw.b2 = up(blk + ".mlp.2.bias", d);
  }
  enc_ln_post_g = up("encoder.ln_post.weight", d);
  enc_ln_post_b = up("encoder.ln_post.bias", d);

  tok_emb = up("decoder.token_embedding.weight", size_t(c.n_vocab) * d);  // row lookup
  tok_emb_tiled = upload_tiled(H("decoder.token_embedding.weight", size_t(c.n_vocab) * d), c.n_vocab, d);  // logits GEMM
  dec_pos = up("decoder.positional_embedding", size_t(c.n_text_ctx) * d);
  dec_blocks_.resize(c.n_text_layer);
  const size_t dd = size_t(d) * d;
  std::vector<float> ckv_w(size_t(c.n_text_layer) * 2 * dd), ckv_b(size_t(c.n_text_layer) * 2 * d, 0.0f);
  for (int l = 0; l < c.n_text_layer; ++l) {
    const std::string blk = "decoder.blocks." + std::to_string(l);
    DecBlockWeights& bw = dec_blocks_[l];
    bw.attn_ln_g = up(blk + ".attn_ln.weight", d);
    bw.attn_ln_b = up(blk + ".attn_ln.bias", d);
    std::vector<float> wqkv, bqkv;
    fused_qkv(blk + ".attn", &wqkv, &bqkv);
    bw.wqkv = upload_tiled(wqkv.data(), 3 * d, d);  // decoder Linears: fp16 planes in MFMA-fragment order
    bw.bqkv = upload(bqkv);
    bw.wo = upload_tiled(H(blk + ".attn.out.weight", dd), d, d);
    bw.bo = up(blk + ".attn.out.bias", d);
    bw.cross_ln_g = up(blk + ".cross_attn_ln.weight", d);
    bw.cross_ln_b = up(blk + ".cross_attn_ln.bias", d);
    bw.cross_wq_t = upload(cross_q_layout(H(blk + ".cross_attn.query.weight", dd), d));
    bw.cross_bq = up(blk + ".cross_attn.query.bias", d);
    bw.cross_wo = upload_tiled(H(blk + ".cross_attn.out.weight", dd), d, d);
    bw.cross_bo = up(blk + ".cross_attn.out.bias", d);
    {
      // Absorbed cross-attention (k_cross_absorbed.hip): scores q_h . (Wk_h e) = (Wk_h^T q_h) . e and contexts
      // sum_j p_j (Wv_h e_j + bv_h) = Wv_h (sum_j p_j e_j) + bv_h, so the decoder works on the encoder output e itself.
      // Wv_h and bv_h are applied to each head's combined context (cross_absorbed_combine).
      const int Hh = c.n_text_head;
      const float* wv = H(blk + ".cross_attn.value.weight", dd);
      const float* bv = H(blk + ".cross_attn.value.bias", d);
      std::vector<float> A, av;
      absorbed_query_matrix(H(blk + ".cross_attn.query.weight", dd), H(blk + ".cross_attn.query.bias", d),
                            H(blk + ".cross_attn.key.weight", dd), Hh, d, &A, &av);
      bw.wq_abs = upload_tiled(A.data(), Hh * d, d);
      bw.bq_abs = upload(av);
      bw.cross_wv_t = upload(cross_q_layout(wv, d));
      bw.cross_bv = upload(std::vector<float>(bv, bv + d));
    }
    // all layers' cross K/V projections act on the same encoder output: one GEMM
    std::memcpy(ckv_w.data() + (size_t(l) * 2 + 0) * dd, H(blk + ".cross_attn.key.weight", dd), dd * 4);
    std::memcpy(ckv_w.data() + (size_t(l) * 2 + 1) * dd, H(blk + ".cross_attn.value.weight", dd), dd * 4);
    std::memcpy(ckv_b.data() + (size_t(l) * 2 + 1) * d, H(blk + ".cross_attn.value.bias", d), size_t(d) * 4);
    bw.mlp_ln_g = up(blk + ".mlp_ln.weight", d);
    bw.mlp_ln_b = up(blk + ".mlp_ln.bias", d);
    bw.w1 = upload_tiled(H(blk + ".mlp.0.weight", 4 * dd), 4 * d, d);
    bw.b1 = up(blk + ".mlp.0.bias", size_t(4) * d);
    bw.w2 = upload_tiled(H(blk + ".mlp.2.weight", 4 * dd), d, 4 * d);
    bw.b2 = up(blk + ".mlp.2.bias", d);
  }
  cross_kv_w = upload(ckv_w);
  cross_kv_b = upload(ckv_b);
  dec_ln_g = up("decoder.ln.weight", d);
  dec_ln_b = up("decoder.ln.bias", d);
  // ---- operand bounds -> fp16 plane scales (engine.h, GemmScale) ----
  // Next to every bound a TYPICAL magnitude of the same operand is propagated (rms under unit-variance inputs:
  // LayerNorm sqrt(mean g^2 + mean b^2), Linear sqrt(mean_n sum_k W^2) * typical input).  The two-plane fp16
  // form keeps 22 significand bits only while the second plane stays a normal fp16 number, i.e. while
  // |x| * scale >= 2^-3; with scale = 2^14 / bound that needs typical / bound >= 2^-17.  A contraction whose
  // operand has bound / typical above kF16Slack (2^12: a factor 32 of margin for elements below the typical
  // magnitude) is given the full-range bf16 three-plane kernels instead — decided here, once, from the
  // weights alone (LayerNorm gains with outliers, heavy-tailed rows: tests/test_gpu_boundary.py).
  {
    auto maxabs = [](const float* w, size_t n) {
      float m = 0.0f;
      for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(w[i]));
      return m;
    };
    auto vmax = [](const std::vector<float>& v) {
      float m = 0.0f;
      for (float x : v) m = std::max(m, x);
      return m;
    };
    struct Op {  // one contraction operand: per-channel bound, typical magnitude
      std::vector<float> bound;
      float typ;
    };
    auto ln_op = [&](const std::string& gname, const std::string& bname) {
      const float* gg = H(gname, d);
      const float* bb = H(bname, d);
      Op o{std::vector<float>(d), 0.0f};
      const float r = std::sqrt(float(d - 1));
      double acc = 0.0;
      for (int i = 0; i < d; ++i) {
        o.bound[i] = std::fabs(gg[i]) * r + std::fabs(bb[i]);
        acc += double(gg[i]) * gg[i] + double(bb[i]) * bb[i];
      }
      o.typ = float(std::sqrt(acc / d));
      return o;
    };
    // out[n] = sum_k |W[n][k]| in[k % in.size()] + |bias[n]|   (conv taps repeat the input bound)
    auto linear_op = [&](const float* w, const float* bias, int N, int K, const Op& in) {
      Op o{std::vector<float>(N), 0.0f};
      double sq = 0.0;
      for (int n = 0; n < N; ++n) {
        double acc = bias ? std::fabs(bias[n]) : 0.0;
        for (int k = 0; k < K; ++k) {
          const double wv = w[size_t(n) * K + k];
          acc += std::fabs(wv) * in.bound[size_t(k) % in.bound.size()];
          sq += wv * wv;
        }
        o.bound[n] = float(acc);
      }
      o.typ = float(std::sqrt(sq / N)) * in.typ;
      return o;
    };
    auto gelu_op = [](Op o) {
      for (float& x : o.bound) x = std::max(x, 0.17f);  // gelu(x) in [-0.17, max(x, 0)]
      o.typ *= 0.5f;
      return o;
    };
    n_f16_fallbacks_ = 0;
    // slack of an operand in bits: log2(kF16Slack * typical / bound) — negative: the contraction leaves the plane kernels
    f16_min_slack_bits_ = 1.0e9f;
    auto slack_ok = [&](const Op& o) {
      const float b = vmax(o.bound);
      if (b > 0.0f && o.typ > 0.0f) f16_min_slack_bits_ = std::min(f16_min_slack_bits_, std::log2(kF16Slack * o.typ / b));
      return !(b > kF16Slack * o.typ);
    };
    auto scale_of = [&](const Op& in, const float* w, size_t n) {
      GemmScale g{f16_scale_for(vmax(in.bound)), f16_scale_for(maxabs(w, n)), slack_ok(in)};
      if (!g.f16_ok) ++n_f16_fallbacks_;
      return g;
    };
    // conv1: host copy in the kernel's [co][kpad] order is gone; bounds use the original [co][ci][3] tensor,
    // which has the same absolute values
    const Op mel_op{std::vector<float>(1, kMelBound), 1.0f};
    const float* c1w = H("encoder.conv1.weight", size_t(d) * nm * 3);
    sc_conv1_ = scale_of(mel_op, c1w, size_t(d) * nm * 3);
    const Op h1 = gelu_op(linear_op(c1w, H("encoder.conv1.bias", d), d, nm * 3, mel_op));
    const float* c2w = H("encoder.conv2.weight", size_t(d) * d * 3);
    sc_conv2_ = scale_of(h1, c2w, size_t(d) * d * 3);
    sc_layers_.assign(c.n_audio_layer, EncLayerScales{});
    const size_t dd2 = size_t(d) * d;
    for (int l = 0; l < c.n_audio_layer; ++l) {
      const std::string blk = "encoder.blocks." + std::to_string(l);
      EncLayerScales& sl = sc_layers_[l];
      const Op ln1 = ln_op(blk + ".attn_ln.weight", blk + ".attn_ln.bias");
      const float* wq = H(blk + ".attn.query.weight", dd2);
      const float* wk = H(blk + ".attn.key.weight", dd2);
      const float* wv = H(blk + ".attn.value.weight", dd2);
      const float wmax = std::max(maxabs(wq, dd2), std::max(maxabs(wk, dd2), maxabs(wv, dd2)));
      sl.qkv = GemmScale{f16_scale_for(vmax(ln1.bound)), f16_scale_for(wmax), slack_ok(ln1)};
      if (!sl.qkv.f16_ok) ++n_f16_fallbacks_;
      const Op qo = linear_op(wq, H(blk + ".attn.query.bias", d), d, d, ln1);
      const Op ko = linear_op(wk, nullptr, d, d, ln1);
      const Op vo = linear_op(wv, H(blk + ".attn.value.bias", d), d, d, ln1);
      sl.q = f16_scale_for(vmax(qo.bound) * 0.125f * 1.44269504f);  // the kernel splits q * d_head^-1/2 * log2(e)
      sl.k = f16_scale_for(vmax(ko.bound));
      sl.v = f16_scale_for(vmax(vo.bound));
      sl.attn_f16_ok = slack_ok(qo) && slack_ok(ko) && slack_ok(vo);
      if (!sl.attn_f16_ok) ++n_f16_fallbacks_;
      sl.out = scale_of(vo, H(blk + ".attn.out.weight", dd2), dd2);  // a convex combination of V rows
      const Op ln2 = ln_op(blk + ".mlp_ln.weight", blk + ".mlp_ln.bias");
      const float* w1 = H(blk + ".mlp.0.weight", 4 * dd2);
      sl.fc1 = scale_of(ln2, w1, 4 * dd2);
      const Op hb = gelu_op(linear_op(w1, H(blk + ".mlp.0.bias", size_t(4) * d), 4 * d, d, ln2));
      sl.fc2 = scale_of(hb, H(blk + ".mlp.2.weight", 4 * dd2), 4 * dd2);
    }
    const Op lnp = ln_op("encoder.ln_post.weight", "encoder.ln_post.bias");
    sc_cross_kv_ = GemmScale{f16_scale_for(vmax(lnp.bound)), f16_scale_for(maxabs(ckv_w.data(), ckv_w.size())), slack_ok(lnp)};
    if (!sc_cross_kv_.f16_ok) ++n_f16_fallbacks_;
    load_ok_.clear();
    for (bool* f : ok_flags()) load_ok_.push_back(*f);
  }
  // ---- encoder weights as fp16 planes (k_gemm_planes.hip): split once, here ----
  {
    conv1_kpad_p_ = int(round_up(size_t(3) * nm, 32));
    std::vector<float> r(size_t(d) * 3 * nm);
    const float* w = H("encoder.conv1.weight", size_t(d) * nm * 3);
    for (int co = 0; co < d; ++co)
      for (int ci = 0; ci < nm; ++ci)
        for (int kk = 0; kk < 3; ++kk) r[size_t(co) * 3 * nm + kk * nm + ci] = w[(size_t(co) * nm + ci) * 3 + kk];
    conv1_p_ = upload_planes(r.data(), d, 3 * nm, conv1_kpad_p_, sc_conv1_.w);
    std::vector<float> r2(size_t(d) * 3 * d);
    const float* w2 = H("encoder.conv2.weight", size_t(d) * d * 3);
    for (int co = 0; co < d; ++co)
      for (int ci = 0; ci < d; ++ci)
        for (int kk = 0; kk < 3; ++kk) r2[size_t(co) * 3 * d + kk * d + ci] = w2[(size_t(co) * d + ci) * 3 + kk];
    conv2_p_ = upload_planes(r2.data(), d, 3 * d, 3 * d, sc_conv2_.w);
    enc_planes_.resize(c.n_audio_layer);
    const size_t dd3 = size_t(d) * d;
    for (int l = 0; l < c.n_audio_layer; ++l) {
      const std::string blk = "encoder.blocks." + std::to_string(l);
      std::vector<float> wqkv, bqkv;
      fused_qkv(blk + ".attn", &wqkv, &bqkv);
      enc_planes_[l].qkv = upload_planes(wqkv.data(), 3 * d, d, d, sc_layers_[l].qkv.w);
      enc_planes_[l].out = upload_planes(H(blk + ".attn.out.weight", dd3), d, d, d, sc_layers_[l].out.w);
      enc_planes_[l].fc1 = upload_planes(H(blk + ".mlp.0.weight", 4 * dd3), 4 * d, d, d, sc_layers_[l].fc1.w);
      enc_planes_[l].fc2 = upload_planes(H(blk + ".mlp.2.weight", 4 * dd3), d, 4 * d, 4 * d, sc_layers_[l].fc2.w);
    }
    cross_kv_p_ = upload_planes(ckv_w.data(), c.n_text_layer * 2 * d, d, d, sc_cross_kv_.w);
  }

}

std::vector<bool*> Engine::ok_flags() {
  std::vector<bool*> f{&sc_conv1_.f16_ok, &sc_conv2_.f16_ok};
  for (EncLayerScales& sl : sc_layers_) {
    for (bool* p : {&sl.qkv.f16_ok, &sl.attn_f16_ok, &sl.out.f16_ok, &sl.fc1.f16_ok, &sl.fc2.f16_ok}) f.push_back(p);
  }
  f.push_back(&sc_cross_kv_.f16_ok);
  return f;
}

void Engine::set_force_fallback(long mask) {
  require_idle();
  const std::vector<bool*> f = ok_flags();
  if (mask < 0 || (f.size() < 63 && (mask >> f.size()) != 0)) throw Error(kErrInvalidArg, "force_fallback: bit beyond the last contraction");
  n_f16_fallbacks_ = 0;
  for (size_t i = 0; i < f.size(); ++i) {
    *f[i] = load_ok_[i] && !((mask >> i) & 1);
    if (!*f[i]) ++n_f16_fallbacks_;
  }
  force_fallback_ = mask;
}

// bf16 storage mode (option "bf16", BASELINE configs[3]): bf16 copies of every matrix a kernel contracts with, made
// from the weight file the first time the mode is switched on.  Biases, LayerNorm gains / shifts, positional tables
// and the embedding rows the decoder looks up stay fp32 (epilogue / prologue operands, never MFMA operands), and so
// does the cross-attention query projection, which runs as an fp32 matrix-vector product inside cross_attention_step.
void Engine::ensure_bf16_weights() {
  if (bf16_ready_) return;
  if (dims_.n_audio_state == 0) throw Error(kErrUnsupported, "front-end-only engine: no model weights loaded");
  const WtwFile file(weights_path_);
  const wtw::Dims& c = dims_;
  const int d = c.n_audio_state, nm = c.n_mels;
  auto H = [&](const std::string& n, size_t expect) -> const float* { return file.get(n, expect); };
  auto up16 = [&](const std::vector<unsigned short>& v) -> const unsigned short* {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(unsigned short) + 256));
    allocations_.push_back(p);
    HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    return static_cast<const unsigned short*>(p);
  };
  const size_t dd = size_t(d) * d;
  {
    bf_.conv1_kpad = int(round_up(size_t(3) * nm, 64));
    std::vector<float> r(size_t(d) * 3 * nm);
    const float* w = H("encoder.conv1.weight", size_t(d) * nm * 3);
    for (int co = 0; co < d; ++co)
      for (int ci = 0; ci < nm; ++ci)
        for (int kk = 0; kk < 3; ++kk) r[size_t(co) * 3 * nm + kk * nm + ci] = w[(size_t(co) * nm + ci) * 3 + kk];
    bf_.conv1 = up16(round_weights_bf16(r.data(), d, 3 * nm, bf_.conv1_kpad));
    std::vector<float> r2(size_t(d) * 3 * d);
    const float* w2 = H("encoder.conv2.weight", size_t(d) * d * 3);
    for (int co = 0; co < d; ++co)
      for (int ci = 0; ci < d; ++ci)
        for (int kk = 0; kk < 3; ++kk) r2[size_t(co) * 3 * d + kk * d + ci] = w2[(size_t(co) * d + ci) * 3 + kk];
    bf_.conv2 = up16(round_weights_bf16(r2.data(), d, 3 * d, 3 * d));
  }
  auto fused_qkv = [&](const std::string& p) {
    std::vector<float> w(3 * dd);
    std::memcpy(w.data(), H(p + ".query.weight", dd), dd * 4);
    std::memcpy(w.data() + dd, H(p + ".key.weight", dd), dd * 4);
    std::memcpy(w.data() + 2 * dd, H(p + ".value.weight", dd), dd * 4);
    return w;
  };
  bf_.layers.resize(c.n_audio_layer);
  for (int l = 0; l < c.n_audio_layer; ++l) {
    const std::string blk = "encoder.blocks." + std::to_string(l);
    bf_.layers[l].qkv = up16(round_weights_bf16(fused_qkv(blk + ".attn").data(), 3 * d, d, d));
    bf_.layers[l].out = up16(round_weights_bf16(H(blk + ".attn.out.weight", dd), d, d, d));
    bf_.layers[l].fc1 = up16(round_weights_bf16(H(blk + ".mlp.0.weight", 4 * dd), 4 * d, d, d));
    bf_.layers[l].fc2 = up16(round_weights_bf16(H(blk + ".mlp.2.weight", 4 * dd), d, 4 * d, 4 * d));
  }
  std::vector<float> ckv_w(size_t(c.n_text_layer) * 2 * dd);
  dec_blocks_bf_ = dec_blocks_;  // fp32 vectors shared; the tiled matrices are replaced below
  for (int l = 0; l < c.n_text_layer; ++l) {
    const std::string blk = "decoder.blocks." + std::to_string(l);
    std::memcpy(ckv_w.data() + (size_t(l) * 2 + 0) * dd, H(blk + ".cross_attn.key.weight", dd), dd * 4);
    std::memcpy(ckv_w.data() + (size_t(l) * 2 + 1) * dd, H(blk + ".cross_attn.value.weight", dd), dd * 4);
    DecBlockWeights& bw = dec_blocks_bf_[l];
    bw.wqkv = TiledW{up16(tile_weights_bf16(fused_qkv(blk + ".attn").data(), 3 * d, d)), 1.0f};
    bw.wo = TiledW{up16(tile_weights_bf16(H(blk + ".attn.out.weight", dd), d, d)), 1.0f};
    bw.cross_wo = TiledW{up16(tile_weights_bf16(H(blk + ".cross_attn.out.weight", dd), d, d)), 1.0f};
    bw.w1 = TiledW{up16(tile_weights_bf16(H(blk + ".mlp.0.weight", 4 * dd), 4 * d, d)), 1.0f};
    bw.w2 = TiledW{up16(tile_weights_bf16(H(blk + ".mlp.2.weight", 4 * dd), d, 4 * d)), 1.0f};
    std::vector<float> A, av;  // absorbed cross-attention: the fused query matrix as bf16 (bq_abs, Wv, bv stay fp32: shared)
    absorbed_query_matrix(H(blk + ".cross_attn.query.weight", dd), H(blk + ".cross_attn.query.bias", d),
                          H(blk + ".cross_attn.key.weight", dd), c.n_text_head, d, &A, &av);
    bw.wq_abs = TiledW{up16(tile_weights_bf16(A.data(), c.n_text_head * d, d)), 1.0f};
  }
  bf_.cross_kv = up16(round_weights_bf16(ckv_w.data(), c.n_text_layer * 2 * d, d, d));
  tok_emb_tiled_bf_ = TiledW{up16(tile_weights_bf16(H("decoder.token_embedding.weight", size_t(c.n_vocab) * d), c.n_vocab, d)), 1.0f};
  bf16_ready_ = true;
}

namespace {
// The linear map the reference's transform actually applies.  whisper.cpp:58-106 is a
// radix-2 recursion (400 -> 200 -> 100 -> 50 -> 25) over a naive 25-point DFT (:37-54) whose
// twiddles are cosf/sinf of an angle already rounded to float (:46, :92): at the leaf the
// angle reaches 2*pi*24*24/25 = 145 rad, so the rounded argument is off by up to ~1e-5 rad.
// That systematic deviation from the ideal DFT is 100x above fp32 rounding noise and shows in
// every bin that sits 40 dB below a frame's peak, so the device transform is built from the
// SAME twiddle values: the recursion is evaluated here in double on unit impulses, giving
// the effective 400x400 complex matrix, which the GPU then applies as one fp32 MFMA GEMM.
using cd = std::complex<double>;
void effective_fft(const std::vector<cd>& in, std::vector<cd>& out) {
  const int N = int(in.size());
  out.assign(N, cd(0, 0));
  if (N == 1) {
    out[0] = in[0];
    return;
  }
  if (N % 2 == 1) {
    for (int k = 0; k < N; ++k) {
      cd acc(0, 0);
      for (int n = 0; n < N; ++n) {
        double a = 2 * M_PI;
        a = a * k;
        a = a * n;
        a = a / N;
        const float angle = static_cast<float>(a);
        acc += in[n] * cd(double(cosf(angle)), -double(sinf(angle)));
      }
      out[k] = acc;
    }
    return;
  }
  std::vector<cd> even(N / 2), odd(N / 2), fe, fo;
  for (int i = 0; i < N / 2; ++i) {
    even[i] = in[2 * i];
    odd[i] = in[2 * i + 1];
  }
  effective_fft(even, fe);
  effective_fft(odd, fo);
  for (int k = 0; k < N / 2; ++k) {
    double a = 2 * M_PI;
    a = a * k;
    a = a / N;
    const float theta = static_cast<float>(a);
    const cd w(double(cosf(theta)), -double(sinf(theta)));
    out[k] = fe[k] + w * fo[k];
    out[k + N / 2] = fe[k] - w * fo[k];
  }
}
}  // namespace

void Engine::build_frontend_tables() {
  // STFT as a GEMM: rows of the basis are the reference's effective transform with the Hann
  // window (whisper.cpp:117-120: double cos, float store) folded in.  All 400 output bins are
  // kept because the reference adds the mirror bin's power computed by the same inexact
  // transform (:164-166).
  const int n_fft = 400, n_bins = 201;
  if (filters_.n_fft != n_bins || filters_.n_mel != dims_.n_mels) {
    have_logmel_ = false;  // front end needs the 80x201 bank; encoder/decoder still work
    return;
  }
  // Round 4: the STFT runs on the fp16-plane GEMM (PCM is bounded, the windowed basis is bounded: the two-plane form is
  // admissible) and writes the POWER spectrum from its epilogue (kEpiPower).  Rows (2 k, 2 k + 1) of the basis are the
  // real and imaginary parts of bin k for k = 0 .. 200 only: the input is real, so the mirror bin the reference adds
  // (P[j] += P[400 - j], whisper.cpp:164-166) is folded into the row of bin j (below).  402 rows, padded to 512;
  // round 3 contracted all 800 rows (896 padded) on six bf16 products and squared / folded in a kernel of its own.
  dft_k = int(round_up(n_fft, 32));  // 416: samples 400..415 meet zero basis entries
  dft_n = int(round_up(2 * n_bins, 128));  // 512
  std::vector<float> basis(size_t(dft_n) * dft_k, 0.0f);
  std::vector<cd> impulse(n_fft), col;
  const double r2 = std::sqrt(2.0);
  float bmax = 0.0f;
  for (int n = 0; n < n_fft; ++n) {
    const float hann = static_cast<float>(0.5 * (1.0 - std::cos((2.0 * M_PI * n) / n_fft)));
    std::fill(impulse.begin(), impulse.end(), cd(0, 0));
    impulse[n] = cd(double(hann), 0);
    effective_fft(impulse, col);
    for (int k = 0; k < n_bins; ++k) {
      // bins 1 .. 199: sqrt(2) * (X_k + conj(X_{400-k})) / 2 — twice its squared magnitude is |X_k|^2 + |X_{400-k}|^2 up to
      // the SQUARE of the two bins' difference (the reference's inexact twiddles make them differ by ~1e-7 of the
      // largest amplitude, which shows at the 2e-4 level in bins eight decades below the maximum: a sweep's far bins)
      const bool folded = k >= 1 && k < n_fft / 2;
      const cd y = folded ? r2 * 0.5 * (col[k] + std::conj(col[n_fft - k])) : col[k];
      const float re = static_cast<float>(y.real()), im = static_cast<float>(y.imag());
      basis[size_t(2 * k) * dft_k + n] = re;
      basis[size_t(2 * k + 1) * dft_k + n] = im;
      bmax = std::max(bmax, std::max(std::fabs(re), std::fabs(im)));
    }
  }
  dft_w_scale_ = f16_scale_for(bmax);
  dft_basis_p_ = upload_planes(basis.data(), dft_n, dft_k, dft_k, dft_w_scale_);
  dft_zero_bias_ = upload(std::vector<float>(size_t(dft_n), 0.0f));
  pw_ld_ = dft_n / 2;  // 256 powers per frame row (201 used)
  mel_k = int(round_up(n_bins, 32));  // 224 (the mel GEMM reads rows of pw_ld_ = 256 powers)
  mel_n = int(round_up(size_t(dims_.n_mels), 128));
  std::vector<float> mw(size_t(mel_n) * mel_k, 0.0f);
  for (int j = 0; j < dims_.n_mels; ++j)
    for (int k = 0; k < n_bins; ++k) mw[size_t(j) * mel_k + k] = filters_.data[size_t(j) * n_bins + k];
  mel_w = upload(mw);
  have_logmel_ = true;
}

// ---------------------------------------------------------- lifecycle ---

void Engine::open_device() {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    throw Error(5, "no HIP device available: this engine has no CPU fallback");
  }
  if (device_ < 0 || device_ >= n_dev) throw Error(5, "device_id out of range");
  HIPCHK(hipSetDevice(device_));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    throw Error(5, std::string("device is ") + prop.gcnArchName +
                            ", kernels are built for gfx950 only");
  }
  n_cu_ = prop.multiProcessorCount;
}

void Engine::create_streams() {
  int prio_lo = 0, prio_hi = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  HIPCHK(hipStreamCreateWithPriority(&stream_full_, hipStreamNonBlocking, prio_lo));
  // Pipelined batches run the encoder on a stream that leaves `reserve` CUs of every XCD to the
  // decoders (mask bit i = CU i/8 of XCD i%8): measured on MI355X, three decoder chains next
  // to an encoder that owns every CU stretch it by 2.5 ms per batch, next to one that owns
  // 224 of 256 CUs by 0.9 ms; with the faster encoder kernels 4..8 CUs per XCD measure
  // within 2 % of each other. Synchronous calls keep the whole chip.
  // (round 4: the absorbed cross-attention needs a third less CU time — 4 per XCD = 224 CUs for the encoder measured
  // 148.6 k against 146.8 k audio-sec/s with 8, 5 / 6 in between, 3: 139.5 k; tools/ab_r4_pipeline.sh)
  int reserve = 4;  // 224 CUs for the pipelined encoder: 188 row tiles of 256 fill one round
  if (const char* v = getenv("WT_ENC_CU_RESERVE")) reserve = std::min(std::max(atoi(v), 0), 16);
  reserve_ = reserve;
  const int n_cu = n_cu_;
  if (reserve > 0 && n_cu >= 64 && n_cu % 8 == 0) {
    int keep = n_cu - 8 * reserve;
    if (const char* v = getenv("WT_ENC_CU_KEEP")) keep = std::min(std::max(atoi(v), 64), n_cu);  // CUs of the masked stream
    enc_cus_masked_ = keep;
    std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
    for (int i = 0; i < keep; ++i) mask[i / 32] |= 1u << (i % 32);
    HIPCHK(hipExtStreamCreateWithCUMask(&stream_masked_, uint32_t(mask.size()), mask.data()));
  }
  stream_ = stream_full_;
  HIPCHK(hipEventCreate(&ev_switch_));
  if (const char* v = getenv("WT_DEC_STREAMS")) n_dec_streams_ = std::min(std::max(atoi(v), 1), kDecStreams);
  // WT_DEC_PARTITION=1 (measurement knob): the decoder streams are confined to the CUs the pipelined encoder stream
  // leaves free — a strict partition instead of "decoders may run anywhere"
  const char* part = getenv("WT_DEC_PARTITION");
  if (part && atoi(part) >= 1 && enc_cus_masked_ > 0 && enc_cus_masked_ < n_cu) {
    // 1 = the reserved CUs only; N > 1 = the reserved CUs and the N CUs of the encoder's share next to them (soft partition)
    const int shared = atoi(part) > 1 ? std::min(atoi(part), enc_cus_masked_) : 0;
    std::vector<uint32_t> dmask((n_cu + 31) / 32, 0u);
    for (int i = enc_cus_masked_ - shared; i < n_cu; ++i) dmask[i / 32] |= 1u << (i % 32);
    for (auto& ds : dstream_) HIPCHK(hipExtStreamCreateWithCUMask(&ds, uint32_t(dmask.size()), dmask.data()));
  } else {
    const char* dp = getenv("WT_DEC_PRIO");  // measurement knob: "lo" = decoder streams at the encoder's priority
    const int prio = dp && dp[0] == 'l' ? prio_lo : prio_hi;
    // (round 4: candidates on a second priority level, or GPU_MAX_HW_QUEUES=8, give the probe no further stream that
    // overlaps with the four in use — the process has four hardware queues, the encoder's and three decoders')
    for (auto& ds : dstream_) HIPCHK(hipStreamCreateWithPriority(&ds, hipStreamNonBlocking, prio));
  }
  pick_decoder_streams();
  for (auto& e : ev_) HIPCHK(hipEventCreate(&e));
  for (Slot& sl : slots_) {
    for (hipEvent_t* e : {&sl.enc_begin, &sl.enc_mid, &sl.enc_done, &sl.dec_begin, &sl.dec_done}) {
      HIPCHK(hipEventCreate(e));
    }
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_ids), 4096 * 32 * sizeof(long long), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_n), 4096 * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_flag), sizeof(int), hipHostMallocDefault));
    *sl.h_flag = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&sl.d_flag), sizeof(int)));
  }
}

Engine::Engine(const std::string& model_prefix, const std::string& vocab_path, bool multilingual,
               int device_id, bool monolith, const std::string& weights_path)
    : device_(device_id), monolith_(monolith), multilingual_(multilingual) {
  // vocab first: a missing vocab file throws exactly like the reference's MmapFile
  read_vocab_file(vocab_path, multilingual, &filters_, &vocab_);
  if (monolith) language = 0;  // HF generate() forces <|en|> for a multilingual checkpoint (export/generate.py:24-30)
  open_device();
  // A constructor that throws does not run the destructor: everything acquired below (streams, events,
  // pinned host memory, device allocations) is released here before the exception leaves, so a failed
  // wt_engine_create (missing or malformed .wtw) leaks nothing.
  try {
    create_streams();
    upload_weights(weights_path.empty() ? model_prefix + ".wtw" : weights_path);
    if (vocab_.n_vocab != dims_.n_vocab && verbose) {
      std::fprintf(stderr, "[wt] note: vocab file n_vocab %d != model n_vocab %d\n", vocab_.n_vocab,
                   dims_.n_vocab);
    }
    build_frontend_tables();
  } catch (...) {
    release();
    throw;
  }
}

Engine::Engine(const FilterBank& filters, int device_id) : device_(device_id), filters_(filters) {
  // front end only (whisper::log_mel_spectrogram as a free function, whisper.h:123): no weights, the
  // reference's fixed audio geometry (whisper.h:34-39)
  dims_ = wtw::Dims{};
  dims_.n_mels = filters.n_mel;
  dims_.n_audio_ctx = 1500;
  open_device();
  try {
    create_streams();
    build_frontend_tables();
  } catch (...) {
    release();
    throw;
  }
}

Engine::~Engine() { release(); }

void Engine::release() noexcept {
  (void)hipSetDevice(device_);
  for (hipStream_t st : {stream_full_, stream_masked_})
    if (st) (void)hipStreamSynchronize(st);
  for (auto& ds : dstream_)
    if (ds) (void)hipStreamSynchronize(ds);
  for (auto& g : graphs_) (void)hipGraphExecDestroy(g.second.exec);
  graphs_.clear();
  for (void* p : ws_.owned) (void)hipFree(p);
  ws_.owned.clear();
  for (void* p : allocations_) (void)hipFree(p);
  allocations_.clear();
  for (auto& e : ev_) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  for (Slot& sl : slots_) {
    for (hipEvent_t* e : {&sl.enc_begin, &sl.enc_mid, &sl.enc_done, &sl.dec_begin, &sl.dec_done}) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
    for (auto& e : sl.kt_events) (void)hipEventDestroy(e);
    for (auto& e : sl.dt_events) (void)hipEventDestroy(e);
    sl.kt_events.clear();
    sl.dt_events.clear();
    if (sl.h_ids) (void)hipHostFree(sl.h_ids);
    if (sl.h_n) (void)hipHostFree(sl.h_n);
    if (sl.h_flag) (void)hipHostFree(sl.h_flag);
    if (sl.d_flag) (void)hipFree(sl.d_flag);
    sl.h_ids = nullptr, sl.h_n = nullptr, sl.h_flag = nullptr, sl.d_flag = nullptr;
  }
  if (ev_switch_) (void)hipEventDestroy(ev_switch_);
  if (trace_base_) (void)hipEventDestroy(trace_base_);
  ev_switch_ = nullptr, trace_base_ = nullptr;
  for (hipStream_t* st : {&stream_full_, &stream_masked_}) {
    if (*st) (void)hipStreamDestroy(*st);
    *st = nullptr;
  }
  for (auto& ds : dstream_) {
    if (ds) (void)hipStreamDestroy(ds);
    ds = nullptr;
  }
  stream_ = nullptr;
}

// The runtime multiplexes the streams of a process onto a few hardware queues, and two streams that share a queue run
// their kernels one after the other.  Which streams share depends on everything the process created before (a second
// engine in one process measured 108 k instead of 131 k audio-sec/s: one of its decoder streams sat on the encoder
// stream's queue).  So the engine does not trust creation order: it times a 300 us one-wavefront kernel on pairs of
// streams (concurrent ~0.3 ms, serialised ~0.6 ms) and moves to the front of dstream_ decoder streams that overlap with
// the pipelined encoder stream and with each other.  ~10 ms at engine creation; WT_NO_STREAM_PROBE=1 skips it.
void Engine::pick_decoder_streams() {
  if (getenv("WT_NO_STREAM_PROBE") || getenv("WT_DEC_PARTITION")) return;
  hipStream_t enc = stream_masked_ ? stream_masked_ : stream_full_;
  constexpr int kSpinUs = 10, kChain = 20;  // a chain of dependent short kernels per stream, like a decoder chain
  const bool trace = getenv("WT_STREAM_PROBE_TRACE") != nullptr;
  auto chain_us = [&](hipStream_t a, hipStream_t b) {
    HIPCHK(hipStreamSynchronize(a));
    if (b) HIPCHK(hipStreamSynchronize(b));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < kChain; ++i) {
      launch_spin(kSpinUs, a);
      if (b) launch_spin(kSpinUs, b);
    }
    HIPCHK(hipStreamSynchronize(a));
    if (b) HIPCHK(hipStreamSynchronize(b));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  launch_spin(50, enc);  // first launch of the kernel (code object load) outside the timed pairs
  double alone = chain_us(enc, nullptr);  // the reference is the fastest of three (a slow first run would hide clashes)
  for (int i = 0; i < 2; ++i) alone = std::min(alone, chain_us(enc, nullptr));
  // a pair shares a hardware queue when it is slow in BOTH of two runs: the timing is host wall-clock over ~0.5 ms, and a
  // single run disturbed by the host (a page fault, another rank starting) would otherwise reject a good stream — or the
  // choice, and with it the throughput, would differ between two runs of the same program
  auto serialised = [&](hipStream_t a, hipStream_t b) {
    const double us = std::min(chain_us(a, b), chain_us(a, b));
    if (trace) std::fprintf(stderr, "[wt] stream probe: pair %p %p %.0f us (one chain alone %.0f us)\n", (void*)a, (void*)b, us, alone);
    return us > 1.5 * alone;
  };
  int chosen = 0;
  const int want = std::min(kDecStreams, n_dec_streams_ + 2);  // + spare streams for the latency form (submit_decoder)
  for (int i = 0; i < kDecStreams && chosen < want; ++i) {
    bool clash = serialised(enc, dstream_[i]);
    for (int j = 0; j < chosen && !clash; ++j) clash = serialised(dstream_[j], dstream_[i]);
    if (clash) continue;
    std::swap(dstream_[chosen], dstream_[i]);
    ++chosen;
  }
  n_spare_streams_ = std::max(0, chosen - n_dec_streams_);
  if (trace || verbose) {
    std::fprintf(stderr, "[wt] stream probe: %d decoder streams (%d wanted + spares) run beside the encoder stream:", chosen, n_dec_streams_);
    for (int i = 0; i < chosen; ++i) std::fprintf(stderr, " %p", (void*)dstream_[i]);
    std::fprintf(stderr, "\n");
  }
}

void Engine::select_stream(bool pipelined) {
  pipelined_call_ = pipelined;
  hipStream_t target = pipelined && stream_masked_ ? stream_masked_ : stream_full_;
  if (target == stream_) return;
  // everything already enqueued on the old stream stays ahead of what follows on the new one
  HIPCHK(hipEventRecord(ev_switch_, stream_));
  HIPCHK(hipStreamWaitEvent(target, ev_switch_, 0));
  stream_ = target;
}

void Engine::bind_device() { HIPCHK(hipSetDevice(device_)); }

void Engine::sync() {
  HIPCHK(hipStreamSynchronize(stream_full_));
  if (stream_masked_) HIPCHK(hipStreamSynchronize(stream_masked_));
  for (auto& ds : dstream_) HIPCHK(hipStreamSynchronize(ds));
}

// The cross K/V cache (layers x 2 x clips x 1500 x d fp32: 590 MB per slot for tiny at 32 clips, 1.2 GB for base at 64)
// belongs to the cached decoder form only — cross_absorb = 0, synchronous calls below 32 clips, a flagged cross
// operand — so it is allocated per slot the first time a batch of that form is encoded into the slot, not for all
// every slot up front (the default absorbed form never touches it).
void Engine::need_cross_kv(Slot& slot) {
  if (slot.cross_kv) return;
  const wtw::Dims& c = dims_;
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, size_t(c.n_text_layer) * 2 * size_t(ws_.batch) * c.n_audio_ctx * c.n_audio_state * sizeof(float)));
  ws_.owned.push_back(p);
  slot.cross_kv = static_cast<float*>(p);
}

void Engine::ensure_batch(int batch) {
  if (batch <= 0 || batch > 4096) throw Error(1, "batch must be in [1, 4096]");
  HIPCHK(hipSetDevice(device_));
  if (batch <= ws_.batch) return;
  if (!inflight_.empty()) throw Error(1, "cannot grow the workspace while batches are in flight");
  HIPCHK(hipStreamSynchronize(stream_));
  for (auto& ds : dstream_) HIPCHK(hipStreamSynchronize(ds));
  for (auto& g : graphs_) (void)hipGraphExecDestroy(g.second.exec);  // captured pointers die with the workspace
  graphs_.clear();
  for (void* p : ws_.owned) (void)hipFree(p);
  ws_ = Workspace();
  const wtw::Dims& c = dims_;
  const size_t B = batch, T0 = mel_frames(), T = c.n_audio_ctx, d = c.n_audio_state;
  auto alloc = [&](size_t n_floats, bool zero) -> float* {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, n_floats * sizeof(float)));
    ws_.owned.push_back(p);
    if (zero) HIPCHK(hipMemsetAsync(p, 0, n_floats * sizeof(float), stream_));
    return static_cast<float*>(p);
  };
  // time-major, one zero row before and after each clip: the k=3 convolutions become
  // plain GEMMs over three consecutive rows
  if (d == 0) {  // front-end-only engine: the staging mel buffer is all the batch needs
    ws_.mel_stage = alloc(B * mel_elems(), false);
    ws_.batch = batch;
    HIPCHK(hipStreamSynchronize(stream_));
    return;
  }
  ws_.melT = alloc(B * (T0 + 2) * c.n_mels + 256, true);
  ws_.h1p = alloc(B * (T0 + 2) * d + 256, true);
  ws_.melTp = reinterpret_cast<unsigned short*>(alloc(B * (T0 + 2) * c.n_mels + 256, true));  // 2 planes of halfs
  ws_.h1pp = reinterpret_cast<unsigned short*>(alloc(B * (T0 + 2) * d + 256, true));
  ws_.x = alloc(B * T * d, false);
  ws_.ln = alloc(B * T * d, false);
  ws_.qkv = alloc(B * T * 3 * d, false);
  ws_.att = alloc(B * T * d, false);
  ws_.hid = alloc(B * T * 4 * d, false);
  ws_.enc_out = alloc(B * T * d, false);
  ws_.cvt = reinterpret_cast<unsigned short*>(alloc(B * T * 4 * d, false));
  for (Slot& sl : slots_) {
    sl.cross_kv = nullptr;  // the cross K/V cache of the CACHED decoder form: allocated when a batch first takes that form (need_cross_kv)
    sl.e_planes = reinterpret_cast<unsigned short*>(alloc(B * T * d, false));  // two planes of halfs
    sl.used = false;
  }
  // decoder rows: one position of the batch, or all prompt positions of a <= 32-clip batch in one pass
  const size_t R = std::max<size_t>(B, kDecRowsMax);
  for (DecWorkspace& dw : dws_) {
    dw.xd = alloc(R * d, false);
    dw.xb = alloc(R * d, false);
    dw.xpart = alloc(R * d, false);
    dw.qkvd = alloc(R * 3 * d, false);
    dw.attd = alloc(R * d, false);
    dw.hd = alloc(R * 4 * d, false);
    dw.cross_ws = alloc(R * c.n_text_head * 8 * 68, false);
    dw.qp = alloc(R * c.n_text_head * d, false);
    dw.abs_ws = alloc(R * c.n_text_head * 16 * (d + 4), false);
    dw.cabs = alloc(R * c.n_text_head * d, false);
    dw.self_kv = alloc(size_t(c.n_text_layer) * 2 * B * self_cap_ * d, true);
    dw.logits = alloc(B * c.n_vocab, false);
    dw.best = reinterpret_cast<unsigned long long*>(alloc(B * 2 * size_t((c.n_vocab + 31) / 32), true));
    dw.ids = reinterpret_cast<long long*>(alloc(B * 32 * 2, true));
    dw.n_ids = reinterpret_cast<int*>(alloc(B, true));
    dw.finished = reinterpret_cast<int*>(alloc(B, true));
  }
  ws_.mel_stage = alloc(B * mel_elems(), false);
  ws_.batch = batch;
  HIPCHK(hipStreamSynchronize(stream_));
}

float* Engine::staging_mel(int batch) {
  ensure_batch(batch);
  return ws_.mel_stage;
}

float* Engine::staging_pcm(int batch) {
  ensure_batch(batch);
  if (!ws_.pcm_stage) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, size_t(ws_.batch) * pcm_elems() * sizeof(float)));
    ws_.owned.push_back(p);
    ws_.pcm_stage = static_cast<float*>(p);
  }
  return ws_.pcm_stage;
}

// ---------------------------------------------------------- front end ---

void Engine::logmel(const float* d_pcm, int batch, float* d_mel, int valid_frames) {
  if (!have_logmel_) throw Error(3, "vocab file carries no 80x201 mel filter bank");
  ensure_batch(batch);
  const size_t T0 = mel_frames(), n_samples = pcm_elems(), pad = n_samples + 512;
  if (!ws_.pcm_planes) {
    const size_t Bc = ws_.batch;
    auto alloc = [&](size_t n_floats) -> float* {
      void* p = nullptr;
      HIPCHK(hipMalloc(&p, n_floats * sizeof(float)));
      ws_.owned.push_back(p);
      HIPCHK(hipMemsetAsync(p, 0, n_floats * sizeof(float), stream_));
      return static_cast<float*>(p);
    };
    // PCM as two fp16 planes, [clip][480000 + 512] each, the tail of every clip zero for good (the reference's zero fill
    // past the last sample, whisper.cpp:149-153: only the first 480000 columns are ever written)
    pcm_plane_ = long(Bc * pad + 1024);
    ws_.pcm_planes = reinterpret_cast<unsigned short*>(alloc((size_t(pcm_plane_) * 2 * sizeof(unsigned short) + 3) / 4 + 64));
    ws_.pw = alloc(Bc * T0 * size_t(pw_ld_));
    ws_.melacc = alloc(Bc * T0 * mel_n);
    ws_.clip_max = reinterpret_cast<unsigned*>(alloc(Bc * kClipMaxStride * kClipMaxWays));
  }
  HIPCHK(hipEventRecord(ev_[0], stream_));
  const long M = long(batch) * long(T0);
  // STFT as a plane GEMM: frame i of a clip is the 416 samples from 160 i (hop-strided rows read in place), PCM within
  // +-kPcmBound (beyond: clamped), power spectrum out of the epilogue
  launch_pcm_to_planes(d_pcm, ws_.pcm_planes, pcm_plane_, f16_scale_for(kPcmBound), kPcmBound, batch, long(n_samples), long(pad),
                       stream_);
  PlaneGemmArgs g;
  g.A = ws_.pcm_planes;
  g.a_plane = pcm_plane_;
  g.a_rpb = int(T0);
  g.a_bs = long(pad);
  g.lda = 160;  // hop: frame i starts at sample 160 * i
  g.W = dft_basis_p_.w;
  g.bias = dft_zero_bias_;
  g.C = ws_.pw;
  g.M = int(M);
  g.N = dft_n;
  g.K = dft_k;
  g.ldc = pw_ld_;
  g.a_scale = f16_scale_for(kPcmBound);
  g.w_scale = dft_w_scale_;
  g.n_cu = stream_ == stream_masked_ && enc_cus_masked_ > 0 ? enc_cus_masked_ : n_cu_;
  launch_gemm_planes(g, kEpiBias | kEpiPower, stream_);
  GemmArgs m;
  m.A = ws_.pw;
  m.lda = pw_ld_;
  m.W = mel_w;
  m.C = ws_.melacc;
  m.M = int(M);
  m.N = mel_n;
  m.K = mel_k;
  m.ldc = mel_n;
  // the mel GEMM stays on the full-range three-plane kernel: power values span more decades than two fp16 planes keep
  m.variant = gemm_variant >= 0 ? int(gemm_variant) : 13;
  launch_gemm(m, 0, stream_);
  HIPCHK(hipMemsetAsync(ws_.clip_max, 0, sizeof(unsigned) * batch * kClipMaxStride * kClipMaxWays, stream_));
  launch_log_clipmax(ws_.melacc, mel_n, d_mel, ws_.clip_max, batch, dims_.n_mels, int(T0), stream_, valid_frames);
  launch_mel_normalize(d_mel, ws_.clip_max, batch, dims_.n_mels, int(T0), stream_);
  HIPCHK(hipEventRecord(ev_[1], stream_));
  timings_.logmel_ms = -1.0f;  // resolved lazily in decode()/sync by the C ABI
}

// ------------------------------------------------------- kernel timer ---

thread_local LaunchTimer g_launch_timer;

namespace {
// The dispatch-attached event pair must not outlive the encoder pass that armed it: a launcher that throws (shape or span
// error) between kt_begin and kt_end would leave it armed for the thread's next timed launch — possibly another engine's.
struct TimerDisarm {
  ~TimerDisarm() { g_launch_timer = LaunchTimer{}; }
};
}  // namespace

void Engine::kt_begin(int cls, double flops, double bytes) {
  if (!kt_on_) return;
  Slot& sl = slots_[enc_slot_];
  const size_t idx = sl.kt_cls.size() * 2;
  while (sl.kt_events.size() < idx + 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    sl.kt_events.push_back(e);
  }
  sl.kt_cls.push_back(cls);
  sl.kt_flops.push_back(flops);
  sl.kt_bytes.push_back(bytes);
  // the default contraction kernels take the pair on the dispatch itself (kernels.h, LaunchTimer): kernel begin -> end
  if (cls == kKcGemm || cls == kKcEncAttn) {
    g_launch_timer.start = sl.kt_events[idx];
    g_launch_timer.stop = sl.kt_events[idx + 1];
    return;
  }
  HIPCHK(hipEventRecord(sl.kt_events[idx], stream_));
}

void Engine::kt_end() {
  if (!kt_on_) return;
  if (g_launch_timer.start) {
    g_launch_timer = LaunchTimer{};
    return;
  }
  Slot& sl = slots_[enc_slot_];
  HIPCHK(hipEventRecord(sl.kt_events[(sl.kt_cls.size() - 1) * 2 + 1], stream_));
}

void Engine::resolve_kernel_stats(int slot) {
  Slot& sl = slots_[slot];
  // class names = the kernels the current options select (what rocprofv3 lists)
  const long gv = gemm_variant;
  kstats_[kKcGemm].name = bf16 ? "gemm_bf16_planes" : "gemm_planes";  // the class: gemm_planes_pp16 / _tile instantiations
  kstats_[kKcEncAttn].name = bf16 ? "encoder_attention_bf16" : "encoder_attention_planes";
  kstats_[kKcGemmAlt].name = gv == 0 ? "gemm_f32_tile" : "gemm_split16_tile";
  kstats_[kKcEncAttnAlt].name = attn_variant == 0 ? "encoder_attention_f32" : "encoder_attention_split";
  for (auto& k : kstats_) k.launches = 0, k.ms = 0, k.flops = 0, k.bytes = 0;
  for (size_t i = 0; i < sl.kt_cls.size(); ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, sl.kt_events[2 * i], sl.kt_events[2 * i + 1]) != hipSuccess) continue;
    KernelStat& k = kstats_[sl.kt_cls[i]];
    k.launches += 1;
    k.ms += ms;
    k.flops += sl.kt_flops[i];
    k.bytes += sl.kt_bytes[i];
  }
}

// ------------------------------------------------------------ encoder ---

void Engine::require_idle() const {
  // A synchronous call would take the next pipeline slot: with batches in flight that slot may still hold an
  // uncollected batch (its cross-KV cache, its id buffers).  Refuse BEFORE anything is enqueued.
  if (!inflight_.empty()) throw Error(kErrInvalidArg, "collect the submitted batches before a synchronous call");
}

void Engine::encode(const float* d_mel, int batch) {
  require_idle();
  select_stream(false);
  encode_enqueue(d_mel, batch);
}

void Engine::encode_enqueue(const float* d_mel, int batch) {
  if (dims_.n_audio_state == 0) throw Error(kErrUnsupported, "front-end-only engine: no model weights loaded");
  ensure_batch(batch);
  TimerDisarm disarm_on_exit;  // (covers the bf16 pass too: it is called from here)
  // per-launch event pairs (bench.py's live roofline figures) on every kernel_timers-th encoder pass: an event pair
  // costs the stream a few microseconds per launch, which 41 launches per pass make visible in the pipeline period
  kt_on_ = kernel_timers > 0 && (enc_count_++ % kernel_timers) == 0;
  if (bf16) {
    encode_enqueue_bf16(d_mel, batch);
    return;
  }
  // The encoder graph.  Every contraction chooses its kernel BY ITSELF: the plane kernels
  // (k_gemm_planes.hip, k_attention_planes.hip: operands as two fp16 planes, made by the kernel that produces them)
  // unless the load-time slack check flagged that contraction's operand (GemmScale::f16_ok, upload_weights) or an
  // explicit gemm_variant / attn_variant asks for the fp32-storage kernels (k_gemm.hip, k_attention.hip: full fp32
  // operand range).  A tensor has exactly one consumer, so its format follows the consumer: a plane kernel that feeds a
  // flagged contraction writes fp32 (C instead of P), LayerNorm has both forms, and where a fall-back kernel feeds a
  // plane kernel its fp32 output is split by launch_f32_to_planes into the scratch planes ws_.cvt.  The residual
  // stream x stays fp32 throughout.
  const wtw::Dims& c = dims_;
  const int T0 = mel_frames(), T = c.n_audio_ctx, d = c.n_audio_state, M = batch * T, nm = c.n_mels;
  const long Bw = ws_.batch;  // plane strides follow the workspace, not the call
  const int cus = (stream_ == stream_masked_ && stream_masked_) ? enc_cus_masked_ : n_cu_;  // CUs of this stream
  Slot& slot = slots_[enc_slot_];
  // the slot's cross-KV cache may still be read by the decoder of the batch before last
  if (slot.used) HIPCHK(hipStreamWaitEvent(stream_, slot.dec_done, 0));
  slot.kt_cls.clear();
  slot.kt_flops.clear();
  slot.kt_bytes.clear();
  slot.batch = batch;
  if (!trace_base_ && getenv("WT_TRACE_PIPELINE")) {
    HIPCHK(hipEventCreate(&trace_base_));
    HIPCHK(hipEventRecord(trace_base_, stream_));
  }
  HIPCHK(hipMemsetAsync(slot.d_flag, 0, sizeof(int), stream_));
  HIPCHK(hipEventRecord(slot.enc_begin, stream_));
  unsigned short* const lnp = reinterpret_cast<unsigned short*>(ws_.ln);
  unsigned short* const qkvp = reinterpret_cast<unsigned short*>(ws_.qkv);
  unsigned short* const attp = reinterpret_cast<unsigned short*>(ws_.att);
  unsigned short* const hidp = reinterpret_cast<unsigned short*>(ws_.hid);
  const long melT_plane = Bw * (T0 + 2) * nm + 128, h1p_plane = Bw * (T0 + 2) * d + 128;
  const long ln_plane = Bw * T * d, qkv_plane = Bw * T * 3 * d, hid_plane = Bw * T * 4 * d, cvt_plane = Bw * T * 4 * d;
  const long e_plane = Bw * T * d;
  const bool absorb = absorb_for(batch);
  const int alt = alt_gemm_variant();

  auto plane_gemm = [&](PlaneGemmArgs& g, const GemmScale& sc, int epi, double flops) {
    g.n_cu = cus; g.a_scale = sc.a; g.w_scale = sc.w;
    kt_begin(kKcGemm, flops, 0);
    const bool ln_fused = launch_gemm_planes(g, epi, stream_);
    kt_end();
    return ln_fused;
  };
  // The LayerNorm that follows a GEMM whose output is the residual stream x (conv2, out-projection, fc2) is made by
  // that GEMM's epilogue when its 384-column tile owns whole rows and both sides run on planes; ln_done then tells the
  // consumer's side not to launch the LayerNorm kernel.
  bool ln_done = false;
  static const bool no_ln_fuse = getenv("WT_NO_LN_FUSE") != nullptr;  // measurement knob: separate LayerNorm launches
  auto fuse_ln = [&](PlaneGemmArgs& g, const float* gain, const float* shift, float scale) {
    if (no_ln_fuse) return;
    g.ln_g = gain; g.ln_b = shift; g.ln_P = lnp; g.ln_plane = ln_plane; g.ln_scale = scale;
  };
  auto alt_gemm = [&](GemmArgs& g, const GemmScale& sc, int epi, double flops) {
    g.variant = alt; g.a_scale = sc.a; g.w_scale = sc.w;
    kt_begin(kKcGemmAlt, flops, 0);
    launch_gemm(g, epi, stream_);
    kt_end();
  };
  auto to_planes = [&](const float* src, unsigned short* dst, long plane, long rows, int ld, const float* scales, int seg) {
    kt_begin(kKcConvert, 0, 2.0 * rows * ld * 4);
    launch_f32_to_planes(src, dst, plane, rows, ld, scales, seg, stream_);
    kt_end();
  };

  const bool c1p = gemm_on_planes(sc_conv1_), c2p = gemm_on_planes(sc_conv2_);
  kt_begin(kKcTranspose, 0, 2.0 * batch * nm * T0 * 4);
  if (c1p) {
    launch_mel_transpose_planes(d_mel, ws_.melTp, melT_plane, sc_conv1_.a, batch, nm, T0, nm, stream_);
  } else {
    launch_mel_transpose(d_mel, ws_.melT, batch, nm, T0, stream_);
  }
  kt_end();
  // conv1 + GELU: rows (clip, t) read melT rows t..t+2 (input t-1..t+1); row t lands at padded row t + 1 of h1
  if (c1p) {
    PlaneGemmArgs g;
    g.A = ws_.melTp; g.a_plane = melT_plane; g.a_rpb = T0; g.a_bs = long(T0 + 2) * nm; g.lda = nm;
    g.W = conv1_p_.w; g.bias = conv1_b;
    if (c2p) {
      g.P = ws_.h1pp + d; g.p_plane = h1p_plane; g.out_scale[0] = sc_conv2_.a;
    } else {
      g.C = ws_.h1p + d;
    }
    g.c_rpb = T0; g.c_bs = long(T0 + 2) * d; g.ldc = d;
    g.M = batch * T0; g.N = d; g.K = conv1_kpad_p_;
    plane_gemm(g, sc_conv1_, kEpiBias | kEpiGelu, 2.0 * g.M * g.N * (3.0 * nm));
  } else {
    GemmArgs g;
    g.A = ws_.melT; g.a_rpb = T0; g.a_bs = long(T0 + 2) * nm; g.lda = nm;
    g.W = conv1_w; g.bias = conv1_b;
    g.C = ws_.h1p + d; g.c_rpb = T0; g.c_bs = long(T0 + 2) * d; g.ldc = d;
    g.M = batch * T0; g.N = d; g.K = conv1_kpad;
    alt_gemm(g, sc_conv1_, kEpiBias | kEpiGelu, 2.0 * g.M * g.N * (3.0 * nm));
    // the pad rows of h1p are zero and stay zero as planes
    if (c2p) to_planes(ws_.h1p, ws_.h1pp, h1p_plane, long(batch) * (T0 + 2), d, &sc_conv2_.a, 0);
  }
  // conv2 (stride 2) + GELU + positional embedding: output t reads padded rows 2t..2t+2
  if (c2p) {
    PlaneGemmArgs g;
    g.A = ws_.h1pp; g.a_plane = h1p_plane; g.a_rpb = T; g.a_bs = long(T0 + 2) * d; g.lda = 2 * d;
    g.W = conv2_p_.w; g.bias = conv2_b; g.pos = enc_pos; g.pos_period = T;
    g.C = ws_.x; g.ldc = d; g.M = M; g.N = d; g.K = 3 * d;
    if (gemm_on_planes(sc_layers_[0].qkv)) fuse_ln(g, enc_blocks_[0].attn_ln_g, enc_blocks_[0].attn_ln_b, sc_layers_[0].qkv.a);
    ln_done = plane_gemm(g, sc_conv2_, kEpiBias | kEpiGelu | kEpiPos, 2.0 * g.M * g.N * g.K);
  } else {
    GemmArgs g;
    g.A = ws_.h1p; g.a_rpb = T; g.a_bs = long(T0 + 2) * d; g.lda = 2 * d;
    g.W = conv2_w; g.bias = conv2_b; g.pos = enc_pos; g.pos_period = T;
    g.C = ws_.x; g.ldc = d; g.M = M; g.N = d; g.K = 3 * d;
    alt_gemm(g, sc_conv2_, kEpiBias | kEpiGelu | kEpiPos, 2.0 * g.M * g.N * g.K);
  }
  constexpr float kQScale = 0.125f * 1.44269504088896340736f;  // d_head^-1/2 * log2(e): softmax as exp2
  auto layernorm_for = [&](bool planes, float scale, const float* g, const float* b) {
    if (ln_done) {  // the producing GEMM's epilogue has written these planes
      ln_done = false;
      return;
    }
    kt_begin(kKcLayerNorm, 0, 2.0 * M * d * 4);
    if (planes) {
      launch_layernorm_planes(ws_.x, lnp, ln_plane, scale, nullptr, g, b, M, d, stream_);
    } else {
      launch_layernorm(ws_.x, ws_.ln, g, b, M, d, stream_);
    }
    kt_end();
  };
  for (int l = 0; l < c.n_audio_layer; ++l) {
    const BlockWeights& w = enc_blocks_[l];
    const EncLayerPlanes& wp = enc_planes_[l];
    const EncLayerScales& sc = sc_layers_[l];
    const bool qp = gemm_on_planes(sc.qkv), op = gemm_on_planes(sc.out), f1p = gemm_on_planes(sc.fc1), f2p = gemm_on_planes(sc.fc2);
    // the plane attention writes planes, so its consumer (the out-projection) must take them
    const bool ap = gemm_variant < 0 && attn_variant == 4 && sc.attn_f16_ok && op;
    const int attn_alt = attn_variant == 4 ? 1 : int(attn_variant);  // fall-back form: three bf16 planes

    layernorm_for(qp, sc.qkv.a, w.attn_ln_g, w.attn_ln_b);
    const float qkv_scales[3] = {kQScale * sc.q, sc.k, sc.v};
    const unsigned short* qkv_src = qkvp;  // the plane attention's operand
    long qkv_src_plane = qkv_plane;
    if (qp) {
      PlaneGemmArgs q;  // q | k | v: the attention kernel's operand planes, or fp32 for the fall-back attention
      q.A = lnp; q.a_plane = ln_plane; q.lda = d; q.W = wp.qkv.w; q.bias = w.attn.bqkv;
      if (ap) {
        q.P = qkvp; q.p_plane = qkv_plane; q.seg = d;
        q.out_scale[0] = qkv_scales[0]; q.out_scale[1] = qkv_scales[1]; q.out_scale[2] = qkv_scales[2];
      } else {
        q.C = ws_.qkv;
      }
      q.ldc = 3 * d; q.M = M; q.N = 3 * d; q.K = d;
      plane_gemm(q, sc.qkv, kEpiBias, 2.0 * q.M * q.N * q.K);
    } else {
      GemmArgs q;
      q.A = ws_.ln; q.lda = d; q.W = w.attn.wqkv; q.bias = w.attn.bqkv; q.C = ws_.qkv; q.ldc = 3 * d;
      q.M = M; q.N = 3 * d; q.K = d;
      alt_gemm(q, sc.qkv, kEpiBias, 2.0 * q.M * q.N * q.K);
      if (ap) {
        to_planes(ws_.qkv, ws_.cvt, cvt_plane, M, 3 * d, qkv_scales, d);
        qkv_src = ws_.cvt; qkv_src_plane = cvt_plane;
      }
    }
    const unsigned short* att_src = attp;  // the plane out-projection's operand
    long att_src_plane = ln_plane;
    if (ap) {
      kt_begin(kKcEncAttn, 4.0 * batch * c.n_audio_head * double(T) * T * 64, 0);
      launch_encoder_attention_planes(qkv_src, qkv_src_plane, attp, ln_plane, batch, T, c.n_audio_head, sc.q, sc.k, sc.v,
                                      sc.out.a, stream_);
      kt_end();
    } else {
      kt_begin(kKcEncAttnAlt, 4.0 * batch * c.n_audio_head * double(T) * T * 64, 0);
      launch_encoder_attention(ws_.qkv, ws_.att, batch, T, c.n_audio_head, attn_alt, stream_);
      kt_end();
      if (op) {
        to_planes(ws_.att, ws_.cvt, cvt_plane, M, d, &sc.out.a, 0);
        att_src = ws_.cvt; att_src_plane = cvt_plane;
      }
    }
    if (op) {
      PlaneGemmArgs o;
      o.A = att_src; o.a_plane = att_src_plane; o.lda = d; o.W = wp.out.w; o.bias = w.attn.bo;
      o.C = ws_.x; o.R = ws_.x; o.ldc = d; o.M = M; o.N = d; o.K = d;
      if (f1p) fuse_ln(o, w.mlp_ln_g, w.mlp_ln_b, sc.fc1.a);
      ln_done = plane_gemm(o, sc.out, kEpiBias | kEpiResidual, 2.0 * o.M * o.N * o.K);
    } else {
      GemmArgs o;
      o.A = ws_.att; o.lda = d; o.W = w.attn.wo; o.bias = w.attn.bo; o.C = ws_.x; o.R = ws_.x; o.ldc = d;
      o.M = M; o.N = d; o.K = d;
      alt_gemm(o, sc.out, kEpiBias | kEpiResidual, 2.0 * o.M * o.N * o.K);
    }
    layernorm_for(f1p, sc.fc1.a, w.mlp_ln_g, w.mlp_ln_b);
    const unsigned short* hid_src = hidp;  // the plane fc2's operand
    long hid_src_plane = hid_plane;
    if (f1p) {
      PlaneGemmArgs f1;
      f1.A = lnp; f1.a_plane = ln_plane; f1.lda = d; f1.W = wp.fc1.w; f1.bias = w.b1;
      if (f2p) {
        f1.P = hidp; f1.p_plane = hid_plane; f1.out_scale[0] = sc.fc2.a;
      } else {
        f1.C = ws_.hid;
      }
      f1.ldc = 4 * d; f1.M = M; f1.N = 4 * d; f1.K = d;
      plane_gemm(f1, sc.fc1, kEpiBias | kEpiGelu, 2.0 * f1.M * f1.N * f1.K);
    } else {
      GemmArgs f1;
      f1.A = ws_.ln; f1.lda = d; f1.W = w.w1; f1.bias = w.b1; f1.C = ws_.hid; f1.ldc = 4 * d;
      f1.M = M; f1.N = 4 * d; f1.K = d;
      alt_gemm(f1, sc.fc1, kEpiBias | kEpiGelu, 2.0 * f1.M * f1.N * f1.K);
      if (f2p) {
        to_planes(ws_.hid, ws_.cvt, cvt_plane, M, 4 * d, &sc.fc2.a, 0);
        hid_src = ws_.cvt; hid_src_plane = cvt_plane;
      }
    }
    if (f2p) {
      PlaneGemmArgs f2;
      f2.A = hid_src; f2.a_plane = hid_src_plane; f2.lda = 4 * d; f2.W = wp.fc2.w; f2.bias = w.b2;
      f2.C = ws_.x; f2.R = ws_.x; f2.ldc = d; f2.M = M; f2.N = d; f2.K = 4 * d;
      if (l + 1 < c.n_audio_layer) {
        if (gemm_on_planes(sc_layers_[l + 1].qkv)) fuse_ln(f2, enc_blocks_[l + 1].attn_ln_g, enc_blocks_[l + 1].attn_ln_b, sc_layers_[l + 1].qkv.a);
      } else if (gemm_on_planes(sc_cross_kv_)) {  // the encoder's final LayerNorm: also the API's fp32 enc_out and the non-finite flag
        fuse_ln(f2, enc_ln_post_g, enc_ln_post_b, sc_cross_kv_.a);
        f2.ln_y32 = ws_.enc_out; f2.nonfinite = slot.d_flag;
        if (absorb && f2.ln_P) {  // absorbed cross-attention: these planes ARE what the decoder streams (no cross-KV projection)
          f2.ln_P = slot.e_planes; f2.ln_plane = e_plane;
        }
        if (!f2.ln_P) f2.ln_y32 = nullptr, f2.nonfinite = nullptr;
      }
      ln_done = plane_gemm(f2, sc.fc2, kEpiBias | kEpiResidual, 2.0 * f2.M * f2.N * f2.K);
    } else {
      GemmArgs f2;
      f2.A = ws_.hid; f2.lda = 4 * d; f2.W = w.w2; f2.bias = w.b2; f2.C = ws_.x; f2.R = ws_.x; f2.ldc = d;
      f2.M = M; f2.N = d; f2.K = 4 * d;
      alt_gemm(f2, sc.fc2, kEpiBias | kEpiResidual, 2.0 * f2.M * f2.N * f2.K);
    }
  }
  const bool kp = gemm_on_planes(sc_cross_kv_);
  if (ln_done) {
    ln_done = false;
  } else {
    kt_begin(kKcLayerNorm, 0, 2.0 * M * d * 4);
    if (kp) {
      launch_layernorm_planes(ws_.x, absorb ? slot.e_planes : lnp, absorb ? e_plane : ln_plane, sc_cross_kv_.a, ws_.enc_out,
                              enc_ln_post_g, enc_ln_post_b, M, d, stream_, slot.d_flag);
    } else {
      launch_layernorm(ws_.x, ws_.enc_out, enc_ln_post_g, enc_ln_post_b, M, d, stream_, slot.d_flag);
    }
    kt_end();
  }
  HIPCHK(hipMemcpyAsync(slot.h_flag, slot.d_flag, sizeof(int), hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipEventRecord(slot.enc_mid, stream_));
  // cross-attention K/V of every decoder layer, projected once per clip into the persistent cache
  // [layer][k|v][clip][head][t][64] (the reference recomputes them inside every decoder Invoke(), whisper.cpp:375) —
  // unless the decoder runs the absorbed form, which needs the planes of the encoder output and nothing else
  slot.absorbed = absorb;
  if (absorb) {
  } else if (kp) {
    PlaneGemmArgs g;
    need_cross_kv(slot);
    g.A = lnp; g.a_plane = ln_plane; g.lda = d; g.W = cross_kv_p_.w; g.bias = cross_kv_b;
    g.C = slot.cross_kv; g.M = M; g.N = c.n_text_layer * 2 * d; g.K = d;
    g.c_rpb = T; g.kv_batch = batch; g.kv_heads = c.n_text_head; g.kv_dmodel = d;
    plane_gemm(g, sc_cross_kv_, kEpiBias | kEpiKvLayout, 2.0 * g.M * g.N * g.K);
  } else {
    GemmArgs g;
    need_cross_kv(slot);
    g.A = ws_.enc_out; g.lda = d; g.W = cross_kv_w; g.bias = cross_kv_b; g.C = slot.cross_kv;
    g.M = M; g.N = c.n_text_layer * 2 * d; g.K = d;
    g.c_rpb = T; g.kv_batch = batch; g.kv_heads = c.n_text_head; g.kv_dmodel = d;
    alt_gemm(g, sc_cross_kv_, kEpiBias | kEpiKvLayout, 2.0 * g.M * g.N * g.K);
  }
  HIPCHK(hipEventRecord(slot.enc_done, stream_));
  slot.used = true;
  last_enc_slot_ = enc_slot_;
  enc_slot_ = (enc_slot_ + 1) % kSlots;
}

// bf16 storage mode (option "bf16"): the same graph with every contraction operand stored as ONE bf16 matrix
// (k_gemm_bf16.hip, encoder_attention_planes<true>); the residual stream x and the API's enc_out stay fp32, the
// cross-KV cache of the slot is written as bf16 (half the bytes of the default mode, same layout).
void Engine::encode_enqueue_bf16(const float* d_mel, int batch) {
  const wtw::Dims& c = dims_;
  const int T0 = mel_frames(), T = c.n_audio_ctx, d = c.n_audio_state, M = batch * T, nm = c.n_mels;
  Slot& slot = slots_[enc_slot_];
  if (slot.used) HIPCHK(hipStreamWaitEvent(stream_, slot.dec_done, 0));
  slot.kt_cls.clear();
  slot.kt_flops.clear();
  slot.kt_bytes.clear();
  slot.batch = batch;
  HIPCHK(hipMemsetAsync(slot.d_flag, 0, sizeof(int), stream_));
  HIPCHK(hipEventRecord(slot.enc_begin, stream_));
  unsigned short* const lnp = reinterpret_cast<unsigned short*>(ws_.ln);
  unsigned short* const qkvp = reinterpret_cast<unsigned short*>(ws_.qkv);
  unsigned short* const attp = reinterpret_cast<unsigned short*>(ws_.att);
  unsigned short* const hidp = reinterpret_cast<unsigned short*>(ws_.hid);

  kt_begin(kKcTranspose, 0, 1.5 * batch * nm * T0 * 4);
  launch_mel_transpose_planes(d_mel, ws_.melTp, 0, 1.0f, batch, nm, T0, nm, stream_, true);
  kt_end();
  {
    PlaneGemmArgs g;  // conv1 + GELU: rows (clip, t) read melT rows t..t+2 (input t-1..t+1)
    g.A = ws_.melTp; g.a_rpb = T0; g.a_bs = long(T0 + 2) * nm; g.lda = nm;
    g.W = bf_.conv1; g.bias = conv1_b;
    g.P = ws_.h1pp + d;  // row t lands at padded row t + 1
    g.c_rpb = T0; g.c_bs = long(T0 + 2) * d; g.ldc = d;
    g.M = batch * T0; g.N = d; g.K = bf_.conv1_kpad;
    kt_begin(kKcGemm, 2.0 * g.M * g.N * (3.0 * nm), 0);
    launch_gemm_bf16_planes(g, kEpiBias | kEpiGelu, stream_);
    kt_end();
  }
  bool have_ln = false;  // the LayerNorm plane of the coming layer is already in lnp (written by a GEMM epilogue)
  {
    PlaneGemmArgs g;  // conv2 (stride 2) + GELU + positional embedding
    g.A = ws_.h1pp; g.a_rpb = T; g.a_bs = long(T0 + 2) * d; g.lda = 2 * d;
    g.W = bf_.conv2; g.bias = conv2_b; g.pos = enc_pos; g.pos_period = T;
    g.C = ws_.x; g.ldc = d;
    g.M = M; g.N = d; g.K = 3 * d;
    // (round 4) the LayerNorm in front of a layer's qkv / fc1 GEMM and ln_post are written by the epilogue of the GEMM
    // that finishes the residual rows (k_gemm_bf16.hip, whole-row tiles): no separate launch re-reads the stream
    if (c.n_audio_layer > 0) g.ln_g = enc_blocks_[0].attn_ln_g, g.ln_b = enc_blocks_[0].attn_ln_b, g.ln_P = lnp;
    kt_begin(kKcGemm, 2.0 * g.M * g.N * g.K, 0);
    have_ln = launch_gemm_bf16_planes(g, kEpiBias | kEpiGelu | kEpiPos, stream_);
    kt_end();
  }
  const bool absorb = absorb_for(batch);  // the decoder streams the encoder output itself (one bf16 plane per slot): no cross-KV GEMM
  bool have_post = false;
  for (int l = 0; l < c.n_audio_layer; ++l) {
    const BlockWeights& w = enc_blocks_[l];
    const Bf16Encoder::Layer& wb = bf_.layers[l];
    if (!have_ln) {
      kt_begin(kKcLayerNorm, 0, 1.5 * M * d * 4);
      launch_layernorm_planes(ws_.x, lnp, 0, 1.0f, nullptr, w.attn_ln_g, w.attn_ln_b, M, d, stream_, nullptr, true);
      kt_end();
    }
    PlaneGemmArgs q;
    q.A = lnp; q.lda = d; q.W = wb.qkv; q.bias = w.attn.bqkv;
    q.P = qkvp; q.ldc = 3 * d; q.M = M; q.N = 3 * d; q.K = d;
    kt_begin(kKcGemm, 2.0 * q.M * q.N * q.K, 0);
    launch_gemm_bf16_planes(q, kEpiBias, stream_);
    kt_end();
    kt_begin(kKcEncAttn, 4.0 * batch * c.n_audio_head * double(T) * T * 64, 0);
    launch_encoder_attention_bf16(qkvp, attp, batch, T, c.n_audio_head, stream_);
    kt_end();
    PlaneGemmArgs o;
    o.A = attp; o.lda = d; o.W = wb.out; o.bias = w.attn.bo;
    o.C = ws_.x; o.R = ws_.x; o.ldc = d; o.M = M; o.N = d; o.K = d;
    o.ln_g = w.mlp_ln_g; o.ln_b = w.mlp_ln_b; o.ln_P = lnp;
    kt_begin(kKcGemm, 2.0 * o.M * o.N * o.K, 0);
    const bool have_mlp_ln = launch_gemm_bf16_planes(o, kEpiBias | kEpiResidual, stream_);
    kt_end();
    if (!have_mlp_ln) {
      kt_begin(kKcLayerNorm, 0, 1.5 * M * d * 4);
      launch_layernorm_planes(ws_.x, lnp, 0, 1.0f, nullptr, w.mlp_ln_g, w.mlp_ln_b, M, d, stream_, nullptr, true);
      kt_end();
    }
    PlaneGemmArgs f1;
    f1.A = lnp; f1.lda = d; f1.W = wb.fc1; f1.bias = w.b1;
    f1.P = hidp; f1.ldc = 4 * d; f1.M = M; f1.N = 4 * d; f1.K = d;
    kt_begin(kKcGemm, 2.0 * f1.M * f1.N * f1.K, 0);
    launch_gemm_bf16_planes(f1, kEpiBias | kEpiGelu, stream_);
    kt_end();
    PlaneGemmArgs f2;
    f2.A = hidp; f2.lda = 4 * d; f2.W = wb.fc2; f2.bias = w.b2;
    f2.C = ws_.x; f2.R = ws_.x; f2.ldc = d; f2.M = M; f2.N = d; f2.K = 4 * d;
    const bool last = l + 1 == c.n_audio_layer;
    if (!last) {
      f2.ln_g = enc_blocks_[l + 1].attn_ln_g; f2.ln_b = enc_blocks_[l + 1].attn_ln_b; f2.ln_P = lnp;
    } else {  // ln_post: the bf16 plane (the decoder's stream in the absorbed form), the fp32 enc_out and the non-finite flag
      f2.ln_g = enc_ln_post_g; f2.ln_b = enc_ln_post_b; f2.ln_P = absorb ? slot.e_planes : lnp;
      f2.ln_y32 = ws_.enc_out; f2.nonfinite = slot.d_flag;
    }
    kt_begin(kKcGemm, 2.0 * f2.M * f2.N * f2.K, 0);
    const bool fused = launch_gemm_bf16_planes(f2, kEpiBias | kEpiResidual, stream_);
    kt_end();
    (last ? have_post : have_ln) = fused;
  }
  kt_begin(kKcLayerNorm, 0, 2.5 * M * d * 4);
  if (!have_post) {
    launch_layernorm_planes(ws_.x, absorb ? slot.e_planes : lnp, 0, 1.0f, ws_.enc_out, enc_ln_post_g, enc_ln_post_b, M, d, stream_,
                            slot.d_flag, true);
  }
  HIPCHK(hipMemcpyAsync(slot.h_flag, slot.d_flag, sizeof(int), hipMemcpyDeviceToHost, stream_));
  kt_end();
  HIPCHK(hipEventRecord(slot.enc_mid, stream_));
  slot.absorbed = absorb;
  if (!absorb) {
    PlaneGemmArgs g;  // cross-attention K/V of every decoder layer into the slot's cache, as bf16
    need_cross_kv(slot);
    g.A = lnp; g.lda = d; g.W = bf_.cross_kv; g.bias = cross_kv_b;
    g.P = reinterpret_cast<unsigned short*>(slot.cross_kv); g.M = M; g.N = c.n_text_layer * 2 * d; g.K = d;
    g.c_rpb = T; g.kv_batch = batch; g.kv_heads = c.n_text_head; g.kv_dmodel = d;
    kt_begin(kKcGemm, 2.0 * g.M * g.N * g.K, 0);
    launch_gemm_bf16_planes(g, kEpiBias | kEpiKvLayout, stream_);
    kt_end();
  }
  HIPCHK(hipEventRecord(slot.enc_done, stream_));
  slot.used = true;
  last_enc_slot_ = enc_slot_;
  enc_slot_ = (enc_slot_ + 1) % kSlots;
}

void Engine::set_bf16(bool on) {
  require_idle();
  if (on) ensure_bf16_weights();
  if ((bf16 != 0) != on && ws_.batch > 0 && dims_.n_audio_state != 0) {
    // the zero pad rows of the two convolution inputs sit at different offsets in the two layouts
    const size_t B = ws_.batch, T0 = mel_frames();
    sync();
    HIPCHK(hipMemset(ws_.melTp, 0, (B * (T0 + 2) * dims_.n_mels + 256) * sizeof(float)));
    HIPCHK(hipMemset(ws_.h1pp, 0, (B * (T0 + 2) * dims_.n_audio_state + 256) * sizeof(float)));
    for (DecWorkspace& dw : dws_) {
      if (dw.self_kv) HIPCHK(hipMemset(dw.self_kv, 0, size_t(dims_.n_text_layer) * 2 * B * self_cap_ * dims_.n_text_state * sizeof(float)));
    }
  }
  bf16 = on ? 1 : 0;
}

// ------------------------------------------------------------ decoder ---

void Engine::decode(int batch, int64_t* ids, int32_t* n_ids, float* logits_host,
                    int logits_steps_cap) {
  require_idle();
  decode_enqueue(batch, last_enc_slot_, logits_host, logits_steps_cap);
  decode_collect(last_enc_slot_, ids, n_ids);
}

void Engine::flush_pending() {
  if (pending_.empty()) return;
  const int p = pending_.front(), n = int(pending_.size());
  pending_.clear();
  decode_enqueue(slots_[p].batch, p, nullptr, 0, n, true);  // the rest of the group never came: a shorter chain
}

// decoder side of a pipelined submit: alone, or together with the neighbouring submits' batches (dec_pair, dec_group)
int Engine::group_of(int batch) const {
  // (WT_PAIR_MAX_BATCH: largest batch that is grouped — 32 until round 4, when a chain's rows were limited to 64)
  static const int pair_max = [] {
    const char* v = getenv("WT_PAIR_MAX_BATCH");
    return v ? atoi(v) : 32;
  }();
  if (dec_pair == 0 || !absorb_active() || batch > pair_max) return 1;
  const int g = int(std::min<long>(std::max<long>(dec_group, 2), 4));
  return std::max(1, std::min(g, kDecRowsMax / batch));
}

// Throughput form: `group` consecutive batches share one decoder chain.  Latency form — the last `last_batches` submits
// of a job (option, counted down here): a chain per batch, on spare decoder streams when the probe found any, so that the
// pipeline drains in one short chain instead of a long shared one behind two others (measured on the driver's 20-step
// command: 16.7 ms from the last encoder pass to the last token with the paired form).
void Engine::submit_decoder(int batch, int s) {
  const bool tail = last_batches > 0;
  if (tail) --last_batches;
  const int group = group_of(batch);
  if (!tail && group > 1 && ws_.batch >= group * batch) {
    if (!pending_.empty() && (slots_[pending_.front()].batch != batch || (pending_.back() + 1) % kSlots != s)) flush_pending();
    pending_.push_back(s);
    if (int(pending_.size()) == group) {
      const int a = pending_.front();
      pending_.clear();
      decode_enqueue(batch, a, nullptr, 0, group, true);
    }
  } else {
    flush_pending();
    int spare = -1;
    // The very last batch decodes on the encoder's own stream, right behind its encoder pass: that hardware queue has
    // nothing else to do once the last encoder pass is through, and the decoder streams are still busy with earlier
    // chains.  (Submitting more batches after "the last" is allowed: their encoder passes queue behind that chain.)
    // The batches before it use a spare decoder stream when the probe found one.
    if (tail && last_batches == 0) spare = kEncAsDec;
    else if (tail && n_spare_streams_ > 0) spare = n_dec_streams_ + int(last_batches % n_spare_streams_);
    decode_enqueue(batch, s, nullptr, 0, 1, true, spare);
  }
  inflight_.push_back(s);
}

void Engine::submit(const float* d_mel, int batch) {
  if (int(inflight_.size()) >= kSlots) throw Error(1, "pipeline is full (24 batches in flight): collect() first");
  if (batch > 64) throw Error(1, "decoder batches are limited to 64 clips per call");
  select_stream(true);
  if (inflight_.empty()) ensure_batch(group_of(batch) * batch);  // the group's decoder rows; never grown in flight
  encode_enqueue(d_mel, batch);
  submit_decoder(batch, last_enc_slot_);
}

void Engine::submit_pcm(const float* d_pcm, int batch) {
  if (int(inflight_.size()) >= kSlots) throw Error(1, "pipeline is full (24 batches in flight): collect() first");
  if (batch > 64) throw Error(1, "decoder batches are limited to 64 clips per call");
  select_stream(true);
  if (inflight_.empty()) ensure_batch(group_of(batch) * batch);
  // one staging mel buffer: the front end of batch k+1 follows the encoder of batch k on the same stream
  float* d_mel = staging_mel(batch);
  logmel(d_pcm, batch, d_mel);
  encode_enqueue(d_mel, batch);
  submit_decoder(batch, last_enc_slot_);
}

void Engine::collect(int64_t* ids, int32_t* n_ids) {
  if (inflight_.empty()) throw Error(1, "collect() without a submitted batch");
  const int slot = inflight_.front();
  if (std::find(pending_.begin(), pending_.end(), slot) != pending_.end()) flush_pending();  // its group never filled: decode now
  inflight_.erase(inflight_.begin());
  decode_collect(slot, ids, n_ids);
}

std::vector<long long> Engine::prompt() const {
  if (!prompt_override.empty()) return prompt_override;
  // EncDec (whisper.cpp:327-339): [sot, 50259 + language, transcribe, notimestamps] whatever `multilingual` is
  // (the ids come from the Vocab, which transform_vocab_multilingual shifted or not, whisper.cpp:218-226).
  // Monolith: the forced decoder ids HF generate() applies inside the reference's single graph
  // (export/generate.py:24-30): English-only checkpoints [sot, notimestamps] — the head of kGoldenGeneratedIDs,
  // whisper.h:27-32 — multilingual ones [sot, language, transcribe, notimestamps].
  if (monolith_ && !multilingual_) return {vocab_.token_sot, vocab_.token_not};
  return {vocab_.token_sot, 50259 + language, vocab_.token_transcribe, vocab_.token_not};
}

void Engine::decode_enqueue(int batch, int slot_idx, float* logits_host, int logits_steps_cap, int group, bool pipelined,
                            int stream_override) {
  const bool paired = group > 1;
  const int per = batch;            // clips per encoder batch
  if (group < 1 || group > 4) throw Error(kErrInvalidArg, "decoder chains take one to four batches");
  batch = group * per;              // the chain decodes them all: rows / clips below count the group
  if (batch > (paired ? kDecRowsMax : 64)) throw Error(1, "decoder batches are limited to 64 clips per call");  // before any stream operation
  for (int j = 0; paired && j < group; ++j) {
    if (!slots_[(slot_idx + j) % kSlots].absorbed || logits_host) throw Error(kErrInvalidArg, "decoder groups: the absorbed form only");
  }
  ensure_batch(batch);
  Slot& slot = slots_[slot_idx];
  // fixed slot -> stream map (few captured graphs); pair leaders are the even slots, so a pair counts as one
  // (stream_override: a spare stream + workspace for a batch decoded in the latency form, submit_decoder)
  auto dec_of = [&](int si) { return stream_override >= 0 ? stream_override : (si / group) % n_dec_streams_; };
  slot.dec = dec_of(slot_idx);
  slot.pair_leader = -1;
  DecWorkspace& dw = dws_[slot.dec];
  hipStream_t const stream_ = dec_stream_at(slot.dec);  // everything below runs on this decoder stream
  HIPCHK(hipStreamWaitEvent(stream_, slot.enc_done, 0));
  for (int j = 1; j < group; ++j) HIPCHK(hipStreamWaitEvent(stream_, slots_[(slot_idx + j) % kSlots].enc_done, 0));
  HIPCHK(hipEventRecord(slot.dec_begin, stream_));
  long long* const h_ids_ = slot.h_ids;
  int* const h_n_ = slot.h_n;
  const wtw::Dims& c = dims_;
  const int d = c.n_text_state, T = c.n_audio_ctx, H = c.n_text_head, V = c.n_vocab;
  const std::vector<long long> prompt = this->prompt();
  const int n_prompt = int(prompt.size()), stride = 32;
  for (long long id : prompt) {
    if (id < 0 || id >= V) throw Error(1, "prompt token id outside the model's vocabulary");
  }
  const int max_pos = int(std::min<long>(std::max<long>(max_tokens, n_prompt), 31));
  const bool forced = !forced_ids.empty();
  if (forced && forced_ids.size() != size_t(batch) * stride) {
    throw Error(kErrInvalidArg, "forced ids are set for another number of clips than this decode has");
  }
  for (long long id : forced_ids) {
    if (id < 0 || id >= V) throw Error(kErrInvalidArg, "forced token id outside the model's vocabulary");
  }
  for (int b = 0; b < batch; ++b) {
    for (int i = 0; i < stride; ++i) {
      h_ids_[size_t(b) * stride + i] = forced ? forced_ids[size_t(b) * stride + i] : (i < n_prompt ? prompt[i] : 0);
    }
    h_n_[b] = n_prompt;
  }
  int chunks = int(cross_chunks);  // 1, 2, 4 or 8 (wt_engine_set_option), 0 = by batch size
  if (chunks == 0) {
    chunks = 1;
    while (chunks < 8 && batch * H * chunks < 192) chunks *= 2;
  }
  // key chunks of the absorbed form: clips x chunks ~ 256 blocks, a chunk a whole number of 32-key tiles
  const bool absorbed = slot.absorbed;  // the form this slot's encoder pass prepared (the same for every slot of a graph set)
  // (option abs_chunks; 0 = 256 / clips: one block per CU when the decoder has the chip; pipelined 128 / clips — fewer,
  // longer blocks: a block costs ~9 us before its first tile, and the decoders share the CUs the encoder leaves)
  // (bf16 storage mode: half the bytes per key row, so half as many blocks again: 122.5 k against 120.5 k on configs[3])
  const int abs_blocks = pipelined ? (bf16 ? 64 : 128) : 256;
  int n_abs = abs_chunks > 0 ? int(abs_chunks) : std::min(16, std::max(1, (abs_blocks + batch - 1) / batch));
  n_abs = std::min(n_abs, (T + 31) / 32);
  int steps = 0;
  // WT_DEC_KERNEL_TIMERS=1 (diagnostics, eager launches only): event pairs around every decoder launch
  static const bool dec_timers = getenv("WT_DEC_KERNEL_TIMERS") != nullptr;
  std::vector<hipEvent_t>& dt_ev = slot.dt_events;
  std::vector<int>& dt_cls = slot.dt_cls;
  size_t dt_used = 0;
  bool dt_on = false;
#define DT(CLS, CALL)                                                         \
  do {                                                                        \
    if (dt_on) {                                                              \
      while (dt_ev.size() < dt_used + 2) {                                    \
        hipEvent_t e_;                                                        \
        HIPCHK(hipEventCreate(&e_));                                          \
        dt_ev.push_back(e_);                                                  \
      }                                                                       \
      HIPCHK(hipEventRecord(dt_ev[dt_used], stream_));                        \
      CALL;                                                                   \
      HIPCHK(hipEventRecord(dt_ev[dt_used + 1], stream_));                    \
      dt_cls.push_back(CLS);                                                  \
      dt_used += 2;                                                           \
    } else {                                                                  \
      CALL;                                                                   \
    }                                                                         \
  } while (0)
  (void)dw;
  auto enqueue_all = [&](int si) {
    Slot& slot = slots_[si];
    const Slot* const member[4] = {&slot, &slots_[(si + 1) % kSlots], &slots_[(si + 2) % kSlots], &slots_[(si + 3) % kSlots]};  // the group's batches
    DecWorkspace& dw = dws_[dec_of(si)];
    hipStream_t const stream_ = dec_stream_at(dec_of(si));
    long long* const h_ids_ = slot.h_ids;
    int* const h_n_ = slot.h_n;
    steps = 0;
    HIPCHK(hipMemcpyAsync(dw.ids, h_ids_, size_t(batch) * stride * sizeof(long long),
                          hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(dw.n_ids, h_n_, size_t(batch) * sizeof(int), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemsetAsync(dw.finished, 0, size_t(batch) * sizeof(int), stream_));

    const bool bf = bf16 != 0;  // bf16 storage mode: bf16 weights and caches (element size 2 in the cache offsets)
    const size_t kv_slab = size_t(batch) * T * d;  // one (layer, k|v) slab of the cross cache
    const size_t self_slab = size_t(batch) * self_cap_ * d;
    auto cache_at = [&](float* base, size_t elems) -> void* {
      return bf ? static_cast<void*>(reinterpret_cast<unsigned short*>(base) + elems) : static_cast<void*>(base + elems);
    };
    float* const x = dw.xd;  // residual stream [rows][d], updated in place by the residual GEMMs
    // fc2 (K = 4 d) runs over twice the blocks when its K splits evenly over 2 x 8 waves x 16
    const bool split = fc2_ksplit == 2 && (4 * d) % 256 == 0;
    // Passes.  The prompt positions of every clip go through the layers TOGETHER when they fit one pass (rows =
    // positions x clips <= 128, at most 4 positions: causal self-attention inside the pass, one sweep of the
    // cross-KV cache for all of them); the reference feeds the same prefix to its graph at once, whisper.cpp:367-375.
    // Every later position is one pass of `batch` rows.
    // (a pair's 64 clips, or a 64-clip batch: the four prompt positions go two and two — 28 passes instead of 30)
    const int np_max = std::max(1, std::min(4, kDecRowsMax / batch));
    const int prompt_end = std::min(n_prompt, max_pos);
    for (int pos0 = 0, np = 1; pos0 < max_pos; pos0 += np) {
      np = pos0 < prompt_end ? std::min(np_max, prompt_end - pos0) : 1;
      const int M = np * batch, last = pos0 + np - 1;
      for (int l = 0; l < c.n_text_layer; ++l) {
        const DecBlockWeights& w = bf ? dec_blocks_bf_[l] : dec_blocks_[l];
        DecGemmArgs q;  // LN + fused q|k|v projection (+ token/positional embedding at layer 0)
        q.bf16 = bf;
        q.Wt = w.wqkv.w; q.w_scale = w.wqkv.scale; q.N = 3 * d; q.K = d; q.B = batch; q.M = M;
        q.xin = x; q.ln_g = w.attn_ln_g; q.ln_b = w.attn_ln_b;
        if (l > 0 && split) {  // the previous layer's fc2 left x in two halves: sum them, block 0 completes x
          q.xin = dw.xb; q.xpart = dw.xpart; q.xout = x;
        }
        if (l == 0) {
          q.ids = dw.ids; q.ids_stride = stride; q.pos = pos0; q.tok_emb = tok_emb; q.pos_emb = dec_pos;
          q.n_vocab = V; q.xout = x;
        }
        q.bias = w.bqkv; q.Y = dw.qkvd; q.ldy = 3 * d;
        DT(0, launch_dec_gemm(q, kProLn, kDecBias, stream_));
        DT(1, launch_self_attention(dw.qkvd, cache_at(dw.self_kv, (size_t(l) * 2 + 0) * self_slab),
                                    cache_at(dw.self_kv, (size_t(l) * 2 + 1) * self_slab), self_cap_, pos0, np, dw.attd,
                                    batch, H, stream_, bf));
        DecGemmArgs o;  // x += attn . Wo^T + bo
        o.bf16 = bf;
        o.Wt = w.wo.w; o.w_scale = w.wo.scale; o.N = d; o.K = d; o.B = batch; o.M = M; o.X = dw.attd; o.ldx = d;
        o.bias = w.bo; o.R = x; o.Y = x; o.ldy = d;
        DT(2, launch_dec_gemm(o, kProNone, kDecResid, stream_));

        if (absorbed) {
          // Cross attention against the encoder output itself (k_cross_absorbed.hip): LN + absorbed query projection
          // q'_h = c0 Wk_h^T (Wq_h LN(x) + bq_h) for all heads in one GEMM, the matrix-core sweep of E per key chunk
          // (positions in groups of 16 / heads query columns), the chunk combine with the heads' value projections,
          // and the ordinary out-projection.
          DecGemmArgs qa;
          qa.Wt = w.wq_abs.w; qa.w_scale = w.wq_abs.scale; qa.N = H * d; qa.K = d; qa.B = batch; qa.M = M;
          qa.bf16 = bf;
          qa.xin = x; qa.ln_g = w.cross_ln_g; qa.ln_b = w.cross_ln_b; qa.bias = w.bq_abs; qa.Y = dw.qp; qa.ldy = H * d;
          DT(3, launch_dec_gemm(qa, kProLn, kDecBias, stream_));
          const int nq_max = cross_absorbed_max_nq(H);
          for (int p0 = 0; p0 < np; p0 += nq_max) {
            CrossAbsorbedArgs ca;
            ca.qp = dw.qp; ca.e = slot.e_planes; ca.e_plane = long(ws_.batch) * T * d; ca.e_scale = sc_cross_kv_.a;
            ca.bf16 = bf;
            if (paired) {
              ca.split = per;
              ca.e2 = member[1]->e_planes;
              if (group > 2) ca.e3 = member[2]->e_planes;
              if (group > 3) ca.e4 = member[3]->e_planes;
            }
            ca.ws = dw.abs_ws; ca.batch = batch; ca.heads = H; ca.d_model = d; ca.T = T; ca.chunks = n_abs;
            ca.nq = std::min(nq_max, np - p0); ca.p0 = p0;
            DT(4, launch_cross_absorbed(ca, stream_));
          }
          DT(8, launch_cross_absorbed_combine(dw.abs_ws, w.cross_wv_t, w.cross_bv, dw.cabs, M, H, n_abs, d, stream_));
          DecGemmArgs co;  // x += o . Wco^T + bco
          co.bf16 = bf;
          co.Wt = w.cross_wo.w; co.w_scale = w.cross_wo.scale; co.N = d; co.K = d; co.B = batch; co.M = M;
          co.X = dw.cabs; co.ldx = d; co.bias = w.cross_bo; co.R = x; co.Y = x; co.ldy = d;
          DT(5, launch_dec_gemm(co, kProNone, kDecResid, stream_));
        } else {
        CrossAttnArgs ca;  // LN + query projection + attention over the cached encoder keys, per key chunk
        ca.x = x; ca.ln_g = w.cross_ln_g; ca.ln_b = w.cross_ln_b; ca.wq_t = w.cross_wq_t; ca.bq = w.cross_bq;
        ca.kc = cache_at(slot.cross_kv, (size_t(l) * 2 + 0) * kv_slab); ca.vc = cache_at(slot.cross_kv, (size_t(l) * 2 + 1) * kv_slab);
        ca.bf16 = bf;
        ca.ws = dw.cross_ws; ca.batch = batch; ca.heads = H; ca.T = T; ca.chunks = chunks; ca.nq = np;
        DT(4, launch_cross_attention(ca, stream_));
        DecGemmArgs co;  // x += combine(chunks) . Wco^T + bco
        co.bf16 = bf;
        co.Wt = w.cross_wo.w; co.w_scale = w.cross_wo.scale; co.N = d; co.K = d; co.B = batch; co.M = M;
        co.cross_ws = dw.cross_ws; co.heads = H; co.chunks = chunks;
        co.bias = w.cross_bo; co.R = x; co.Y = x; co.ldy = d;
        DT(5, launch_dec_gemm(co, kProCombine, kDecResid, stream_));
        }

        DecGemmArgs f1;  // LN + fc1 + GELU
        f1.bf16 = bf;
        f1.Wt = w.w1.w; f1.w_scale = w.w1.scale; f1.N = 4 * d; f1.K = d; f1.B = batch; f1.M = M;
        f1.xin = x; f1.ln_g = w.mlp_ln_g; f1.ln_b = w.mlp_ln_b;
        f1.bias = w.b1; f1.Y = dw.hd; f1.ldy = 4 * d;
        DT(6, launch_dec_gemm(f1, kProLn, kDecBiasGelu, stream_));
        DecGemmArgs f2;  // x += h . W2^T + b2
        f2.bf16 = bf;
        f2.Wt = w.w2.w; f2.w_scale = w.w2.scale; f2.N = d; f2.K = 4 * d; f2.B = batch; f2.M = M; f2.X = dw.hd; f2.ldx = 4 * d;
        f2.bias = w.b2; f2.R = x; f2.Y = x; f2.ldy = d;
        if (split) {
          f2.Y = dw.xb; f2.ksplit = 2; f2.part = dw.xpart;
        }
        DT(7, launch_dec_gemm(f2, kProNone, kDecResid, stream_));
      }
      if (last >= n_prompt - 1) {
        // logits against the tied embedding + greedy argmax (whisper.cpp:379-399); only the
        // last position's rows exist here, the reference computes and drops the others
        const size_t off = size_t(np - 1) * batch * d;
        DecGemmArgs lg;  // final LayerNorm (of x, or of the two halves a K-split fc2 left) + logits + argmax records
        lg.bf16 = bf;
        lg.logits_blocks = pipelined ? 256 : 0;
        lg.Wt = (bf ? tok_emb_tiled_bf_ : tok_emb_tiled).w; lg.w_scale = (bf ? tok_emb_tiled_bf_ : tok_emb_tiled).scale; lg.N = V; lg.K = d; lg.B = batch;
        lg.xin = (split ? dw.xb : x) + off; lg.xpart = split ? dw.xpart + off : nullptr; lg.ln_g = dec_ln_g; lg.ln_b = dec_ln_b;
        lg.Y = logits_host ? dw.logits : nullptr; lg.ldy = V; lg.best = dw.best;
        DT(9, launch_dec_gemm(lg, kProLn, kDecLogits, stream_));
        if (logits_host && steps < logits_steps_cap) {
          HIPCHK(hipMemcpy2DAsync(logits_host + size_t(steps) * V, size_t(logits_steps_cap) * V * sizeof(float),
                                  dw.logits, size_t(V) * sizeof(float), size_t(V) * sizeof(float), batch,
                                  hipMemcpyDeviceToHost, stream_));
        }
        DT(10, launch_select_token(dw.best, (V + 31) / 32, dw.ids, stride, last, dw.n_ids, dw.finished,
                            vocab_.token_eot, int(stop_at_eot), batch, stream_, forced));
        ++steps;
      }
    }
    // Every slot receives its OWN clips' ids in its own pinned buffers: a pair's second batch used to be read out of
    // the leader's buffers, which the leader's next submit (collected first, reused first) could overwrite.
    for (int j = 0; j < group; ++j) {
      HIPCHK(hipMemcpyAsync(member[j]->h_ids, dw.ids + size_t(j) * per * stride, size_t(per) * stride * sizeof(long long),
                            hipMemcpyDeviceToHost, stream_));
      HIPCHK(hipMemcpyAsync(member[j]->h_n, dw.n_ids + size_t(j) * per, size_t(per) * sizeof(int), hipMemcpyDeviceToHost, stream_));
    }
  };
  // The ~1050 launches of a decode are identical from call to call for a given (slot, batch,
  // options). The first call with a signature runs them eagerly (which also performs the
  // kernels' one-time attribute set-up) and then captures one hipGraph per slot; later calls
  // replay the slot's graph: one host call instead of ~1100. The logits tap stays eager.
  auto key_of = [&](int si) {
    return std::vector<long long>{si, batch, max_pos, n_prompt, chunks, long(stop_at_eot), fc2_ksplit, bf16, absorbed ? 1 : 0, n_abs, group, stream_override, forced ? 1 : 0, pipelined ? 1 : 0};
  };
  // the cached form's graphs are captured for EVERY slot at once (below) and hold each slot's cache pointer: all of
  // them must exist before the capture (need_cross_kv allocates; nothing may be allocated inside a capture)
  if (!absorbed && use_graphs && !logits_host) {
    for (Slot& s : slots_) need_cross_kv(s);
  }
  hipGraphExec_t exec = nullptr;
  if (use_graphs && !logits_host) {
    auto it = graphs_.find(key_of(slot_idx));
    if (it != graphs_.end()) {
      exec = it->second.exec;
      steps = it->second.steps;
    }
  }
  if (exec) {
    HIPCHK(hipGraphLaunch(exec, stream_));
  } else {
    dt_on = dec_timers && !logits_host;
    dt_cls.clear();
    enqueue_all(slot_idx);
    dt_on = false;
    const int eager_steps = steps;
    // the eager work of THIS batch is queued: its completion event and step count are set before anything that can
    // fail, so that submit() / collect() stay consistent whatever happens to the captures below
    HIPCHK(hipEventRecord(slot.dec_done, stream_));
    slot.steps = eager_steps;
    for (int j = 1; j < group; ++j) {
      Slot& sb = slots_[(slot_idx + j) % kSlots];
      HIPCHK(hipEventRecord(sb.dec_done, stream_));
      sb.pair_leader = slot_idx, sb.steps = eager_steps, sb.dec = slot.dec;
    }
    if (use_graphs && !logits_host) {
      // A capture or instantiation failure is not fatal: the decoder keeps launching eagerly (same kernels, same
      // results, more host time per batch) and the engine stops trying.
      std::map<std::vector<long long>, GraphEntry> fresh;
      try {
        for (int si = 0; si < kSlots; ++si) {
          hipStream_t cs = dec_stream_at(dec_of(si));
          hipGraph_t graph = nullptr;
          hipGraphExec_t ge = nullptr;
          HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
          try {
            enqueue_all(si);
          } catch (...) {
            (void)hipStreamEndCapture(cs, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
          }
          HIPCHK(hipStreamEndCapture(cs, &graph));
          const hipError_t ie = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
          (void)hipGraphDestroy(graph);
          if (ie != hipSuccess) throw Error(kErrDevice, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
          fresh[key_of(si)] = GraphEntry{ge, steps};
        }
        for (auto& g : fresh) graphs_[g.first] = g.second;  // all slots or none
      } catch (const std::exception& e) {
        for (auto& g : fresh) (void)hipGraphExecDestroy(g.second.exec);
        (void)hipGetLastError();
        use_graphs = 0;
        std::fprintf(stderr, "[wt] decoder hipGraph capture failed (%s): continuing with eager launches\n", e.what());
      }
    }
    steps = eager_steps;
    return;
  }
  HIPCHK(hipEventRecord(slot.dec_done, stream_));
  slot.steps = steps;
  for (int j = 1; j < group; ++j) {
    Slot& sb = slots_[(slot_idx + j) % kSlots];
    HIPCHK(hipEventRecord(sb.dec_done, stream_));
    sb.pair_leader = slot_idx, sb.steps = steps, sb.dec = slot.dec;
  }
}

#undef DT

void Engine::debug_concurrency(const float* d_mel, int batch, int n_dec, int n_enc, float* dec_ms,
                               float* enc_ms) {
  if (!inflight_.empty()) throw Error(1, "collect the submitted batches first");
  if (n_dec < 0 || n_dec > 4 || n_enc < 0 || n_enc > 4) throw Error(1, "debug_concurrency: 0..4 decodes, 0..4 encoder passes");
  ensure_batch(batch);
  sync();
  for (int i = 0; i < n_dec; ++i)
    if (!slots_[i].used) throw Error(1, "debug_concurrency: run a batch per slot first so every slot holds a cross-KV cache");
  select_stream(true);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < n_dec; ++i) decode_enqueue(batch, i, nullptr, 0);
  HIPCHK(hipEventRecord(e0, stream_));
  for (int i = 0; i < n_enc; ++i) {
    if (enc_slot_ < 4) enc_slot_ = 4;  // encoder passes alternate over slots 4 and 5, away from the decoders' caches
    encode_enqueue(d_mel, batch);
  }
  HIPCHK(hipEventRecord(e1, stream_));
  sync();
  for (int i = 0; i < n_dec; ++i) HIPCHK(hipEventElapsedTime(&dec_ms[i], slots_[i].dec_begin, slots_[i].dec_done));
  HIPCHK(hipEventElapsedTime(enc_ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

void Engine::decode_collect(int slot_idx, int64_t* ids, int32_t* n_ids) {
  Slot& slot = slots_[slot_idx];
  HIPCHK(hipEventSynchronize(slot.dec_done));
  HIPCHK(hipGetLastError());
  if (*slot.h_flag != 0) {
    *slot.h_flag = 0;
    throw Error(5, "non-finite encoder output: an operand left the fp16 range of the default contraction kernels "
                   "(|activation| > 65504 or |weight| > 1023) or the input holds NaN/Inf; gemm_variant 16 and "
                   "attn_variant 1 (bf16 three-plane split) have the full fp32 range");
  }
  const int batch = slot.batch, stride = 32;
  // (a batch decoded by its pair leader's chain has its ids in its OWN buffers; the leader only times the chain)
  const Slot& src = slot.pair_leader >= 0 ? slots_[slot.pair_leader] : slot;
  for (int b = 0; b < batch; ++b) {
    n_ids[b] = slot.h_n[b];
    for (int i = 0; i < stride; ++i)
      ids[size_t(b) * stride + i] = i < slot.h_n[b] ? slot.h_ids[size_t(b) * stride + i] : 0;
  }
  resolve_kernel_stats(slot_idx);
  float ms = 0;
  timings_.batch = batch;
  timings_.decoder_steps = slot.steps;
  if (hipEventElapsedTime(&ms, slot.enc_begin, slot.enc_mid) == hipSuccess) timings_.encoder_ms = ms;
  if (hipEventElapsedTime(&ms, slot.enc_mid, slot.enc_done) == hipSuccess) timings_.cross_kv_ms = ms;
  if (hipEventElapsedTime(&ms, src.dec_begin, slot.dec_done) == hipSuccess) timings_.decoder_ms = ms;
  if (hipEventElapsedTime(&ms, slot.enc_begin, slot.dec_done) == hipSuccess) timings_.total_ms = ms;
  if (!slot.dt_cls.empty()) {
    static const char* kNames[11] = {"qkv(LN)", "self_attn", "o_proj", "q_abs(LN)", "cross_attn", "co",
                                     "fc1(LN)", "fc2", "abs_combine", "logits", "select"};
    double tot[11] = {0};
    int cnt[11] = {0};
    for (size_t i = 0; i < slot.dt_cls.size(); ++i) {
      float ms_k = 0;
      if (hipEventElapsedTime(&ms_k, slot.dt_events[2 * i], slot.dt_events[2 * i + 1]) == hipSuccess) {
        tot[slot.dt_cls[i]] += ms_k;
        cnt[slot.dt_cls[i]] += 1;
      }
    }
    fprintf(stderr, "[wt-dec-timers] slot %d:", slot_idx);
    for (int c = 0; c < 11; ++c)
      if (cnt[c]) fprintf(stderr, " %s %.2f ms (%d x %.1f us)", kNames[c], tot[c], cnt[c], 1e3 * tot[c] / cnt[c]);
    fprintf(stderr, "\n");
    slot.dt_cls.clear();
  }
  static const bool trace = getenv("WT_TRACE_PIPELINE") != nullptr;
  if (trace) {  // device timeline of the batch relative to the first traced batch, for pipeline analysis
    hipEvent_t base = trace_base_;
    float t[4] = {0, 0, 0, 0};
    hipEvent_t evs[4] = {slot.enc_begin, slot.enc_done, src.dec_begin, slot.dec_done};  // a paired batch: its leader's chain
    for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&t[i], base, evs[i]);
    (void)hipGetLastError();
    fprintf(stderr, "[wt-trace] slot %d enc %.3f..%.3f dec %.3f..%.3f\n", slot_idx, t[0], t[1], t[2], t[3]);
  }
  if (timings_.logmel_ms < 0) {
    timings_.logmel_ms = 0;
    if (hipEventElapsedTime(&ms, ev_[0], ev_[1]) == hipSuccess) timings_.logmel_ms = ms;
  }
}

}  // namespace wt

// ORACLE — TEST INFRASTRUCTURE ONLY (see wt_oracle.h).
//
// CPU fp32 restatement of the two graphs the reference runs through
// tflite::Interpreter::Invoke() (whisper.tflite/whisper.cpp:295 encoder, :375
// decoder) and of the greedy loop around them (:327-402).  The graphs themselves
// are not in the reference tree; they are OpenAI Whisper's AudioEncoder and
// TextDecoder as traced by export/generate_onnx.py:85-120:
//
//   encoder: Conv1d(n_mels->d,k3,p1)+GELU -> Conv1d(d->d,k3,s2,p1)+GELU -> +pos
//            -> L x { x += Attn(LN(x)); x += W2 GELU(W1 LN(x)) } -> LN
//   attn   : q = Wq x + bq, k = Wk x, v = Wv x + bv; q,k scaled by d_head^-1/4 each;
//            softmax(q k^T) v per head; out = Wo . + bo
//   decoder: tok_emb[ids] + pos_emb -> L x { causal self-attn; cross-attn; MLP }
//            -> LN -> logits = x . tok_emb^T
//
// Model-arithmetic parity vs the reference itself is UNPINNED (no TFLite runtime,
// no .tflite file); this file is pinned against HuggingFace transformers' Whisper
// (tests/golden/model_*.npz, tools/gen_golden.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "wt_oracle.h"

namespace {

// ------------------------------------------------------------ .wtw reader ---
// Independent reader of the product's weight-file layout (documented in
// whisper.tflite_amd/csrc/wtw_format.h); deliberately shares no code with it.
struct Dims {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
};

struct TensorView {
  const float* p = nullptr;
  uint32_t shape[4] = {0, 0, 0, 0};
  uint32_t ndim = 0;
};

}  // namespace

struct wto_model {
  Dims dims{};
  void* map = nullptr;
  size_t map_bytes = 0;
  std::map<std::string, TensorView> tensors;
  const float* get(const std::string& name) const {
    auto it = tensors.find(name);
    if (it == tensors.end()) {
      std::fprintf(stderr, "[oracle] missing tensor %s\n", name.c_str());
      std::abort();
    }
    return it->second.p;
  }
};

namespace {

// ------------------------------------------------------------- primitives ---

void parallel_for(int n_threads, int n_items, const std::function<void(int, int)>& body) {
  if (n_threads <= 1 || n_items <= 1) {
    body(0, n_items);
    return;
  }
  const int nt = std::min(n_threads, n_items);
  std::vector<std::thread> pool;
  const int chunk = (n_items + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) {
    const int b = t * chunk, e = std::min(n_items, b + chunk);
    if (b < e) pool.emplace_back(body, b, e);
  }
  body(0, std::min(n_items, chunk));
  for (auto& th : pool) th.join();
}

typedef float v8f __attribute__((vector_size(32), aligned(4)));

inline float hsum(v8f v) {
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  return s;
}

// out[m][n] = sum_k a[m*lda + k] * w[n*ldw + k] (+ bias[n]); rows m in [m0, m1)
void linear_rows(float* out, int ldo, const float* a, int lda, const float* w, int ldw,
                 const float* bias, int m0, int m1, int N, int K) {
  const int K8 = K & ~7;
  int m = m0;
  for (; m + 4 <= m1; m += 4) {
    int n = 0;
    for (; n + 3 <= N; n += 3) {
      v8f acc[4][3];
      for (auto& r : acc)
        for (auto& c : r) c = v8f{0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < K8; k += 8) {
        const v8f w0 = *reinterpret_cast<const v8f*>(w + size_t(n + 0) * ldw + k);
        const v8f w1 = *reinterpret_cast<const v8f*>(w + size_t(n + 1) * ldw + k);
        const v8f w2 = *reinterpret_cast<const v8f*>(w + size_t(n + 2) * ldw + k);
        for (int i = 0; i < 4; ++i) {
          const v8f av = *reinterpret_cast<const v8f*>(a + size_t(m + i) * lda + k);
          acc[i][0] += av * w0;
          acc[i][1] += av * w1;
          acc[i][2] += av * w2;
        }
      }
      for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) {
          float s = hsum(acc[i][j]);
          for (int k = K8; k < K; ++k) s += a[size_t(m + i) * lda + k] * w[size_t(n + j) * ldw + k];
          out[size_t(m + i) * ldo + n + j] = s + (bias ? bias[n + j] : 0.0f);
        }
      }
    }
    for (; n < N; ++n) {
      for (int i = 0; i < 4; ++i) {
        v8f acc = v8f{0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < K8; k += 8) {
          acc += *reinterpret_cast<const v8f*>(a + size_t(m + i) * lda + k) *
                 *reinterpret_cast<const v8f*>(w + size_t(n) * ldw + k);
        }
        float s = hsum(acc);
        for (int k = K8; k < K; ++k) s += a[size_t(m + i) * lda + k] * w[size_t(n) * ldw + k];
        out[size_t(m + i) * ldo + n] = s + (bias ? bias[n] : 0.0f);
      }
    }
  }
  for (; m < m1; ++m) {
    int n = 0;
    for (; n + 4 <= N; n += 4) {
      v8f acc[4];
      for (auto& c : acc) c = v8f{0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < K8; k += 8) {
        const v8f av = *reinterpret_cast<const v8f*>(a + size_t(m) * lda + k);
        for (int j = 0; j < 4; ++j)
          acc[j] += av * *reinterpret_cast<const v8f*>(w + size_t(n + j) * ldw + k);
      }
      for (int j = 0; j < 4; ++j) {
        float s = hsum(acc[j]);
        for (int k = K8; k < K; ++k) s += a[size_t(m) * lda + k] * w[size_t(n + j) * ldw + k];
        out[size_t(m) * ldo + n + j] = s + (bias ? bias[n + j] : 0.0f);
      }
    }
    for (; n < N; ++n) {
      v8f acc = v8f{0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < K8; k += 8) {
        acc += *reinterpret_cast<const v8f*>(a + size_t(m) * lda + k) *
               *reinterpret_cast<const v8f*>(w + size_t(n) * ldw + k);
      }
      float s = hsum(acc);
      for (int k = K8; k < K; ++k) s += a[size_t(m) * lda + k] * w[size_t(n) * ldw + k];
      out[size_t(m) * ldo + n] = s + (bias ? bias[n] : 0.0f);
    }
  }
}

// y = x W^T + b with W [N][K] (torch.nn.Linear layout).  Parallel over rows, or
// over output columns when there are few rows (decoder steps).
void linear(float* out, const float* in, const float* w, const float* bias, int M, int N, int K,
            int nt) {
  if (M >= 4 * nt || nt <= 1) {
    const int blocks = (M + 3) / 4;
    parallel_for(nt, blocks, [&](int b, int e) {
      linear_rows(out, N, in, K, w, K, bias, b * 4, std::min(M, e * 4), N, K);
    });
  } else {
    const int blocks = (N + 47) / 48;
    parallel_for(nt, blocks, [&](int b, int e) {
      const int n0 = b * 48, n1 = std::min(N, e * 48);
      linear_rows(out + n0, N, in, K, w + size_t(n0) * K, K, bias ? bias + n0 : nullptr, 0, M,
                  n1 - n0, K);
    });
  }
}

void layer_norm(float* out, const float* in, const float* g, const float* b, int M, int d) {
  for (int m = 0; m < M; ++m) {  // torch.nn.LayerNorm, eps 1e-5, biased variance
    const float* x = in + size_t(m) * d;
    double mean = 0;
    for (int i = 0; i < d; ++i) mean += x[i];
    mean /= d;
    double var = 0;
    for (int i = 0; i < d; ++i) var += (x[i] - mean) * (x[i] - mean);
    var /= d;
    const float rstd = static_cast<float>(1.0 / std::sqrt(var + 1e-5));
    const float mu = static_cast<float>(mean);
    float* y = out + size_t(m) * d;
    for (int i = 0; i < d; ++i) y[i] = (x[i] - mu) * rstd * g[i] + b[i];
  }
}

inline float gelu(float x) {  // torch.nn.GELU() default: exact erf form
  return 0.5f * x * (1.0f + std::erf(x * 0.70710678118654752440f));
}

void add_inplace(float* x, const float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] += y[i];
}

// Multi-head attention core on projected q [Tq][d], k,v [Tk][d] (rows are time).
// causal_offset >= 0: query row i may see keys j <= i + causal_offset; -1: no mask.
void attention(float* out, const float* q, const float* k, const float* v, int Tq, int Tk, int d,
               int heads, int causal_offset, int nt) {
  const int dh = d / heads;
  const float scale = std::pow(static_cast<float>(dh), -0.25f);
  parallel_for(nt, heads, [&](int h0, int h1) {
    const int Tkp = (Tk + 7) & ~7;
    std::vector<float> qs(size_t(Tq) * dh), ks(size_t(Tk) * dh), vt(size_t(dh) * Tkp, 0.0f);
    std::vector<float> s(size_t(Tq) * Tkp, 0.0f), o(size_t(Tq) * dh);
    for (int h = h0; h < h1; ++h) {
      for (int t = 0; t < Tq; ++t)
        for (int c = 0; c < dh; ++c) qs[size_t(t) * dh + c] = q[size_t(t) * d + h * dh + c] * scale;
      for (int t = 0; t < Tk; ++t) {
        for (int c = 0; c < dh; ++c) {
          ks[size_t(t) * dh + c] = k[size_t(t) * d + h * dh + c] * scale;
          vt[size_t(c) * Tkp + t] = v[size_t(t) * d + h * dh + c];
        }
      }
      linear_rows(s.data(), Tkp, qs.data(), dh, ks.data(), dh, nullptr, 0, Tq, Tk, dh);
      for (int t = 0; t < Tq; ++t) {
        float* row = s.data() + size_t(t) * Tkp;
        const int lim = causal_offset >= 0 ? std::min(Tk, t + causal_offset + 1) : Tk;
        float mx = -INFINITY;
        for (int j = 0; j < lim; ++j) mx = std::max(mx, row[j]);
        float sum = 0;
        for (int j = 0; j < lim; ++j) {
          row[j] = std::exp(row[j] - mx);
          sum += row[j];
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < lim; ++j) row[j] *= inv;
        for (int j = lim; j < Tkp; ++j) row[j] = 0.0f;
      }
      linear_rows(o.data(), dh, s.data(), Tkp, vt.data(), Tkp, nullptr, 0, Tq, dh, Tkp);
      for (int t = 0; t < Tq; ++t)
        for (int c = 0; c < dh; ++c) out[size_t(t) * d + h * dh + c] = o[size_t(t) * dh + c];
    }
  });
}

// Conv1d, kernel 3, padding 1, given stride; in [Cin][Tin] (channel-major, as the mel
// is), weight [Cout][Cin][3]; out [Tout][Cout] (time-major), GELU applied.
void conv1d_k3_gelu(float* out, const float* in, const float* w, const float* bias, int Cin,
                    int Tin, int Cout, int stride, bool in_time_major, int nt) {
  const int Tout = (Tin + 2 - 3) / stride + 1;
  const int K = Cin * 3;
  std::vector<float> col(size_t(Tout) * K);
  for (int t = 0; t < Tout; ++t) {
    for (int ci = 0; ci < Cin; ++ci) {
      for (int kk = 0; kk < 3; ++kk) {
        const int ti = t * stride + kk - 1;
        float x = 0.0f;
        if (ti >= 0 && ti < Tin) x = in_time_major ? in[size_t(ti) * Cin + ci] : in[size_t(ci) * Tin + ti];
        col[size_t(t) * K + ci * 3 + kk] = x;
      }
    }
  }
  linear(out, col.data(), w, bias, Tout, Cout, K, nt);
  const size_t n = size_t(Tout) * Cout;
  for (size_t i = 0; i < n; ++i) out[i] = gelu(out[i]);
}

struct AttnW {
  const float *wq, *bq, *wk, *wv, *bv, *wo, *bo;
};
AttnW attn_weights(const wto_model* m, const std::string& p) {
  return AttnW{m->get(p + ".query.weight"), m->get(p + ".query.bias"), m->get(p + ".key.weight"),
               m->get(p + ".value.weight"), m->get(p + ".value.bias"), m->get(p + ".out.weight"),
               m->get(p + ".out.bias")};
}

void mlp_block(const wto_model* m, const std::string& blk, float* x, int T, int d, int nt) {
  std::vector<float> ln(size_t(T) * d), hid(size_t(T) * 4 * d), y(size_t(T) * d);
  layer_norm(ln.data(), x, m->get(blk + ".mlp_ln.weight"), m->get(blk + ".mlp_ln.bias"), T, d);
  linear(hid.data(), ln.data(), m->get(blk + ".mlp.0.weight"), m->get(blk + ".mlp.0.bias"), T, 4 * d, d, nt);
  for (auto& h : hid) h = gelu(h);
  linear(y.data(), hid.data(), m->get(blk + ".mlp.2.weight"), m->get(blk + ".mlp.2.bias"), T, d, 4 * d, nt);
  add_inplace(x, y.data(), y.size());
}

void encode(const wto_model* m, const float* mel, float* out, int nt) {
  const Dims& c = m->dims;
  const int d = c.n_audio_state, T0 = 2 * c.n_audio_ctx, T = c.n_audio_ctx;
  std::vector<float> h1(size_t(T0) * d);
  conv1d_k3_gelu(h1.data(), mel, m->get("encoder.conv1.weight"), m->get("encoder.conv1.bias"),
                 c.n_mels, T0, d, 1, false, nt);
  std::vector<float> x(size_t(T) * d);
  conv1d_k3_gelu(x.data(), h1.data(), m->get("encoder.conv2.weight"), m->get("encoder.conv2.bias"),
                 d, T0, d, 2, true, nt);
  add_inplace(x.data(), m->get("encoder.positional_embedding"), x.size());
  std::vector<float> ln(size_t(T) * d), q(size_t(T) * d), k(size_t(T) * d), v(size_t(T) * d),
      a(size_t(T) * d), y(size_t(T) * d);
  for (int l = 0; l < c.n_audio_layer; ++l) {
    const std::string blk = "encoder.blocks." + std::to_string(l);
    const AttnW w = attn_weights(m, blk + ".attn");
    layer_norm(ln.data(), x.data(), m->get(blk + ".attn_ln.weight"), m->get(blk + ".attn_ln.bias"), T, d);
    linear(q.data(), ln.data(), w.wq, w.bq, T, d, d, nt);
    linear(k.data(), ln.data(), w.wk, nullptr, T, d, d, nt);
    linear(v.data(), ln.data(), w.wv, w.bv, T, d, d, nt);
    attention(a.data(), q.data(), k.data(), v.data(), T, T, d, c.n_audio_head, -1, nt);
    linear(y.data(), a.data(), w.wo, w.bo, T, d, d, nt);
    add_inplace(x.data(), y.data(), y.size());
    mlp_block(m, blk, x.data(), T, d, nt);
  }
  layer_norm(out, x.data(), m->get("encoder.ln_post.weight"), m->get("encoder.ln_post.bias"), T, d);
}

// Decoder state: self-attention K/V per layer for the positions processed so far,
// cross-attention K/V per layer projected once from the encoder output.
struct DecState {
  int n_past = 0;
  std::vector<std::vector<float>> self_k, self_v, cross_k, cross_v;
};

void cross_kv(const wto_model* m, const float* enc_out, DecState& st, int nt) {
  const Dims& c = m->dims;
  const int d = c.n_text_state, Ta = c.n_audio_ctx;
  st.cross_k.assign(c.n_text_layer, std::vector<float>(size_t(Ta) * d));
  st.cross_v.assign(c.n_text_layer, std::vector<float>(size_t(Ta) * d));
  st.self_k.assign(c.n_text_layer, std::vector<float>(size_t(c.n_text_ctx) * d));
  st.self_v.assign(c.n_text_layer, std::vector<float>(size_t(c.n_text_ctx) * d));
  st.n_past = 0;
  for (int l = 0; l < c.n_text_layer; ++l) {
    const AttnW w = attn_weights(m, "decoder.blocks." + std::to_string(l) + ".cross_attn");
    linear(st.cross_k[l].data(), enc_out, w.wk, nullptr, Ta, d, d, nt);
    linear(st.cross_v[l].data(), enc_out, w.wv, w.bv, Ta, d, d, nt);
  }
}

// Runs positions [st.n_past, st.n_past + n_new) and returns the logits of the last one.
void decode_positions(const wto_model* m, DecState& st, const int64_t* ids, int n_new,
                      float* logits_last, int nt) {
  const Dims& c = m->dims;
  const int d = c.n_text_state, Ta = c.n_audio_ctx, T = n_new, p0 = st.n_past;
  const float* emb = m->get("decoder.token_embedding.weight");
  const float* pos = m->get("decoder.positional_embedding");
  std::vector<float> x(size_t(T) * d), ln(size_t(T) * d), q(size_t(T) * d), a(size_t(T) * d),
      y(size_t(T) * d);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < d; ++i)
      x[size_t(t) * d + i] = emb[size_t(ids[t]) * d + i] + pos[size_t(p0 + t) * d + i];
  for (int l = 0; l < c.n_text_layer; ++l) {
    const std::string blk = "decoder.blocks." + std::to_string(l);
    {
      const AttnW w = attn_weights(m, blk + ".attn");
      layer_norm(ln.data(), x.data(), m->get(blk + ".attn_ln.weight"), m->get(blk + ".attn_ln.bias"), T, d);
      linear(q.data(), ln.data(), w.wq, w.bq, T, d, d, nt);
      linear(st.self_k[l].data() + size_t(p0) * d, ln.data(), w.wk, nullptr, T, d, d, nt);
      linear(st.self_v[l].data() + size_t(p0) * d, ln.data(), w.wv, w.bv, T, d, d, nt);
      attention(a.data(), q.data(), st.self_k[l].data(), st.self_v[l].data(), T, p0 + T, d,
                c.n_text_head, p0, nt);
      linear(y.data(), a.data(), w.wo, w.bo, T, d, d, nt);
      add_inplace(x.data(), y.data(), y.size());
    }
    {
      const AttnW w = attn_weights(m, blk + ".cross_attn");
      layer_norm(ln.data(), x.data(), m->get(blk + ".cross_attn_ln.weight"),
                 m->get(blk + ".cross_attn_ln.bias"), T, d);
      linear(q.data(), ln.data(), w.wq, w.bq, T, d, d, nt);
      attention(a.data(), q.data(), st.cross_k[l].data(), st.cross_v[l].data(), T, Ta, d,
                c.n_text_head, -1, nt);
      linear(y.data(), a.data(), w.wo, w.bo, T, d, d, nt);
      add_inplace(x.data(), y.data(), y.size());
    }
    mlp_block(m, blk, x.data(), T, d, nt);
  }
  st.n_past += T;
  std::vector<float> last(d);
  layer_norm(last.data(), x.data() + size_t(T - 1) * d, m->get("decoder.ln.weight"),
             m->get("decoder.ln.bias"), 1, d);
  linear(logits_last, last.data(), emb, nullptr, 1, c.n_vocab, d, nt);
}

int decode_greedy(const wto_model* m, const float* enc_out, const int64_t* prompt, int n_prompt,
                  int max_positions, int64_t eot, int stop_at_eot, int use_cache, int nt,
                  int64_t* ids_out, int* n_ids_out, float* logits_out) {
  const Dims& c = m->dims;
  std::vector<int64_t> ids(prompt, prompt + n_prompt);
  std::vector<float> logits(c.n_vocab);
  DecState st;
  if (use_cache) cross_kv(m, enc_out, st, nt);
  int steps = 0;
  // whisper.cpp:367: i is the index of the last position fed to the decoder
  for (int i = n_prompt - 1; i < max_positions; ++i) {
    if (use_cache) {
      decode_positions(m, st, ids.data() + st.n_past, int(ids.size()) - st.n_past, logits.data(), nt);
    } else {
      // whisper.cpp:368-375: every Invoke() re-runs the whole prefix, cross K/V included
      cross_kv(m, enc_out, st, nt);
      decode_positions(m, st, ids.data(), int(ids.size()), logits.data(), nt);
    }
    if (logits_out) std::memcpy(logits_out + size_t(steps) * c.n_vocab, logits.data(), sizeof(float) * c.n_vocab);
    const int64_t next = wto_argmax_last(logits.data(), c.n_vocab);
    ids.push_back(next);
    ++steps;
    if (stop_at_eot && next == eot) break;  // :397-399
  }
  std::copy(ids.begin(), ids.end(), ids_out);
  *n_ids_out = static_cast<int>(ids.size());
  return steps;
}

}  // namespace

extern "C" wto_model* wto_model_open(const char* path) {
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) return nullptr;
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 128) {
    ::close(fd);
    return nullptr;
  }
  void* map = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (map == MAP_FAILED) return nullptr;
  const uint8_t* base = static_cast<const uint8_t*>(map);
  uint32_t magic, version, n_tensors, table_off;
  std::memcpy(&magic, base + 0, 4);
  std::memcpy(&version, base + 4, 4);
  std::memcpy(&n_tensors, base + 8, 4);
  std::memcpy(&table_off, base + 12, 4);
  if (magic != 0x31575457u || version != 1 ||
      size_t(table_off) + size_t(n_tensors) * 128 > size_t(st.st_size)) {
    munmap(map, st.st_size);
    return nullptr;
  }
  auto* m = new wto_model;
  m->map = map;
  m->map_bytes = st.st_size;
  std::memcpy(&m->dims, base + 16, sizeof(Dims));
  for (uint32_t i = 0; i < n_tensors; ++i) {
    const uint8_t* e = base + table_off + size_t(i) * 128;
    char name[81];
    std::memcpy(name, e, 80);
    name[80] = 0;
    TensorView tv;
    std::memcpy(&tv.ndim, e + 84, 4);
    std::memcpy(tv.shape, e + 88, 16);
    uint64_t off, nbytes;
    std::memcpy(&off, e + 104, 8);
    std::memcpy(&nbytes, e + 112, 8);
    if (off + nbytes > uint64_t(st.st_size)) {
      wto_model_close(m);
      return nullptr;
    }
    tv.p = reinterpret_cast<const float*>(base + off);
    m->tensors[name] = tv;
  }
  return m;
}

extern "C" void wto_model_close(wto_model* m) {
  if (!m) return;
  if (m->map) munmap(m->map, m->map_bytes);
  delete m;
}

extern "C" void wto_model_dims(const wto_model* m, int32_t out[10]) {
  std::memcpy(out, &m->dims, sizeof(Dims));
}

extern "C" int wto_encode(const wto_model* m, const float* mel, float* enc_out, int n_threads) {
  encode(m, mel, enc_out, std::max(1, n_threads));
  return 0;
}

extern "C" int wto_decode_greedy(const wto_model* m, const float* enc_out, const int64_t* prompt,
                                 int n_prompt, int max_positions, int64_t eot, int stop_at_eot,
                                 int use_cache, int n_threads, int64_t* ids_out, int* n_ids_out,
                                 float* logits_out) {
  return decode_greedy(m, enc_out, prompt, n_prompt, max_positions, eot, stop_at_eot, use_cache,
                       std::max(1, n_threads), ids_out, n_ids_out, logits_out);
}

extern "C" int wto_encdec_batch(const wto_model* m, const float* mel, int batch,
                                const int64_t* prompt, int n_prompt, int max_positions, int64_t eot,
                                int stop_at_eot, int use_cache, int n_threads, int64_t* ids_out,
                                int* n_ids_out) {
  const Dims& c = m->dims;
  const size_t mel_sz = size_t(c.n_mels) * 2 * c.n_audio_ctx;
  parallel_for(std::max(1, n_threads), batch, [&](int b0, int b1) {
    std::vector<float> enc(size_t(c.n_audio_ctx) * c.n_audio_state);
    for (int b = b0; b < b1; ++b) {
      encode(m, mel + size_t(b) * mel_sz, enc.data(), 1);
      decode_greedy(m, enc.data(), prompt, n_prompt, max_positions, eot, stop_at_eot, use_cache, 1,
                    ids_out + size_t(b) * (max_positions + 1), n_ids_out + b, nullptr);
    }
  });
  return 0;
}

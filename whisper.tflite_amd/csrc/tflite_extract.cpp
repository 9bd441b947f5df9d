// .tflite -> .wtw weight extractor.
//
// The reference loads "<prefix>.encoder.tflite" and "<prefix>.decoder.tflite" through
// tflite::FlatBufferModel::BuildFromFile (whisper.tflite/whisper.cpp:261-271, :743-744); the files are written by
// export/generate_onnx.py:135-163 (PyTorch -> ONNX -> TF -> TFLite, converter.optimizations = [DEFAULT]: weights
// stored as int8 with float scales, "dynamic range" quantisation, float I/O).  This build has no TFLite and no
// FlatBuffers library, so the container format is walked by hand:
//
//   file   = [u32 offset of the root table][4-byte identifier "TFL3"] ...
//   table  = [i32 distance back to its vtable] fields...;  vtable = [u16 vtable bytes][u16 table bytes][u16 field
//            offsets...] (0 = field absent); a field holding a table / vector / string is a u32 offset relative to
//            the field's own position;  vector = [u32 count] elements...
//
// and the schema's field numbers (tensorflow/lite/schema/schema.fbs, version 3) are:
//   Model        { 0 version, 1 operator_codes, 2 subgraphs, 3 description, 4 buffers }
//   SubGraph     { 0 tensors, 1 inputs, 2 outputs, 3 operators, 4 name }
//   Tensor       { 0 shape [i32], 1 type (u8), 2 buffer (u32), 3 name, 4 quantization }
//   Quantization { 0 min, 1 max, 2 scale [f32], 3 zero_point [i64], 4 details_type, 5 details, 6 quantized_dimension }
//   Buffer       { 0 data [u8], 1 offset (u64), 2 size (u64) }       (offset/size: payload outside the FlatBuffer)
//   Operator     { 0 opcode_index, 1 inputs [i32], 2 outputs [i32] }
//   OperatorCode { 0 deprecated_builtin_code (i8), 1 custom_code, 2 version, 3 builtin_code (i32) }
//   TensorType   FLOAT32 0, FLOAT16 1, INT32 2, UINT8 3, INT64 4, INT8 9
//
// Constants are de-quantised as TFLite's hybrid kernels define them: w = scale[c] * (q - zero_point[c]) with c the
// index along quantized_dimension (one scale: per tensor).
//
// Which constant is which parameter: tensor names survive the ONNX -> TF -> TFLite chain only partly, so two rules
// are applied in order.  (1) A constant whose name contains an OpenAI parameter path ("blocks.0.attn.query.weight",
// "conv1.bias", ... with the graph's "encoder." / "decoder." prefix optional) is that parameter.  (2) The remaining
// parameters are matched in forward order against the remaining constants in the order the graph's operators first
// use them, by element count: the trace of whisper's forward() fixes that order (conv1, conv2, positional
// embedding, then per block attn_ln, query, key, value, out, [cross_attn_ln, cross query, key, value, out,] mlp_ln,
// mlp.0, mlp.2, and the final LayerNorm).  A 2-D weight stored [in][out] (a MatMul right-hand side) is transposed to
// torch's [out][in]; when the two extents are equal the consuming operator decides (FULLY_CONNECTED keeps [out][in]).
// Convolution kernels are brought from TFLite's [out][1][k][in] (or TF's [1][k][in][out]) to torch's [out][in][k].
//
// PARITY UNPINNED: no .tflite file exists in this environment; the reader is exercised on files produced by
// tests/tflite_writer.py (same schema, same quantisation formulas) and the mapping rules on its two naming modes.
#include "tflite_extract.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "error.h"
#include "weights_gen.h"

namespace wt {

bool file_exists(const std::string& path) {
  struct stat st;
  return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

namespace {

// ---------------------------------------------------------------- FlatBuffer view ---
struct Fb {
  const uint8_t* base = nullptr;
  size_t size = 0;
  std::string path;

  [[noreturn]] void bad(const char* what) const { throw Error(kErrFormat, path + ": malformed .tflite (" + what + ")"); }
  void need(size_t off, size_t n) const {
    if (off > size || n > size - off) bad("offset outside the file");
  }
  template <class T>
  T rd(size_t off) const {
    need(off, sizeof(T));
    T v;
    std::memcpy(&v, base + off, sizeof(T));
    return v;
  }
  // position of field `idx` of the table at `t`, or 0 when absent
  size_t field(size_t t, int idx) const {
    const int32_t back = rd<int32_t>(t);
    const int64_t vt = int64_t(t) - back;
    if (vt < 0 || size_t(vt) + 4 > size) bad("vtable outside the file");
    const uint16_t vt_bytes = rd<uint16_t>(size_t(vt));
    const size_t slot = 4 + 2 * size_t(idx);
    if (slot + 2 > vt_bytes) return 0;
    const uint16_t off = rd<uint16_t>(size_t(vt) + slot);
    return off ? t + off : 0;
  }
  size_t indirect(size_t pos) const {  // follow a u32 relative offset stored at pos
    const size_t target = pos + rd<uint32_t>(pos);
    need(target, 4);
    return target;
  }
  template <class T>
  T scalar(size_t t, int idx, T dflt) const {
    const size_t p = field(t, idx);
    return p ? rd<T>(p) : dflt;
  }
  size_t table(size_t t, int idx) const {
    const size_t p = field(t, idx);
    return p ? indirect(p) : 0;
  }
  // vector field: returns element count and the position of element 0
  size_t vec(size_t t, int idx, size_t elem_bytes, size_t* first) const {
    const size_t p = field(t, idx);
    if (!p) {
      *first = 0;
      return 0;
    }
    const size_t v = indirect(p);
    const uint32_t n = rd<uint32_t>(v);
    need(v + 4, size_t(n) * elem_bytes);
    *first = v + 4;
    return n;
  }
  size_t vec_table(size_t first, size_t i) const { return indirect(first + 4 * i); }
  std::string str(size_t t, int idx) const {
    size_t first = 0;
    const size_t n = vec(t, idx, 1, &first);
    return n ? std::string(reinterpret_cast<const char*>(base + first), n) : std::string();
  }
};

struct MappedFile {
  void* p = MAP_FAILED;
  size_t n = 0;
  explicit MappedFile(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(kErrIo, "Failed to open file: " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 16) {
      ::close(fd);
      throw Error(kErrFormat, path + ": too small to be a .tflite file");
    }
    n = size_t(st.st_size);
    p = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) throw Error(kErrIo, "Failed to mmap file: " + path);
  }
  ~MappedFile() {
    if (p != MAP_FAILED) munmap(p, n);
  }
};

// ------------------------------------------------------------------- constants ---
enum TfType : int { kF32 = 0, kF16 = 1, kI32 = 2, kU8 = 3, kI64 = 4, kI8 = 9 };
constexpr int kOpFullyConnected = 9;  // BuiltinOperator_FULLY_CONNECTED

struct Constant {
  std::string name;
  std::vector<int> shape;
  std::vector<float> data;  // de-quantised
  int first_use = 1 << 30;  // index of the first operator that reads it
  int first_use_slot = 0;   // its position among that operator's inputs
  int first_use_opcode = -1;
  bool used = false;
  size_t numel() const { return data.size(); }
};

float half_to_float(uint16_t h) {
  const uint32_t sign = uint32_t(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: normalise
      int e = -1;
      uint32_t m = man;
      do {
        ++e;
        m <<= 1;
      } while ((m & 0x400u) == 0);
      bits = sign | uint32_t(127 - 15 - e) << 23 | (m & 0x3FFu) << 13;
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | man << 13;
  } else {
    bits = sign | (exp + 127 - 15) << 23 | man << 13;
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

std::vector<Constant> read_constants(const std::string& path) {
  const MappedFile file(path);
  Fb fb{static_cast<const uint8_t*>(file.p), file.n, path};
  if (std::memcmp(fb.base + 4, "TFL3", 4) != 0) fb.bad("identifier is not TFL3");
  const size_t model = fb.indirect(0);
  size_t codes0 = 0, subs0 = 0, bufs0 = 0;
  const size_t n_codes = fb.vec(model, 1, 4, &codes0);
  const size_t n_subs = fb.vec(model, 2, 4, &subs0);
  const size_t n_bufs = fb.vec(model, 4, 4, &bufs0);
  if (n_subs < 1) fb.bad("no subgraph");
  std::vector<int> opcode(n_codes, -1);
  for (size_t i = 0; i < n_codes; ++i) {
    const size_t oc = fb.vec_table(codes0, i);
    const int dep = fb.scalar<int8_t>(oc, 0, 0), full = fb.scalar<int32_t>(oc, 3, 0);
    opcode[i] = std::max(dep, full);  // builtin_code supersedes the deprecated byte once codes pass 127
  }
  const size_t sub = fb.vec_table(subs0, 0);
  size_t tens0 = 0, ops0 = 0;
  const size_t n_tens = fb.vec(sub, 0, 4, &tens0);
  const size_t n_ops = fb.vec(sub, 3, 4, &ops0);

  std::vector<Constant> out;
  std::vector<int> const_of_tensor(n_tens, -1);
  for (size_t ti = 0; ti < n_tens; ++ti) {
    const size_t t = fb.vec_table(tens0, ti);
    const uint32_t buf = fb.scalar<uint32_t>(t, 2, 0);
    if (buf == 0 || buf >= n_bufs) continue;  // buffer 0 is the empty sentinel: an activation
    const size_t b = fb.vec_table(bufs0, buf);
    size_t data0 = 0;
    size_t nbytes = fb.vec(b, 0, 1, &data0);
    if (nbytes == 0) {  // payload appended after the FlatBuffer (models above 2 GB)
      const uint64_t off = fb.scalar<uint64_t>(b, 1, 0), sz = fb.scalar<uint64_t>(b, 2, 0);
      if (off > 1 && sz > 0) {
        fb.need(size_t(off), size_t(sz));
        data0 = size_t(off);
        nbytes = size_t(sz);
      }
    }
    if (nbytes == 0) continue;
    Constant c;
    c.name = fb.str(t, 3);
    size_t shp0 = 0;
    const size_t rank = fb.vec(t, 0, 4, &shp0);
    size_t numel = 1;
    for (size_t k = 0; k < rank; ++k) {
      const int32_t e = fb.rd<int32_t>(shp0 + 4 * k);
      if (e < 0 || (e > 0 && numel > (size_t(1) << 40) / size_t(e))) fb.bad("tensor shape");
      c.shape.push_back(e);
      numel *= size_t(e);
    }
    const int type = fb.scalar<uint8_t>(t, 1, 0);
    const uint8_t* raw = fb.base + data0;
    if (type == kF32) {
      if (nbytes != numel * 4) fb.bad("float32 constant size");
      c.data.resize(numel);
      std::memcpy(c.data.data(), raw, nbytes);
    } else if (type == kF16) {
      if (nbytes != numel * 2) fb.bad("float16 constant size");
      c.data.resize(numel);
      for (size_t i = 0; i < numel; ++i) {
        uint16_t h;
        std::memcpy(&h, raw + 2 * i, 2);
        c.data[i] = half_to_float(h);
      }
    } else if (type == kI8 || type == kU8) {
      if (nbytes != numel) fb.bad("int8 constant size");
      const size_t q = fb.table(t, 4);
      size_t sc0 = 0, zp0 = 0;
      const size_t n_sc = q ? fb.vec(q, 2, 4, &sc0) : 0;
      const size_t n_zp = q ? fb.vec(q, 3, 8, &zp0) : 0;
      if (n_sc == 0) continue;  // an integer table without scales is not a weight
      const int qdim = q ? fb.scalar<int32_t>(q, 6, 0) : 0;
      size_t inner = 1, extent = 1;
      if (n_sc > 1) {
        if (qdim < 0 || size_t(qdim) >= rank || size_t(c.shape[qdim]) != n_sc) fb.bad("per-axis scale count");
        extent = n_sc;
        for (size_t k = size_t(qdim) + 1; k < rank; ++k) inner *= size_t(c.shape[k]);
      }
      c.data.resize(numel);
      for (size_t i = 0; i < numel; ++i) {
        const size_t ch = n_sc > 1 ? (i / inner) % extent : 0;
        const float scale = fb.rd<float>(sc0 + 4 * ch);
        const int64_t zp = n_zp ? fb.rd<int64_t>(zp0 + 8 * (n_zp > 1 ? ch : 0)) : 0;
        const int v = type == kI8 ? int(static_cast<int8_t>(raw[i])) : int(raw[i]);
        c.data[i] = scale * float(v - int(zp));
      }
    } else {
      continue;  // int32 / int64 constants are shapes and indices, not weights
    }
    const_of_tensor[ti] = int(out.size());
    out.push_back(std::move(c));
  }
  for (size_t oi = 0; oi < n_ops; ++oi) {
    const size_t op = fb.vec_table(ops0, oi);
    const uint32_t ci = fb.scalar<uint32_t>(op, 0, 0);
    size_t in0 = 0;
    const size_t n_in = fb.vec(op, 1, 4, &in0);
    for (size_t k = 0; k < n_in; ++k) {
      const int32_t ti = fb.rd<int32_t>(in0 + 4 * k);
      if (ti < 0 || size_t(ti) >= n_tens || const_of_tensor[ti] < 0) continue;
      Constant& c = out[size_t(const_of_tensor[ti])];
      if (int(oi) < c.first_use) {
        c.first_use = int(oi);
        c.first_use_slot = int(k);
        c.first_use_opcode = ci < opcode.size() ? opcode[ci] : -1;
      }
    }
  }
  return out;
}

// ---------------------------------------------------------------------- mapping ---
size_t numel_of(const std::vector<uint32_t>& s) {
  size_t n = 1;
  for (uint32_t v : s) n *= v;
  return n;
}

std::vector<int> squeezed(const std::vector<int>& s) {
  std::vector<int> o;
  for (int v : s)
    if (v != 1) o.push_back(v);
  return o;
}

// brings a constant's data into the torch layout `want` (shape of the .wtw tensor); false when the shapes cannot
// be reconciled
bool to_torch_layout(const Constant& c, const std::vector<uint32_t>& want, std::vector<float>* out) {
  const std::vector<int> have = squeezed(c.shape);
  std::vector<int> w;
  for (uint32_t v : want)
    if (v != 1) w.push_back(int(v));
  if (c.numel() != numel_of(want)) return false;
  if (want.size() == 1 || have == w) {
    if (want.size() == 2 && want[0] == want[1] && c.first_use_opcode != kOpFullyConnected && c.first_use_opcode >= 0) {
      // square MatMul right-hand side [in][out]
      const size_t n = want[0];
      out->resize(c.numel());
      for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) (*out)[i * n + j] = c.data[j * n + i];
      return true;
    }
    if (want.size() == 3 && want[0] == want[1]) {
      // conv2 [out][in][k] vs TFLite [out][k][in]: ambiguous by extents; the torch order only survives when k is last
      if (have.size() == 3 && have[2] != int(want[2])) return false;
    }
    *out = c.data;
    return true;
  }
  if (want.size() == 2 && have.size() == 2 && have[0] == w[1] && have[1] == w[0]) {  // stored [in][out]
    const size_t N = want[0], K = want[1];
    out->resize(c.numel());
    for (size_t n = 0; n < N; ++n)
      for (size_t k = 0; k < K; ++k) (*out)[n * K + k] = c.data[k * N + n];
    return true;
  }
  if (want.size() == 3 && have.size() == 3) {
    const size_t O = want[0], I = want[1], Kk = want[2];
    out->resize(c.numel());
    if (size_t(have[0]) == O && size_t(have[1]) == Kk && size_t(have[2]) == I) {  // TFLite CONV_2D filter [out][k][in]
      for (size_t o = 0; o < O; ++o)
        for (size_t i = 0; i < I; ++i)
          for (size_t k = 0; k < Kk; ++k) (*out)[(o * I + i) * Kk + k] = c.data[(o * Kk + k) * I + i];
      return true;
    }
    if (size_t(have[0]) == Kk && size_t(have[1]) == I && size_t(have[2]) == O) {  // TF filter [k][in][out]
      for (size_t o = 0; o < O; ++o)
        for (size_t i = 0; i < I; ++i)
          for (size_t k = 0; k < Kk; ++k) (*out)[(o * I + i) * Kk + k] = c.data[(k * I + i) * O + o];
      return true;
    }
  }
  return false;
}

bool name_matches(const std::string& tensor_name, const std::string& param, const std::string& graph) {
  // `param` = "encoder.blocks.0.attn.query.weight"; inside the encoder graph the path is "blocks.0.attn.query.weight"
  const std::string local = param.substr(graph.size() + 1);
  size_t p = tensor_name.find(local);
  while (p != std::string::npos) {
    const bool left_ok = p == 0 || !(std::isalnum(static_cast<unsigned char>(tensor_name[p - 1])) || tensor_name[p - 1] == '_');
    const size_t e = p + local.size();
    const bool right_ok = e == tensor_name.size() || !(std::isalnum(static_cast<unsigned char>(tensor_name[e])) || tensor_name[e] == '_');
    // "attn.query.weight" must not match inside "cross_attn.query.weight": the character before is '_' there
    if (left_ok && right_ok) return true;
    p = tensor_name.find(local, p + 1);
  }
  return false;
}

void assign_graph(std::vector<Constant>& consts, const std::string& graph, std::vector<wtw::NamedTensor>& tensors,
                  const std::string& path, std::vector<std::string>* report) {
  // (1) by name
  for (wtw::NamedTensor& t : tensors) {
    if (t.name.compare(0, graph.size() + 1, graph + ".") != 0 || !t.data.empty()) continue;
    for (Constant& c : consts) {
      if (c.used || !name_matches(c.name, t.name, graph)) continue;
      if (to_torch_layout(c, t.shape, &t.data)) {
        c.used = true;
        if (report) report->push_back(t.name + " <- \"" + c.name + "\" (rule 1: name)");
        break;
      }
    }
  }
  // (2) by order of first use — operator index, then position among that operator's inputs (a FULLY_CONNECTED carries
  // its weight at input 1 and its bias at input 2) — and element count
  std::vector<Constant*> order;
  for (Constant& c : consts)
    if (!c.used && c.first_use < (1 << 30)) order.push_back(&c);
  std::stable_sort(order.begin(), order.end(), [](const Constant* a, const Constant* b) {
    return a->first_use != b->first_use ? a->first_use < b->first_use : a->first_use_slot < b->first_use_slot;
  });
  for (wtw::NamedTensor& t : tensors) {
    if (t.name.compare(0, graph.size() + 1, graph + ".") != 0 || !t.data.empty()) continue;
    bool found = false;
    for (size_t i = 0; i < order.size() && !found; ++i) {
      if (order[i]->used || order[i]->numel() != numel_of(t.shape)) continue;
      // Parameters of equal size are told apart by first-use order only.  Two unnamed candidates that ONE operator
      // reads (a folded LayerNorm's gain and shift, a fused q|k|v) have no order to go by: refuse rather than guess.
      for (size_t j = i + 1; j < order.size(); ++j) {
        if (order[j]->used || order[j]->numel() != order[i]->numel()) continue;
        if (order[j]->first_use == order[i]->first_use) {
          throw Error(kErrFormat, path + ": cannot tell which constant is " + t.name + ": operator #" +
                                      std::to_string(order[i]->first_use) + " reads several unnamed constants of " +
                                      std::to_string(order[i]->numel()) + " elements (\"" + order[i]->name + "\", \"" +
                                      order[j]->name + "\")");
        }
        break;  // the next candidate of this size belongs to a later operator
      }
      if (to_torch_layout(*order[i], t.shape, &t.data)) {
        order[i]->used = true;
        found = true;
        if (report) {
          report->push_back(t.name + " <- \"" + order[i]->name + "\" (rule 2: first use, operator #" +
                            std::to_string(order[i]->first_use) + " input " + std::to_string(order[i]->first_use_slot) + ")");
        }
      }
    }
    if (!found) throw Error(kErrFormat, path + ": no constant found for " + t.name);
  }
}

}  // namespace

void convert_tflite(const std::string& model_prefix, const std::string& out_path) {
  const std::string enc_path = model_prefix + ".encoder.tflite", dec_path = model_prefix + ".decoder.tflite";
  std::vector<Constant> enc = read_constants(enc_path), dec = read_constants(dec_path);
  // the architecture follows from two sizes: conv1 (d x 80 x 3 elements, the encoder's only constant of that
  // count) and the token embedding (n_vocab x d, the decoder's largest constant)
  wtw::Dims dims{};
  bool known = false;
  for (const char* arch : {"tiny", "tiny.en", "base", "micro"}) {
    wtw::Dims d;
    if (!wtw::dims_by_name(arch, &d)) continue;
    const size_t conv1 = size_t(d.n_audio_state) * d.n_mels * 3, emb = size_t(d.n_vocab) * d.n_text_state;
    const size_t fc1 = size_t(4) * d.n_audio_state * d.n_audio_state;
    bool c1 = false, e1 = false, f1 = false;
    for (const Constant& c : enc) c1 = c1 || c.numel() == conv1, f1 = f1 || c.numel() == fc1;
    for (const Constant& c : dec) e1 = e1 || c.numel() == emb;
    if (c1 && e1 && f1) {
      dims = d;
      known = true;
      break;
    }
  }
  if (!known) {
    throw Error(kErrFormat, model_prefix + ": the .tflite pair matches none of the supported architectures (tiny, tiny.en, base)");
  }
  std::vector<wtw::NamedTensor> tensors = wtw::tensor_specs(dims);
  // WT_VERBOSE: which rule matched each parameter (a real converter file orders and names constants its own way;
  // this mapping is what to read first when a transcript comes out as garbage)
  std::vector<std::string> report;
  const bool verbose = getenv("WT_VERBOSE") != nullptr;
  assign_graph(enc, "encoder", tensors, enc_path, verbose ? &report : nullptr);
  assign_graph(dec, "decoder", tensors, dec_path, verbose ? &report : nullptr);
  for (const std::string& line : report) std::fprintf(stderr, "[wt-convert] %s\n", line.c_str());
  std::string err;
  const int rc = wtw::write_tensors(out_path.c_str(), dims, tensors, 0, &err);
  if (rc != 0) throw Error(rc == 1 ? kErrFormat : kErrIo, err);
}

}  // namespace wt

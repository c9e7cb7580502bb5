// See model_files.h.  Field numbers are those of the public onnx.proto3 schema and are quoted next to each use; nothing in a
// model file is executed — bytes in, float32 arrays out.
#include "model_files.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

namespace pfhip_files {
namespace {

[[noreturn]] void bad(const std::string& msg) { throw FormatError(msg); }

// ---- protobuf wire format -----------------------------------------------------------------------------------------------
struct Span {
  const uint8_t* p = nullptr;
  size_t n = 0;
};

uint64_t varint(const uint8_t* buf, size_t end, size_t& pos) {
  uint64_t result = 0;
  for (int shift = 0;; shift += 7) {
    if (pos >= end) bad("truncated varint");
    if (shift > 63) bad("varint longer than 10 bytes");
    const uint8_t b = buf[pos++];
    result |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return result;
  }
}

struct Field {
  uint32_t no = 0;
  int wt = 0;
  uint64_t v = 0;      // varint / fixed
  Span s;              // length-delimited
};

// iterates the fields of one message
class Fields {
 public:
  explicit Fields(Span m) : m_(m) {}
  bool next(Field& f) {
    if (pos_ >= m_.n) return false;
    const uint64_t key = varint(m_.p, m_.n, pos_);
    f.no = (uint32_t)(key >> 3);
    f.wt = (int)(key & 7);
    if (f.no == 0) bad("field number 0");
    switch (f.wt) {
      case 0: f.v = varint(m_.p, m_.n, pos_); break;
      case 1:
        if (pos_ + 8 > m_.n) bad("truncated fixed64");
        std::memcpy(&f.v, m_.p + pos_, 8);
        pos_ += 8;
        break;
      case 2: {
        const uint64_t n = varint(m_.p, m_.n, pos_);
        if (n > m_.n - pos_) bad("length-delimited field runs past its message");
        f.s = Span{m_.p + pos_, (size_t)n};
        pos_ += (size_t)n;
        break;
      }
      case 5: {
        if (pos_ + 4 > m_.n) bad("truncated fixed32");
        uint32_t u;
        std::memcpy(&u, m_.p + pos_, 4);
        f.v = u;
        pos_ += 4;
        break;
      }
      default: bad("unsupported wire type " + std::to_string(f.wt) + " (groups are not used by ONNX)");
    }
    return true;
  }

 private:
  Span m_;
  size_t pos_ = 0;
};

std::string text(Span s) { return std::string(reinterpret_cast<const char*>(s.p), s.n); }

void ints(const Field& f, std::vector<int64_t>& out) {      // repeated int64: packed (wire type 2) or one varint per key
  if (f.wt == 2) {
    size_t pos = 0;
    while (pos < f.s.n) out.push_back((int64_t)varint(f.s.p, f.s.n, pos));
  } else {
    out.push_back((int64_t)f.v);
  }
}

size_t dtype_size(int dt) {
  switch (dt) {
    case 1: case 6: case 12: return 4;      // FLOAT INT32 UINT32
    case 2: case 3: case 9: return 1;       // UINT8 INT8 BOOL
    case 4: case 5: case 10: case 16: return 2;   // UINT16 INT16 FLOAT16 BFLOAT16
    case 7: case 11: case 13: return 8;     // INT64 DOUBLE UINT64
    default: return 0;
  }
}

// TensorProto: 1 dims, 2 data_type, 4 float_data, 5 int32_data, 7 int64_data, 8 name, 9 raw_data, 10 double_data, 11 uint64_data,
// 13 external_data, 14 data_location
Initializer tensor(Span buf) {
  Initializer t;
  Fields it(buf);
  Field f;
  while (it.next(f)) {
    switch (f.no) {
      case 1: ints(f, t.dims); break;
      case 2: t.dtype = (int)f.v; break;
      case 8: t.name = text(f.s); break;
      case 9: t.raw = f.s.p; t.raw_bytes = f.s.n; break;
      case 4:
        if (f.wt == 2) {
          const size_t n = f.s.n / 4, at = t.f32.size();
          t.f32.resize(at + n);
          std::memcpy(t.f32.data() + at, f.s.p, n * 4);
        } else {
          const uint32_t u = (uint32_t)f.v;
          float x;
          std::memcpy(&x, &u, 4);
          t.f32.push_back(x);
        }
        break;
      case 5: case 7: case 11: ints(f, t.ints); break;
      case 10:
        if (f.wt == 2) {
          const size_t n = f.s.n / 8, at = t.f64.size();
          t.f64.resize(at + n);
          std::memcpy(t.f64.data() + at, f.s.p, n * 8);
        } else {
          double x;
          std::memcpy(&x, &f.v, 8);
          t.f64.push_back(x);
        }
        break;
      case 13: t.external = true; break;
      case 14: if (f.v == 1) t.external = true; break;
      default: break;
    }
  }
  for (int64_t d : t.dims)
    if (d < 0) bad("tensor " + t.name + ": negative dimension");
  if (t.external) return t;
  const size_t n = t.count();
  if (!dtype_size(t.dtype)) bad("tensor " + t.name + ": unsupported data type " + std::to_string(t.dtype));
  if (t.raw) {
    if (t.raw_bytes != n * dtype_size(t.dtype))
      bad("tensor " + t.name + ": raw_data holds " + std::to_string(t.raw_bytes) + " bytes, its dims need " + std::to_string(n * dtype_size(t.dtype)));
  } else {
    const size_t have = !t.f32.empty() ? t.f32.size() : !t.f64.empty() ? t.f64.size() : t.ints.size();
    if (have != n) bad("tensor " + t.name + ": " + std::to_string(have) + " values for " + std::to_string(n) + " elements");
  }
  return t;
}

// AttributeProto: 1 name, 3 i (the only attributes the weight walk needs: transB, hidden_size)
void attribute(Span buf, Node& nd) {
  std::string name;
  bool has_i = false;
  int64_t iv = 0;
  Fields it(buf);
  Field f;
  while (it.next(f)) {
    if (f.no == 1 && f.wt == 2) name = text(f.s);
    else if (f.no == 3 && f.wt == 0) { iv = (int64_t)f.v; has_i = true; }
  }
  if (has_i) nd.int_attrs[name] = iv;
}

// NodeProto: 1 input, 2 output, 3 name, 4 op_type, 5 attribute
Node node(Span buf) {
  Node nd;
  Fields it(buf);
  Field f;
  while (it.next(f)) {
    if (f.wt != 2) continue;
    switch (f.no) {
      case 1: nd.inputs.push_back(text(f.s)); break;
      case 2: nd.outputs.push_back(text(f.s)); break;
      case 3: nd.name = text(f.s); break;
      case 4: nd.op_type = text(f.s); break;
      case 5: attribute(f.s, nd); break;
      default: break;
    }
  }
  return nd;
}

std::string value_info_name(Span buf) {      // ValueInfoProto: 1 name
  Fields it(buf);
  Field f;
  while (it.next(f))
    if (f.no == 1 && f.wt == 2) return text(f.s);
  return "";
}

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, u;
  if (exp == 0) {
    if (man == 0) u = sign;
    else {                                   // subnormal: normalise
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400));
      u = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FF) << 13;
    }
  } else if (exp == 31) {
    u = sign | 0x7F800000u | man << 13;
  } else {
    u = sign | (exp + 127 - 15) << 23 | man << 13;
  }
  float x;
  std::memcpy(&x, &u, 4);
  return x;
}

// element i of an initializer as double (integers, halves, floats)
double element(const Initializer& t, size_t i) {
  if (t.raw) {
    const uint8_t* p = t.raw + i * dtype_size(t.dtype);
    switch (t.dtype) {
      case 1: { float x; std::memcpy(&x, p, 4); return x; }
      case 2: case 9: return *p;
      case 3: return (int8_t)*p;
      case 4: { uint16_t x; std::memcpy(&x, p, 2); return x; }
      case 5: { int16_t x; std::memcpy(&x, p, 2); return x; }
      case 6: { int32_t x; std::memcpy(&x, p, 4); return x; }
      case 7: { int64_t x; std::memcpy(&x, p, 8); return (double)x; }
      case 10: { uint16_t x; std::memcpy(&x, p, 2); return half_to_float(x); }
      case 11: { double x; std::memcpy(&x, p, 8); return x; }
      case 12: { uint32_t x; std::memcpy(&x, p, 4); return x; }
      case 13: { uint64_t x; std::memcpy(&x, p, 8); return (double)x; }
      case 16: { uint16_t x; std::memcpy(&x, p, 2); uint32_t u = (uint32_t)x << 16; float y; std::memcpy(&y, &u, 4); return y; }
      default: return 0;
    }
  }
  if (!t.f32.empty()) return t.f32[i];
  if (!t.f64.empty()) return t.f64[i];
  if (t.dtype == 10) return half_to_float((uint16_t)(t.ints[i] & 0xFFFF));      // float16 travels as its bit pattern in int32_data
  if (t.dtype == 16) { uint32_t u = (uint32_t)(t.ints[i] & 0xFFFF) << 16; float y; std::memcpy(&y, &u, 4); return y; }
  return (double)t.ints[i];
}

bool is_float_type(int dt) { return dt == 1 || dt == 10 || dt == 11 || dt == 16; }

// float32 Array of an initializer: a view when it is raw float32, otherwise converted
Array as_array(const Initializer& t) {
  Array a;
  a.dims = t.dims;
  if (t.dtype == 1 && t.raw) {
    a.view = t.raw;
    return a;
  }
  a.own = std::make_shared<std::vector<float>>(t.count());
  for (size_t i = 0; i < a.own->size(); ++i) (*a.own)[i] = (float)element(t, i);
  return a;
}

Array transposed2d(const Array& a) {
  if (a.dims.size() != 2) bad("transpose of a tensor that is not 2-d");
  Array t = a;
  t.dims = {a.dims[1], a.dims[0]};
  t.transposed = !a.transposed;
  return t;
}

bool ends_with(const std::string& s, const std::string& sfx) {
  return s.size() >= sfx.size() && s.compare(s.size() - sfx.size(), sfx.size(), sfx) == 0;
}
bool starts_with(const std::string& s, const std::string& pfx) { return s.compare(0, pfx.size(), pfx) == 0; }

// '/encoder/encoders.3/feed_forward/w_1/MatMul' -> 'encoder.encoders.3.feed_forward.w_1' (the torch exporter names nodes after
// the module path); "" when the node name carries no path
std::string module_of_node(const std::string& name) {
  if (name.empty() || name[0] != '/') return "";
  std::vector<std::string> parts;
  std::stringstream ss(name);
  std::string p;
  while (std::getline(ss, p, '/'))
    if (!p.empty()) parts.push_back(p);
  if (parts.size() < 2) return "";
  std::string out;
  for (size_t i = 0; i + 1 < parts.size(); ++i) out += (i ? "." : "") + parts[i];
  return out;
}

bool is_anonymous(const std::string& name) {
  if (starts_with(name, "onnx::") || starts_with(name, "_v_") || name.find("::") != std::string::npos) return true;
  return !name.empty() && std::all_of(name.begin(), name.end(), [](char c) { return c >= '0' && c <= '9'; });
}

void put(State& state, const std::string& key, const Array& a) {
  auto it = state.find(key);
  if (it != state.end()) {
    std::vector<float> x(a.count()), y(it->second.count());
    if (x.size() == y.size()) {
      a.copy_to(x.data());
      it->second.copy_to(y.data());
      if (it->second.dims == a.dims && std::memcmp(x.data(), y.data(), x.size() * 4) == 0) return;
    }
    bad("two different tensors resolve to " + key);
  }
  state.emplace(key, a);
}

std::vector<char> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  f.seekg(0, std::ios::end);
  const std::streamoff n = f.tellg();
  f.seekg(0);
  std::vector<char> out((size_t)n);
  if (n && !f.read(out.data(), n)) throw std::runtime_error("cannot read " + path);
  return out;
}

bool exists(const std::string& path) {
  struct stat st;
  return ::stat(path.c_str(), &st) == 0;
}

}  // namespace

size_t Initializer::count() const {
  size_t n = 1;
  for (int64_t d : dims) n *= (size_t)d;
  return n;
}

size_t Array::count() const {
  size_t n = 1;
  for (int64_t d : dims) n *= (size_t)d;
  return n;
}

void Array::copy_to(float* dst) const {
  const size_t n = count();
  if (!view) {
    const float* src = own->data();
    if (!transposed) { std::memcpy(dst, src, n * 4); return; }
    const size_t R = (size_t)dims[0], C = (size_t)dims[1];
    for (size_t r = 0; r < R; ++r)
      for (size_t c = 0; c < C; ++c) dst[r * C + c] = src[c * R + r];
    return;
  }
  if (!transposed) { std::memcpy(dst, view, n * 4); return; }
  // [C, R] in the file -> [R, C]: 32 x 32 blocks so that both sides stay in cache (a 2048 x 512 weight: ~1 ms)
  const size_t R = (size_t)dims[0], C = (size_t)dims[1];
  for (size_t r0 = 0; r0 < R; r0 += 32)
    for (size_t c0 = 0; c0 < C; c0 += 32) {
      const size_t r1 = std::min(R, r0 + 32), c1 = std::min(C, c0 + 32);
      for (size_t c = c0; c < c1; ++c) {
        const uint8_t* src = view + (c * R + r0) * 4;
        for (size_t r = r0; r < r1; ++r, src += 4) std::memcpy(&dst[r * C + c], src, 4);
      }
    }
}

// ModelProto: 1 ir_version, 2 producer_name, 7 graph;  GraphProto: 1 node, 5 initializer, 11 input, 12 output
void read_onnx_bytes(const void* data, size_t bytes, OnnxModel& m) {
  if (m.image.empty() || m.image.data() != data) {
    m.image.assign(static_cast<const char*>(data), static_cast<const char*>(data) + bytes);
  }
  const Span whole{reinterpret_cast<const uint8_t*>(m.image.data()), m.image.size()};
  Span graph;
  {
    Fields it(whole);
    Field f;
    while (it.next(f)) {
      if (f.no == 1 && f.wt == 0) m.ir_version = (int64_t)f.v;
      else if (f.no == 2 && f.wt == 2) m.producer = text(f.s);
      else if (f.no == 7 && f.wt == 2) graph = f.s;
    }
  }
  if (!graph.p) bad("no graph in the ModelProto (not an ONNX file?)");
  Fields it(graph);
  Field f;
  while (it.next(f)) {
    if (f.wt != 2) continue;
    if (f.no == 1) m.nodes.push_back(node(f.s));
    else if (f.no == 5) {
      Initializer t = tensor(f.s);
      if (t.external) m.external.push_back(t.name);
      else m.initializers[t.name] = std::move(t);
    } else if (f.no == 11) m.inputs.push_back(value_info_name(f.s));
    else if (f.no == 12) m.outputs.push_back(value_info_name(f.s));
  }
}

void read_onnx(const std::string& path, OnnxModel& m) {
  m.image = read_file(path);
  try {
    read_onnx_bytes(m.image.data(), m.image.size(), m);
  } catch (const FormatError& e) {
    throw FormatError(path + ": " + e.what());
  }
}

std::vector<std::string> check_closed(const OnnxModel& m) {
  std::set<std::string> known(m.inputs.begin(), m.inputs.end());
  for (const auto& kv : m.initializers) known.insert(kv.first);
  known.insert(m.external.begin(), m.external.end());
  known.insert("");
  std::vector<std::string> out;
  for (const Node& nd : m.nodes) {
    for (const std::string& x : nd.inputs)
      if (!known.count(x)) out.push_back(nd.op_type + ":" + nd.name + ":" + x);
    known.insert(nd.outputs.begin(), nd.outputs.end());
  }
  for (const std::string& o : m.outputs)
    if (!known.count(o)) out.push_back("output:" + o);
  return out;
}

void torch_style_state(const OnnxModel& m, State& state) {
  // float view of every initializer, dequantised ones added under their base name:  W = (W_q - zero_point) * scale, per tensor or
  // per output column (onnxruntime quantize_dynamic leaves <W>_quantized / <W>_scale / <W>_zero_point)
  std::map<std::string, Array> init;
  for (const auto& kv : m.initializers)
    if (is_float_type(kv.second.dtype)) init[kv.first] = as_array(kv.second);
  for (const auto& kv : m.initializers) {
    const std::string& k = kv.first;
    if (!ends_with(k, "_quantized")) continue;
    const std::string base = k.substr(0, k.size() - 10);
    auto sc = m.initializers.find(base + "_scale"), zp = m.initializers.find(base + "_zero_point");
    if (sc == m.initializers.end() || zp == m.initializers.end()) continue;
    const Initializer& q = kv.second;
    Array a;
    a.dims = q.dims;
    a.own = std::make_shared<std::vector<float>>(q.count());
    const size_t ns = sc->second.count(), nz = zp->second.count();
    const size_t cols = q.dims.empty() ? 1 : (size_t)q.dims.back();
    if ((ns != 1 && ns != cols) || (nz != 1 && nz != cols)) bad(base + ": scale / zero point neither per tensor nor per column");
    for (size_t i = 0; i < a.own->size(); ++i) {
      const double s = element(sc->second, ns == 1 ? 0 : i % cols), z = element(zp->second, nz == 1 ? 0 : i % cols);
      (*a.own)[i] = (float)((float)((int)element(q, i) - (int)z) * (float)s);
    }
    init[base] = a;
  }
  std::map<std::string, std::vector<const Node*>> consumers;
  for (const Node& nd : m.nodes)
    for (const std::string& x : nd.inputs) consumers[x].push_back(&nd);
  std::set<std::string> used;

  for (const Node& nd : m.nodes) {
    if (nd.op_type == "MatMul" || nd.op_type == "MatMulInteger" || nd.op_type == "DynamicQuantizeMatMul") {
      const std::string wname = nd.inputs.size() > 1 ? nd.inputs[1] : "";
      const std::string base = ends_with(wname, "_quantized") ? wname.substr(0, wname.size() - 10) : wname;
      auto wi = init.find(base);
      if (wi == init.end() || wi->second.dims.size() != 2) continue;
      const Array& w = wi->second;
      std::string module;
      // bias sibling: MatMul -> [Cast / Mul (quantised paths)] -> Add(named bias)
      std::vector<std::string> frontier = nd.outputs;
      for (int hops = 0; !frontier.empty() && hops < 4 && module.empty(); ++hops) {
        std::vector<std::string> nxt;
        for (const std::string& o : frontier) {
          auto ci = consumers.find(o);
          if (ci == consumers.end()) continue;
          for (const Node* c : ci->second) {
            if (c->op_type == "Add") {
              for (const std::string& x : c->inputs) {
                auto bi = m.initializers.find(x);
                if (bi != m.initializers.end() && ends_with(x, ".bias") && bi->second.dims.size() == 1 && bi->second.dims[0] == w.dims[1])
                  module = x.substr(0, x.size() - 5);
              }
            } else if (c->op_type == "Cast" || c->op_type == "Mul") {
              nxt.insert(nxt.end(), c->outputs.begin(), c->outputs.end());
            }
          }
        }
        frontier = nxt;
      }
      if (module.empty()) module = module_of_node(ends_with(nd.name, "_quant") ? nd.name.substr(0, nd.name.size() - 6) : nd.name);
      if (module.empty() && !is_anonymous(base)) module = ends_with(base, ".weight") ? base.substr(0, base.size() - 7) : base;
      if (module.empty()) continue;
      put(state, module + ".weight", transposed2d(w));
      used.insert({wname, base, base + "_scale", base + "_zero_point"});
    } else if (nd.op_type == "Gemm" && nd.inputs.size() > 1 && init.count(nd.inputs[1])) {
      if (!is_anonymous(nd.inputs[1])) continue;          // named: passes through below in its stored layout
      std::string module;
      if (nd.inputs.size() > 2 && ends_with(nd.inputs[2], ".bias")) module = nd.inputs[2].substr(0, nd.inputs[2].size() - 5);
      if (module.empty()) module = module_of_node(nd.name);
      if (module.empty()) continue;
      auto tb = nd.int_attrs.find("transB");
      const Array& w = init[nd.inputs[1]];
      put(state, module + ".weight", tb != nd.int_attrs.end() && tb->second ? w : transposed2d(w));
      used.insert(nd.inputs[1]);
    } else if (nd.op_type == "LSTM" && nd.inputs.size() > 2 && init.count(nd.inputs[1]) && init.count(nd.inputs[2])) {
      // ONNX LSTM: W [dirs, 4h, in], R [dirs, 4h, h], B [dirs, 8h] = Wb | Rb, gate order i, o, f, c; torch: i, f, g(c), o
      std::string module = module_of_node(nd.name);
      if (module.empty())
        for (int k = 1; k <= 2; ++k)
          if (!is_anonymous(nd.inputs[k])) module = nd.inputs[k].substr(0, nd.inputs[k].rfind('.'));
      if (module.empty()) continue;
      const Array& Wm = init[nd.inputs[1]];
      const Array& Rm = init[nd.inputs[2]];
      const Array* Bm = nd.inputs.size() > 3 && !nd.inputs[3].empty() && init.count(nd.inputs[3]) ? &init[nd.inputs[3]] : nullptr;
      if (Wm.dims.size() != 3 || Rm.dims.size() != 3 || Wm.dims[1] % 4) bad("LSTM " + module + ": unexpected W / R shapes");
      const size_t dirs = (size_t)Wm.dims[0], h = (size_t)Rm.dims[2], in = (size_t)Wm.dims[2];
      if (dirs < 1 || dirs > 2 || h == 0 || (size_t)Wm.dims[1] != 4 * h || (size_t)Rm.dims[0] != dirs || (size_t)Rm.dims[1] != 4 * h ||
          (Bm && Bm->count() != dirs * 8 * h))
        bad("LSTM " + module + ": W / R / B shapes do not agree");
      std::vector<float> Wv(Wm.count()), Rv(Rm.count()), Bv(Bm ? Bm->count() : 0);
      Wm.copy_to(Wv.data());
      Rm.copy_to(Rv.data());
      if (Bm) Bm->copy_to(Bv.data());
      auto torch_gates = [h](const float* a, size_t width, std::vector<int64_t> dims) {      // rows (i, o, f, c) -> (i, f, c, o)
        Array out;
        out.dims = std::move(dims);
        out.own = std::make_shared<std::vector<float>>(4 * h * width);
        const int order[4] = {0, 2, 3, 1};
        for (int g = 0; g < 4; ++g) std::memcpy(out.own->data() + (size_t)g * h * width, a + (size_t)order[g] * h * width, h * width * 4);
        return out;
      };
      for (size_t d = 0; d < dirs; ++d) {
        const std::string sfx = d == 1 ? "_reverse" : "";
        put(state, module + ".weight_ih_l0" + sfx, torch_gates(Wv.data() + d * 4 * h * in, in, {(int64_t)(4 * h), (int64_t)in}));
        put(state, module + ".weight_hh_l0" + sfx, torch_gates(Rv.data() + d * 4 * h * h, h, {(int64_t)(4 * h), (int64_t)h}));
        if (Bm) {
          put(state, module + ".bias_ih_l0" + sfx, torch_gates(Bv.data() + d * 8 * h, 1, {(int64_t)(4 * h)}));
          put(state, module + ".bias_hh_l0" + sfx, torch_gates(Bv.data() + d * 8 * h + 4 * h, 1, {(int64_t)(4 * h)}));
        }
      }
      for (size_t k = 1; k < nd.inputs.size() && k < 4; ++k) used.insert(nd.inputs[k]);
    }
  }
  for (const auto& kv : init) {
    const std::string& k = kv.first;
    if (used.count(k) || is_anonymous(k) || ends_with(k, "_quantized") || ends_with(k, "_scale") || ends_with(k, "_zero_point")) continue;
    if (!state.count(k)) state.emplace(k, kv.second);
  }
}

// ---- am.mvn ---------------------------------------------------------------------------------------------------------------
// kaldi-nnet text: the row after `<AddShift>` / `<Rescale>` starts `<LearnRateCoef> 0 [` and ends `]`; LoadCmvn keeps the
// values in between (paraformer.cpp:325-360: items 3 .. size-2 of the whitespace split)
void parse_am_mvn(const std::string& text_, std::vector<float>& shift, std::vector<float>& rescale) {
  shift.clear();
  rescale.clear();
  std::vector<std::vector<std::string>> rows;
  std::stringstream ss(text_);
  std::string line;
  while (std::getline(ss, line)) {
    std::stringstream ls(line);
    std::vector<std::string> items;
    std::string w;
    while (ls >> w) items.push_back(w);
    rows.push_back(items);
  }
  bool got_shift = false, got_scale = false;
  for (size_t i = 0; i + 1 < rows.size(); ++i) {
    if (rows[i].empty() || rows[i + 1].empty() || rows[i + 1][0] != "<LearnRateCoef>") continue;
    std::vector<float>* dst = rows[i][0] == "<AddShift>" ? &shift : rows[i][0] == "<Rescale>" ? &rescale : nullptr;
    if (!dst) continue;
    const auto& r = rows[i + 1];
    for (size_t j = 3; j + 1 < r.size(); ++j) {
      char* end = nullptr;
      const float v = std::strtof(r[j].c_str(), &end);
      if (end == r[j].c_str()) bad("am.mvn: '" + r[j] + "' is not a number");
      dst->push_back(v);
    }
    (dst == &shift ? got_shift : got_scale) = true;
  }
  if (!got_shift || !got_scale) bad("am.mvn: <AddShift>/<Rescale> rows not found");
}

// ---- YAML subset -----------------------------------------------------------------------------------------------------------
// Block mappings and sequences by indentation, flow sequences / mappings on one logical line, quoted and plain scalars, comments.
// No anchors, tags, multi-document streams or block scalars beyond skipping them: what FunASR's config.yaml files use.
namespace {

struct YLine {
  int indent;
  std::string s;
};

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
  return s.substr(a, b - a);
}

std::string strip_comment(const std::string& s) {
  char q = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (q) {
      if (c == q) q = 0;
      else if (c == '\\' && q == '"') ++i;
    } else if (c == '"' || c == '\'') {
      q = c;
    } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
      return s.substr(0, i);
    }
  }
  return s;
}

YNode scalar(const std::string& raw) {
  YNode n;
  std::string s = trim(raw);
  if (s.empty() || s == "~" || s == "null") return n;
  n.kind = YNode::SCALAR;
  if (s.size() >= 2 && s.front() == '"' && s.back() == '"') {
    std::string out;
    for (size_t i = 1; i + 1 < s.size(); ++i) {
      if (s[i] == '\\' && i + 2 < s.size()) {
        ++i;
        out += s[i] == 'n' ? '\n' : s[i] == 't' ? '\t' : s[i];
      } else {
        out += s[i];
      }
    }
    n.s = out;
  } else if (s.size() >= 2 && s.front() == '\'' && s.back() == '\'') {
    std::string out;
    for (size_t i = 1; i + 1 < s.size(); ++i) {
      out += s[i];
      if (s[i] == '\'' && s[i + 1] == '\'') ++i;
    }
    n.s = out;
  } else {
    n.s = s;
  }
  return n;
}

// position of the ':' that ends a mapping key on this line (followed by blank or end), outside quotes and brackets
size_t key_colon(const std::string& s) {
  char q = 0;
  int depth = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (q) { if (c == q) q = 0; continue; }
    if (c == '"' || c == '\'') { q = c; continue; }
    if (c == '[' || c == '{') ++depth;
    else if (c == ']' || c == '}') --depth;
    else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ' || s[i + 1] == '\t')) return i;
  }
  return std::string::npos;
}

YNode flow(const std::string& s, size_t& i);

void skip_ws(const std::string& s, size_t& i) {
  while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
}

std::string flow_token(const std::string& s, size_t& i, const char* stops) {
  skip_ws(s, i);
  const size_t a = i;
  if (i < s.size() && (s[i] == '"' || s[i] == '\'')) {
    const char q = s[i++];
    while (i < s.size() && s[i] != q) i += (s[i] == '\\' && q == '"') ? 2 : 1;
    if (i < s.size()) ++i;
    return s.substr(a, i - a);
  }
  while (i < s.size() && !std::strchr(stops, s[i])) ++i;
  return trim(s.substr(a, i - a));
}

YNode flow(const std::string& s, size_t& i) {
  skip_ws(s, i);
  YNode n;
  if (i < s.size() && s[i] == '[') {
    n.kind = YNode::SEQ;
    ++i;
    for (;;) {
      skip_ws(s, i);
      if (i >= s.size()) bad("yaml: unterminated [");
      if (s[i] == ']') { ++i; break; }
      if (s[i] == '[' || s[i] == '{') n.seq.push_back(flow(s, i));
      else n.seq.push_back(scalar(flow_token(s, i, ",]")));
      skip_ws(s, i);
      if (i < s.size() && s[i] == ',') ++i;
    }
  } else if (i < s.size() && s[i] == '{') {
    n.kind = YNode::MAP;
    ++i;
    for (;;) {
      skip_ws(s, i);
      if (i >= s.size()) bad("yaml: unterminated {");
      if (s[i] == '}') { ++i; break; }
      const std::string k = scalar(flow_token(s, i, ":,}")).s;
      skip_ws(s, i);
      YNode v;
      if (i < s.size() && s[i] == ':') {
        ++i;
        skip_ws(s, i);
        if (i < s.size() && (s[i] == '[' || s[i] == '{')) v = flow(s, i);
        else v = scalar(flow_token(s, i, ",}"));
      }
      n.map.emplace_back(k, v);
      skip_ws(s, i);
      if (i < s.size() && s[i] == ',') ++i;
    }
  } else {
    n = scalar(s.substr(i));
    i = s.size();
  }
  return n;
}

class YParser {
 public:
  explicit YParser(const std::string& text_) {
    std::stringstream ss(text_);
    std::string raw;
    while (std::getline(ss, raw)) {
      std::string s = strip_comment(raw);
      size_t ind = 0;
      while (ind < s.size() && s[ind] == ' ') ++ind;
      s = trim(s);
      if (s.empty() || s == "---" || s == "...") continue;
      lines_.push_back({(int)ind, s});
    }
    // a flow collection that spans lines becomes one logical line
    for (size_t i = 0; i < lines_.size(); ++i) {
      while (open_brackets(lines_[i].s) > 0 && i + 1 < lines_.size()) {
        lines_[i].s += " " + lines_[i + 1].s;
        lines_.erase(lines_.begin() + (long)i + 1);
      }
    }
  }
  YNode parse() {
    if (lines_.empty()) return YNode();
    size_t i = 0;
    return block(i, lines_[0].indent);
  }

 private:
  std::vector<YLine> lines_;

  static int open_brackets(const std::string& s) {
    int depth = 0;
    char q = 0;
    for (const char c : s) {
      if (q) { if (c == q) q = 0; continue; }
      if (c == '"' || c == '\'') q = c;
      else if (c == '[' || c == '{') ++depth;
      else if (c == ']' || c == '}') --depth;
    }
    return depth;
  }
  static bool is_item(const std::string& s) { return s == "-" || starts_with(s, "- "); }

  // value after "key:" / "- ": inline scalar or flow collection, a block scalar (skipped), or a nested block
  YNode value(const std::string& rest, size_t& i, int parent_indent, bool parent_is_map) {
    const std::string r = trim(rest);
    if (!r.empty() && (r[0] == '|' || r[0] == '>')) {          // block scalar: its lines are joined
      YNode n;
      n.kind = YNode::SCALAR;
      while (i < lines_.size() && lines_[i].indent > parent_indent) n.s += (n.s.empty() ? "" : "\n") + lines_[i++].s;
      return n;
    }
    if (!r.empty()) {
      size_t k = 0;
      return flow(r, k);
    }
    if (i < lines_.size()) {
      const YLine& nx = lines_[i];
      if (nx.indent > parent_indent) return block(i, nx.indent);
      if (parent_is_map && nx.indent == parent_indent && is_item(nx.s)) return block(i, nx.indent);   // "key:\n- a\n- b"
    }
    return YNode();
  }

  YNode block(size_t& i, int indent) {
    YNode n;
    if (is_item(lines_[i].s)) {
      n.kind = YNode::SEQ;
      while (i < lines_.size() && lines_[i].indent == indent && is_item(lines_[i].s)) {
        const std::string rest = lines_[i].s == "-" ? "" : trim(lines_[i].s.substr(2));
        const size_t colon = rest.empty() || rest[0] == '[' || rest[0] == '{' || rest[0] == '"' || rest[0] == '\'' ? std::string::npos : key_colon(rest);
        if (colon != std::string::npos) {
          // "- key: value" opens a mapping whose further keys are indented past the dash
          const int inner = indent + 2 + (int)(lines_[i].s.size() - 2 - rest.size());
          lines_[i] = {inner, rest};
          n.seq.push_back(block(i, inner));
        } else {
          ++i;
          n.seq.push_back(value(rest, i, indent, false));
        }
      }
      return n;
    }
    n.kind = YNode::MAP;
    while (i < lines_.size() && lines_[i].indent == indent && !is_item(lines_[i].s)) {
      const std::string& s = lines_[i].s;
      const size_t colon = key_colon(s);
      if (colon == std::string::npos) bad("yaml: expected 'key: value' in '" + s + "'");
      const std::string key = scalar(s.substr(0, colon)).s;
      const std::string rest = s.substr(colon + 1);
      ++i;
      n.map.emplace_back(key, value(rest, i, indent, true));
    }
    if (i < lines_.size() && lines_[i].indent > indent) bad("yaml: unexpected indentation at '" + lines_[i].s + "'");
    return n;
  }
};

}  // namespace

const YNode* YNode::get(const std::string& key) const {
  for (const auto& kv : map)
    if (kv.first == key) return &kv.second;
  return nullptr;
}
double YNode::number(const std::string& key, double dflt) const {
  const YNode* v = get(key);
  if (!v || v->kind != SCALAR) return dflt;
  char* end = nullptr;
  const double x = std::strtod(v->s.c_str(), &end);
  return end == v->s.c_str() ? dflt : x;
}
std::string YNode::str(const std::string& key, const std::string& dflt) const {
  const YNode* v = get(key);
  return v && v->kind == SCALAR ? v->s : dflt;
}
YNode parse_yaml(const std::string& text_) { return YParser(text_).parse(); }

// ---- containers ------------------------------------------------------------------------------------------------------------
namespace {

constexpr size_t kAlign = 256;      // bytes, weights.py ALIGN

struct Spec {
  std::string name;
  std::vector<int64_t> shape;
};

struct SpecList {
  std::vector<Spec> v;
  void add(const std::string& n, std::vector<int64_t> s) { v.push_back({n, std::move(s)}); }
  void lin(const std::string& n, int64_t out_f, int64_t in_f, bool bias = true) {
    add(n + ".w", {out_f, in_f});
    if (bias) add(n + ".b", {out_f});
  }
  void ln(const std::string& n, int64_t d) {
    add(n + ".g", {d});
    add(n + ".b", {d});
  }
};

struct AsrConfig {      // weights.py PARAFORMER_LARGE
  int d_model = 512, n_head = 4, ffn = 2048, enc_layers = 50, dec_layers = 16, dec_ffn = 2048, kernel = 11, vocab = 8404, n_mels = 80,
      lfr_m = 7, lfr_n = 6, pred_residual = 0, contextual = 0, timestamp = 0, fs = 16000, upsample_times = 3;
  double cif_threshold = 1.0, tail_threshold = 0.45, smooth_factor = 1.0, noise_threshold = 0.0, smooth_factor2 = 1.0, noise_threshold2 = 0.0;
  bool has_sf2 = false, has_nt2 = false;
  std::string lang;
};

// weights.py tensor_specs, same order
SpecList asr_specs(const AsrConfig& c) {
  SpecList s;
  const int64_t d = c.d_model, f = c.ffn, fd = c.dec_ffn, V = c.vocab, k = c.kernel, feat = (int64_t)c.n_mels * c.lfr_m;
  s.add("cmvn.mean", {feat});
  s.add("cmvn.istd", {feat});
  for (int i = 0; i < c.enc_layers; ++i) {
    const std::string p = "enc." + std::to_string(i) + ".";
    const int64_t in_f = i == 0 ? feat : d;
    s.ln(p + "norm1", in_f);
    s.lin(p + "qkv", 3 * d, in_f);
    s.add(p + "fsmn.w", {d, k});
    s.lin(p + "out", d, d);
    s.ln(p + "norm2", d);
    s.lin(p + "ffn1", f, d);
    s.lin(p + "ffn2", d, f);
  }
  s.ln("enc.after_norm", d);
  s.add("pred.conv.w", {d, d, 3});
  s.add("pred.conv.b", {d});
  s.add("pred.out.w", {1, d});
  s.add("pred.out.b", {1});
  for (int i = 0; i < c.dec_layers; ++i) {
    const std::string p = "dec." + std::to_string(i) + ".";
    s.ln(p + "norm1", d);
    s.lin(p + "ffn1", fd, d);
    s.ln(p + "ffn_norm", fd);
    s.lin(p + "ffn2", d, fd, false);
    s.ln(p + "norm2", d);
    s.add(p + "fsmn.w", {d, k});
    s.ln(p + "norm3", d);
    s.lin(p + "q", d, d);
    s.lin(p + "kv", 2 * d, d);
    s.lin(p + "out", d, d);
  }
  s.ln("dec3.norm1", d);
  s.lin("dec3.ffn1", fd, d);
  s.ln("dec3.ffn_norm", fd);
  s.lin("dec3.ffn2", d, fd, false);
  s.ln("dec.after_norm", d);
  s.lin("dec.out", V, d);
  if (c.timestamp) {
    s.add("pred.up.w", {d, d, 3});
    s.add("pred.up.b", {d});
    for (const char* sfx : {"", "_r"}) {
      s.add(std::string("pred.blstm.w_ih") + sfx, {4 * d, d});
      s.add(std::string("pred.blstm.w_hh") + sfx, {4 * d, d});
      s.add(std::string("pred.blstm.b_ih") + sfx, {4 * d});
      s.add(std::string("pred.blstm.b_hh") + sfx, {4 * d});
    }
    s.add("pred.out2.w", {1, 2 * d});
    s.add("pred.out2.b", {1});
  }
  if (c.contextual) {
    s.add("bias.embed.w", {V, d});
    s.add("bias.lstm.w_ih", {4 * d, d});
    s.add("bias.lstm.w_hh", {4 * d, d});
    s.add("bias.lstm.b_ih", {4 * d});
    s.add("bias.lstm.b_hh", {4 * d});
    s.ln("bias.dec.norm3", d);
    s.lin("bias.dec.q", d, d);
    s.lin("bias.dec.kv", 2 * d, d);
    s.lin("bias.dec.out", d, d);
    s.add("bias.out.w", {d, 2 * d});
  }
  return s;
}

using NameMap = std::map<std::string, std::string>;

void map_lin(NameMap& m, const std::string& dst, const std::string& src, bool bias = true) {
  m[dst + ".w"] = src + ".weight";
  if (bias) m[dst + ".b"] = src + ".bias";
}
void map_ln(NameMap& m, const std::string& dst, const std::string& src) {
  m[dst + ".g"] = src + ".weight";
  m[dst + ".b"] = src + ".bias";
}

// convert.py paraformer_name_map: container tensor -> UPSTREAM FunASR state_dict key (from memory of
// funasr/models/{sanm,paraformer,contextual_paraformer,bicif_paraformer}; no such file is available offline)
NameMap asr_name_map(const AsrConfig& c) {
  NameMap m;
  for (int i = 0; i < c.enc_layers; ++i) {
    const std::string src = i == 0 ? "encoder.encoders0.0" : "encoder.encoders." + std::to_string(i - 1);
    const std::string p = "enc." + std::to_string(i) + ".";
    map_ln(m, p + "norm1", src + ".norm1");
    map_lin(m, p + "qkv", src + ".self_attn.linear_q_k_v");
    m[p + "fsmn.w"] = src + ".self_attn.fsmn_block.weight";
    map_lin(m, p + "out", src + ".self_attn.linear_out");
    map_ln(m, p + "norm2", src + ".norm2");
    map_lin(m, p + "ffn1", src + ".feed_forward.w_1");
    map_lin(m, p + "ffn2", src + ".feed_forward.w_2");
  }
  map_ln(m, "enc.after_norm", "encoder.after_norm");
  m["pred.conv.w"] = "predictor.cif_conv1d.weight";
  m["pred.conv.b"] = "predictor.cif_conv1d.bias";
  map_lin(m, "pred.out", "predictor.cif_output");
  for (int i = 0; i < c.dec_layers; ++i) {
    std::string src = "decoder.decoders." + std::to_string(i);
    if (c.contextual && i == c.dec_layers - 1) src = "decoder.last_decoder";
    const std::string p = "dec." + std::to_string(i) + ".";
    map_ln(m, p + "norm1", src + ".norm1");
    map_lin(m, p + "ffn1", src + ".feed_forward.w_1");
    map_ln(m, p + "ffn_norm", src + ".feed_forward.norm");
    map_lin(m, p + "ffn2", src + ".feed_forward.w_2", false);
    map_ln(m, p + "norm2", src + ".norm2");
    m[p + "fsmn.w"] = src + ".self_attn.fsmn_block.weight";
    map_ln(m, p + "norm3", src + ".norm3");
    map_lin(m, p + "q", src + ".src_attn.linear_q");
    map_lin(m, p + "kv", src + ".src_attn.linear_k_v");
    map_lin(m, p + "out", src + ".src_attn.linear_out");
  }
  map_ln(m, "dec3.norm1", "decoder.decoders3.0.norm1");
  map_lin(m, "dec3.ffn1", "decoder.decoders3.0.feed_forward.w_1");
  map_ln(m, "dec3.ffn_norm", "decoder.decoders3.0.feed_forward.norm");
  map_lin(m, "dec3.ffn2", "decoder.decoders3.0.feed_forward.w_2", false);
  map_ln(m, "dec.after_norm", "decoder.after_norm");
  map_lin(m, "dec.out", "decoder.output_layer");
  if (c.timestamp) {
    m["pred.up.w"] = "predictor.upsample_cnn.weight";
    m["pred.up.b"] = "predictor.upsample_cnn.bias";
    for (const auto& pr : {std::make_pair("", ""), std::make_pair("_r", "_reverse")}) {
      m[std::string("pred.blstm.w_ih") + pr.first] = std::string("predictor.blstm.weight_ih_l0") + pr.second;
      m[std::string("pred.blstm.w_hh") + pr.first] = std::string("predictor.blstm.weight_hh_l0") + pr.second;
      m[std::string("pred.blstm.b_ih") + pr.first] = std::string("predictor.blstm.bias_ih_l0") + pr.second;
      m[std::string("pred.blstm.b_hh") + pr.first] = std::string("predictor.blstm.bias_hh_l0") + pr.second;
    }
    map_lin(m, "pred.out2", "predictor.cif_output2");
  }
  if (c.contextual) {
    m["bias.embed.w"] = "bias_embed.weight";
    m["bias.lstm.w_ih"] = "bias_encoder.weight_ih_l0";
    m["bias.lstm.w_hh"] = "bias_encoder.weight_hh_l0";
    m["bias.lstm.b_ih"] = "bias_encoder.bias_ih_l0";
    m["bias.lstm.b_hh"] = "bias_encoder.bias_hh_l0";
    map_ln(m, "bias.dec.norm3", "decoder.bias_decoder.norm3");
    map_lin(m, "bias.dec.q", "decoder.bias_decoder.src_attn.linear_q");
    map_lin(m, "bias.dec.kv", "decoder.bias_decoder.src_attn.linear_k_v");
    map_lin(m, "bias.dec.out", "decoder.bias_decoder.src_attn.linear_out");
    m["bias.out.w"] = "decoder.bias_output.weight";
  }
  return m;
}

struct VadConfig {      // weights.py FSMN_VAD
  int n_mels = 80, lfr_m = 5, lfr_n = 1, input_dim = 400, affine = 140, linear = 250, proj = 128, lorder = 20, layers = 4, out_affine = 140,
      n_out = 248, fs = 16000, max_end_silence_time = 800, max_single_segment_time = 60000;
  double speech_noise_thres = 0.9;
};

SpecList vad_specs(const VadConfig& c) {
  SpecList s;
  s.add("cmvn.mean", {c.input_dim});
  s.add("cmvn.istd", {c.input_dim});
  s.lin("in1", c.affine, c.input_dim);
  s.lin("in2", c.linear, c.affine);
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = "blk." + std::to_string(i) + ".";
    s.lin(p + "linear", c.proj, c.linear, false);
    s.add(p + "fsmn.w", {c.proj, c.lorder});
    s.lin(p + "affine", c.linear, c.proj);
  }
  s.lin("out1", c.out_affine, c.linear);
  s.lin("out2", c.n_out, c.out_affine);
  return s;
}

NameMap vad_name_map(const VadConfig& c) {
  NameMap m;
  for (const auto& pr : {std::make_pair("in1", "encoder.in_linear1.linear"), std::make_pair("in2", "encoder.in_linear2.linear"),
                         std::make_pair("out1", "encoder.out_linear1.linear"), std::make_pair("out2", "encoder.out_linear2.linear")})
    map_lin(m, pr.first, pr.second);
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = "blk." + std::to_string(i), src = "encoder.fsmn." + std::to_string(i);
    m[p + ".linear.w"] = src + ".linear.linear.weight";
    m[p + ".fsmn.w"] = src + ".fsmn_block.conv_left.weight";      // [proj, 1, lorder, 1]
    m[p + ".affine.w"] = src + ".affine.linear.weight";
    m[p + ".affine.b"] = src + ".affine.linear.bias";
  }
  return m;
}

struct PuncConfig {     // weights.py CT_TRANSFORMER
  int vocab = 272727, d_model = 256, n_head = 8, ffn = 1024, layers = 4, kernel = 11, n_punc = 6, sanm_shift = 0;
  std::vector<std::string> punc_list;
};

SpecList punc_specs(const PuncConfig& c) {
  SpecList s;
  const int64_t d = c.d_model;
  s.add("embed.w", {c.vocab, d});
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = "enc." + std::to_string(i) + ".";
    s.ln(p + "norm1", d);
    s.lin(p + "qkv", 3 * d, d);
    s.add(p + "fsmn.w", {d, c.kernel});
    s.lin(p + "out", d, d);
    s.ln(p + "norm2", d);
    s.lin(p + "ffn1", c.ffn, d);
    s.lin(p + "ffn2", d, c.ffn);
  }
  s.ln("enc.after_norm", d);
  s.lin("out", c.n_punc, d);
  return s;
}

NameMap punc_name_map(const PuncConfig& c) {
  NameMap m;
  m["embed.w"] = "embed.weight";
  m["out.w"] = "decoder.weight";
  m["out.b"] = "decoder.bias";
  map_ln(m, "enc.after_norm", "encoder.after_norm");
  for (int i = 0; i < c.layers; ++i) {
    const std::string src = i == 0 ? "encoder.encoders0.0" : "encoder.encoders." + std::to_string(i - 1);
    const std::string p = "enc." + std::to_string(i) + ".";
    map_ln(m, p + "norm1", src + ".norm1");
    map_ln(m, p + "norm2", src + ".norm2");
    map_lin(m, p + "qkv", src + ".self_attn.linear_q_k_v");
    map_lin(m, p + "out", src + ".self_attn.linear_out");
    map_lin(m, p + "ffn1", src + ".feed_forward.w_1");
    map_lin(m, p + "ffn2", src + ".feed_forward.w_2");
    m[p + "fsmn.w"] = src + ".self_attn.fsmn_block.weight";
  }
  return m;
}

// FunASR's export wrappers keep the original module under `.model` (encoder.model.encoders0.0...): a key is looked up as
// written and with every `model.` path component dropped.
std::string drop_model_components(const std::string& key) {
  std::string out;
  std::stringstream ss(key);
  std::string part;
  bool first = true;
  while (std::getline(ss, part, '.')) {
    if (part == "model") continue;
    out += (first ? "" : ".") + part;
    first = false;
  }
  return out;
}

const Array* find_key(const State& state, const std::map<std::string, std::string>& normalised, const std::string& key) {
  auto it = state.find(key);
  if (it != state.end()) return &it->second;
  auto nt = normalised.find(key);
  if (nt != normalised.end()) return &state.at(nt->second);
  return nullptr;
}

std::map<std::string, std::string> normalised_keys(const State& state) {
  std::map<std::string, std::string> out;
  for (const auto& kv : state) {
    const std::string n = drop_model_components(kv.first);
    if (n != kv.first && !state.count(n)) out.emplace(n, kv.first);
  }
  return out;
}

std::string json_escape(const std::string& s) {
  std::string out;
  for (const unsigned char c : s) {
    if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
    else if (c == '\n') out += "\\n";
    else if (c == '\t') out += "\\t";
    else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); out += b; }
    else out += (char)c;
  }
  return out;
}

std::string num(double v) {
  char b[40];
  if (v == std::floor(v) && std::fabs(v) < 1e15) std::snprintf(b, sizeof b, "%.1f", v);
  else std::snprintf(b, sizeof b, "%.17g", v);
  return b;
}

// lays the specs out (256-byte aligned, weights.py build_manifest) and fills them from `extra` or the state through the name map;
// a tensor is reshaped when only singleton dims differ ([d,1,k] -> [d,k])
void fill(const SpecList& specs, const NameMap& names, const State& state, const std::map<std::string, std::vector<float>>& extra,
          const std::string& config_json, Container& out) {
  std::vector<size_t> offs;
  size_t off = 0;
  for (const Spec& sp : specs.v) {
    offs.push_back(off);
    size_t n = 1;
    for (int64_t d : sp.shape) n *= (size_t)d;
    off += (n * 4 + kAlign - 1) / kAlign * kAlign;
  }
  out.blob.assign(off / 4, 0.f);
  const auto normalised = normalised_keys(state);
  std::vector<std::string> missing;
  std::set<std::string> consumed;
  std::string tensors;
  for (size_t i = 0; i < specs.v.size(); ++i) {
    const Spec& sp = specs.v[i];
    size_t n = 1;
    std::string shape;
    for (int64_t d : sp.shape) {
      n *= (size_t)d;
      shape += (shape.empty() ? "" : ", ") + std::to_string(d);
    }
    tensors += (i ? ", " : "") + std::string("\"") + sp.name + "\": {\"shape\": [" + shape + "], \"offset\": " + std::to_string(offs[i]) + "}";
    float* dst = out.blob.data() + offs[i] / 4;
    auto ex = extra.find(sp.name);
    if (ex != extra.end()) {
      if (ex->second.size() != n) bad(sp.name + ": " + std::to_string(ex->second.size()) + " values, the model needs " + std::to_string(n));
      std::memcpy(dst, ex->second.data(), n * 4);
      continue;
    }
    auto nm = names.find(sp.name);
    const Array* a = nm == names.end() ? nullptr : find_key(state, normalised, nm->second);
    if (!a) {
      missing.push_back(sp.name + " <- " + (nm == names.end() ? "?" : nm->second));
      continue;
    }
    consumed.insert(nm->second);
    std::vector<int64_t> have, want;
    for (int64_t d : a->dims) if (d != 1) have.push_back(d);
    for (int64_t d : sp.shape) if (d != 1) want.push_back(d);
    if (a->count() != n || have != want) {
      std::string got;
      for (int64_t d : a->dims) got += (got.empty() ? "" : ", ") + std::to_string(d);
      bad(sp.name + ": the file's shape [" + got + "] does not fit [" + shape + "]");
    }
    a->copy_to(dst);
  }
  if (!missing.empty()) {
    std::string msg = "the model files lack " + std::to_string(missing.size()) + " tensors: ";
    for (size_t i = 0; i < missing.size() && i < 8; ++i) msg += (i ? "; " : "") + missing[i];
    if (missing.size() > 8) msg += " ...";
    // what IS there under a similar path helps to spot a renamed module
    std::string near;
    int shown = 0;
    for (const auto& kv : state) {
      if (consumed.count(kv.first) || consumed.count(drop_model_components(kv.first))) continue;
      near += (shown ? ", " : "") + kv.first;
      if (++shown == 8) break;
    }
    if (shown) msg += "  (unmatched keys in the files: " + near + (state.size() > consumed.size() + 8 ? ", ..." : "") + ")";
    bad(msg);
  }
  out.manifest = "{\"config\": " + config_json + ", \"tensors\": {" + tensors + "}, \"total_bytes\": " + std::to_string(off) + "}";
}

std::string dir_of(const std::string& path) {
  const size_t s = path.rfind('/');
  return s == std::string::npos ? "." : path.substr(0, s);
}
std::string base_of(const std::string& path) {
  const size_t s = path.rfind('/');
  return s == std::string::npos ? path : path.substr(s + 1);
}

// The reference hands over <dir>/model.onnx, model_quant.onnx, decoder.onnx, model_eb.onnx ... — or, with use_gpu, the
// TorchScript name of the same model (offline-stream.cpp:79-84, :63-71).  TorchScript archives are pickles and are not opened:
// the ONNX file of the same stem beside it is read (model.torchscript -> model.onnx, else model_quant.onnx).
std::string resolve_onnx(const std::string& path) {
  if (path.empty()) return path;
  if (ends_with(path, ".onnx") && exists(path)) return path;
  const std::string dir = dir_of(path);
  std::string stem = base_of(path);
  const size_t dot = stem.find('.');
  if (dot != std::string::npos) stem = stem.substr(0, dot);
  for (const char* sfx : {"_blade", "_quant"})
    if (ends_with(stem, sfx)) stem = stem.substr(0, stem.size() - std::strlen(sfx));
  for (const std::string& cand : {dir + "/" + stem + ".onnx", dir + "/" + stem + "_quant.onnx"})
    if (exists(cand)) return cand;
  throw std::runtime_error("no ONNX file for " + path + " (looked for " + stem + ".onnx / " + stem + "_quant.onnx in " + dir + ")");
}

bool load_container_pair(const std::string& model, const std::string& config, Container& out);

// A directory that holds no ONNX file but a container converted earlier (tools/convert_funasr.py, or this reader's own cache
// after the sources were removed): <dir>/<stem>.pfhip.{bin,json}, or the names the stand-alone harnesses of rounds 1-3 used.
bool preconverted(const std::string& model, const char* legacy_stem, Container& out) {
  try {
    (void)resolve_onnx(model);
    return false;
  } catch (const std::runtime_error&) {
    const std::string dir = dir_of(model);
    std::string stem = base_of(model);
    const size_t dot = stem.find('.');
    if (dot != std::string::npos) stem = stem.substr(0, dot);
    for (const std::string& cand : {dir + "/" + stem + ".pfhip.bin", dir + "/model.pfhip.bin", dir + "/" + legacy_stem + ".pfhip.bin"})
      if (exists(cand) && exists(cand.substr(0, cand.size() - 4) + ".json")) return load_container_pair(cand, "", out);
    throw;
  }
}

std::string slurp(const std::string& path) {
  const std::vector<char> v = read_file(path);
  return std::string(v.begin(), v.end());
}

void add_file_state(const std::string& path, State& state, std::vector<OnnxModel>& keep, Container& out) {
  keep.emplace_back();
  OnnxModel& m = keep.back();
  read_onnx(path, m);
  if (!m.external.empty())
    bad(path + ": " + std::to_string(m.external.size()) + " initializers live in external data files (not supported)");
  const std::vector<std::string> open = check_closed(m);
  if (!open.empty()) bad(path + ": graph is not closed over its initializers, e.g. " + open[0]);
  State st;
  torch_style_state(m, st);
  for (auto& kv : st) state[kv.first] = kv.second;      // later files win on a clash
  out.sources.push_back(path);
}

// ---- cache: <dir>/<stem>.pfhip.{bin,json} written on the first load, reused while every source keeps its size and mtime ----------
struct SourceStamp {
  std::string name;
  long long bytes = 0, mtime_ns = 0;
};

bool stamp_of(const std::string& path, SourceStamp& s) {
  struct stat st;
  if (::stat(path.c_str(), &st) != 0) return false;
  s.name = base_of(path);
  s.bytes = (long long)st.st_size;
  s.mtime_ns = (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
  return true;
}

constexpr int kLoaderVersion = 1;

std::string sources_json(const std::vector<std::string>& files) {
  std::string s = "{\"loader\": " + std::to_string(kLoaderVersion) + ", \"files\": [";
  bool first = true;
  for (const std::string& f : files) {
    SourceStamp st;
    if (!stamp_of(f, st)) continue;
    s += (first ? "" : ", ") + std::string("[\"") + json_escape(st.name) + "\", " + std::to_string(st.bytes) + ", " + std::to_string(st.mtime_ns) + "]";
    first = false;
  }
  return s + "]}";
}

bool cache_enabled() {
  const char* e = std::getenv("PFHIP_MODEL_CACHE");
  return !(e && *e == '0');
}

bool try_cache(const std::string& prefix, const std::vector<std::string>& files, Container& out) {
  if (!cache_enabled() || !exists(prefix + ".bin") || !exists(prefix + ".json")) return false;
  try {
    const std::string man = slurp(prefix + ".json");
    const std::string want = "\"sources\": " + sources_json(files);
    if (man.find(want) == std::string::npos) return false;
    const std::vector<char> raw = read_file(prefix + ".bin");
    const size_t tb = man.rfind("\"total_bytes\": ");
    if (tb == std::string::npos || (size_t)std::strtoull(man.c_str() + tb + 15, nullptr, 10) != raw.size()) return false;
    out.blob.resize(raw.size() / 4);
    std::memcpy(out.blob.data(), raw.data(), out.blob.size() * 4);
    out.manifest = man;
    out.sources = files;
    out.from_cache = true;
    return true;
  } catch (const std::exception&) {
    return false;
  }
}

// the manifest gains a "sources" member (ignored by the library) so that a later load can tell whether the files changed
void finish(const std::string& prefix, const std::vector<std::string>& files, Container& out) {
  const size_t tb = out.manifest.rfind(", \"total_bytes\"");
  out.manifest.insert(tb, ", \"sources\": " + sources_json(files));
  if (!cache_enabled()) return;
  const std::string tmp_bin = prefix + ".bin.tmp" + std::to_string((long)::getpid()), tmp_json = prefix + ".json.tmp" + std::to_string((long)::getpid());
  {
    std::ofstream fb(tmp_bin, std::ios::binary);
    if (!fb) return;                                  // a read-only model directory: no cache, not an error
    fb.write(reinterpret_cast<const char*>(out.blob.data()), (std::streamsize)(out.blob.size() * 4));
    std::ofstream fj(tmp_json, std::ios::binary);
    fj << out.manifest;
    if (!fb || !fj) { std::remove(tmp_bin.c_str()); std::remove(tmp_json.c_str()); return; }
  }
  if (std::rename(tmp_bin.c_str(), (prefix + ".bin").c_str()) != 0 || std::rename(tmp_json.c_str(), (prefix + ".json").c_str()) != 0) {
    std::remove(tmp_bin.c_str());
    std::remove(tmp_json.c_str());
  }
}

std::string cache_prefix(const std::string& onnx_path) {
  std::string stem = base_of(onnx_path);
  stem = stem.substr(0, stem.size() - 5);             // ".onnx"
  return dir_of(onnx_path) + "/" + stem + ".pfhip";
}

// a container file pair: model = x.pfhip.bin, config = its manifest (or x.pfhip.json beside it)
bool load_container_pair(const std::string& model, const std::string& config, Container& out) {
  if (!ends_with(model, ".pfhip.bin")) return false;
  const std::vector<char> raw = read_file(model);
  out.blob.resize(raw.size() / 4);
  std::memcpy(out.blob.data(), raw.data(), out.blob.size() * 4);
  const std::string man = ends_with(config, ".json") && exists(config) ? config : model.substr(0, model.size() - 4) + ".json";
  out.manifest = slurp(man);
  out.sources = {model, man};
  return true;
}

int64_t dim_of(const State& state, const std::string& key, size_t axis, const char* what) {
  const auto normalised = normalised_keys(state);
  const Array* a = find_key(state, normalised, key);
  if (!a || a->dims.size() <= axis) bad(std::string("cannot size ") + what + ": " + key + " is not in the model files");
  return a->dims[axis];
}

bool has_key(const State& state, const std::string& key) {
  if (state.count(key)) return true;
  for (const auto& kv : state)
    if (drop_model_components(kv.first) == key) return true;
  return false;
}

}  // namespace

void load_asr(const std::string& model, const std::string& second, const std::string& hotword, const std::string& cmvn,
              const std::string& config, Container& out) {
  out = Container();
  if (load_container_pair(model, config, out) || preconverted(model, "model", out)) return;
  std::vector<std::string> onnx = {resolve_onnx(model)};
  if (!second.empty()) onnx.push_back(resolve_onnx(second));
  if (!hotword.empty()) onnx.push_back(resolve_onnx(hotword));
  std::vector<std::string> files = onnx;
  files.push_back(cmvn);
  files.push_back(config);
  const std::string prefix = cache_prefix(onnx[0]);
  if (try_cache(prefix, files, out)) return;

  // config.yaml (LoadConfigFromYaml / LoadOnlineConfigFromYaml, paraformer.cpp:178-241; the architecture keys are what the
  // exported graph was traced with)
  const YNode y = parse_yaml(slurp(config));
  const YNode none;
  auto sect = [&](const char* k) { const YNode* n = y.get(k); return n && n->kind == YNode::MAP ? *n : none; };
  const YNode enc = sect("encoder_conf"), dec = sect("decoder_conf"), pred = sect("predictor_conf"), fe = sect("frontend_conf");
  AsrConfig c;
  c.d_model = (int)enc.number("output_size", c.d_model);
  c.n_head = (int)enc.number("attention_heads", c.n_head);
  c.ffn = (int)enc.number("linear_units", c.ffn);
  c.enc_layers = (int)enc.number("num_blocks", c.enc_layers);
  c.kernel = (int)enc.number("kernel_size", c.kernel);
  c.dec_layers = (int)dec.number("att_layer_num", dec.number("num_blocks", c.dec_layers));
  c.dec_ffn = (int)dec.number("linear_units", c.dec_ffn);
  c.cif_threshold = pred.number("threshold", c.cif_threshold);
  c.tail_threshold = pred.number("tail_threshold", c.tail_threshold);
  c.smooth_factor = pred.number("smooth_factor", c.smooth_factor);
  c.noise_threshold = pred.number("noise_threshold", c.noise_threshold);
  c.has_sf2 = pred.get("smooth_factor2") != nullptr;
  c.has_nt2 = pred.get("noise_threshold2") != nullptr;
  c.smooth_factor2 = pred.number("smooth_factor2", c.smooth_factor2);
  c.noise_threshold2 = pred.number("noise_threshold2", c.noise_threshold2);
  c.upsample_times = (int)pred.number("upsample_times", c.upsample_times);
  c.n_mels = (int)fe.number("n_mels", c.n_mels);
  c.lfr_m = (int)fe.number("lfr_m", c.lfr_m);
  c.lfr_n = (int)fe.number("lfr_n", c.lfr_n);
  c.fs = (int)fe.number("fs", c.fs);
  c.lang = y.str("lang", "");

  std::vector<OnnxModel> keep;
  keep.reserve(onnx.size());
  State state;
  for (const std::string& p : onnx) add_file_state(p, state, keep, out);
  // the heads that are there decide (config.yaml names the model class, the weights decide)
  c.contextual = has_key(state, "decoder.bias_output.weight") || has_key(state, "bias_embed.weight") ? 1 : 0;
  c.timestamp = has_key(state, "predictor.upsample_cnn.weight") ? 1 : 0;
  c.vocab = (int)dim_of(state, "decoder.output_layer.weight", 0, "the vocabulary");

  std::vector<float> shift, rescale;
  parse_am_mvn(slurp(cmvn), shift, rescale);
  out.sources.push_back(cmvn);
  out.sources.push_back(config);

  std::string cj = "{\"d_model\": " + std::to_string(c.d_model) + ", \"n_head\": " + std::to_string(c.n_head) + ", \"ffn\": " + std::to_string(c.ffn) +
                   ", \"enc_layers\": " + std::to_string(c.enc_layers) + ", \"dec_layers\": " + std::to_string(c.dec_layers) +
                   ", \"dec_ffn\": " + std::to_string(c.dec_ffn) + ", \"kernel\": " + std::to_string(c.kernel) + ", \"vocab\": " + std::to_string(c.vocab) +
                   ", \"n_mels\": " + std::to_string(c.n_mels) + ", \"lfr_m\": " + std::to_string(c.lfr_m) + ", \"lfr_n\": " + std::to_string(c.lfr_n) +
                   ", \"cif_threshold\": " + num(c.cif_threshold) + ", \"tail_threshold\": " + num(c.tail_threshold) +
                   ", \"smooth_factor\": " + num(c.smooth_factor) + ", \"noise_threshold\": " + num(c.noise_threshold) +
                   ", \"pred_residual\": 0, \"contextual\": " + std::to_string(c.contextual) + ", \"timestamp\": " + std::to_string(c.timestamp) +
                   ", \"fs\": " + std::to_string(c.fs);
  if (c.timestamp) {
    if (c.has_sf2) cj += ", \"smooth_factor2\": " + num(c.smooth_factor2);
    if (c.has_nt2) cj += ", \"noise_threshold2\": " + num(c.noise_threshold2);
    cj += ", \"upsample_times\": " + std::to_string(c.upsample_times);
  }
  if (!c.lang.empty()) cj += ", \"lang\": \"" + json_escape(c.lang) + "\"";
  cj += "}";
  fill(asr_specs(c), asr_name_map(c), state, {{"cmvn.mean", shift}, {"cmvn.istd", rescale}}, cj, out);
  finish(prefix, files, out);
}

void load_vad(const std::string& model, const std::string& cmvn, const std::string& config, Container& out) {
  out = Container();
  if (load_container_pair(model, config, out) || preconverted(model, "vad", out)) return;
  const std::string onnx = resolve_onnx(model);
  const std::vector<std::string> files = {onnx, cmvn, config};
  const std::string prefix = cache_prefix(onnx);
  if (try_cache(prefix, files, out)) return;
  // FsmnVad::LoadConfigFromYaml (fsmn-vad.cpp:21-50): frontend_conf.fs, model_conf.{max_end_silence_time,
  // max_single_segment_time, speech_noise_thres}; encoder_conf sizes the FSMN (UPSTREAM key names)
  const YNode y = parse_yaml(slurp(config));
  const YNode none;
  auto sect = [&](const char* k) { const YNode* n = y.get(k); return n && n->kind == YNode::MAP ? *n : none; };
  const YNode fe = sect("frontend_conf"), post = sect("model_conf"), enc = sect("encoder_conf");
  VadConfig c;
  c.fs = (int)fe.number("fs", c.fs);
  c.n_mels = (int)fe.number("n_mels", c.n_mels);
  c.lfr_m = (int)fe.number("lfr_m", c.lfr_m);
  c.lfr_n = (int)fe.number("lfr_n", c.lfr_n);
  c.max_end_silence_time = (int)post.number("max_end_silence_time", c.max_end_silence_time);
  c.max_single_segment_time = (int)post.number("max_single_segment_time", c.max_single_segment_time);
  c.speech_noise_thres = post.number("speech_noise_thres", c.speech_noise_thres);
  c.layers = (int)enc.number("fsmn_layers", c.layers);
  std::vector<OnnxModel> keep;
  keep.reserve(1);
  State state;
  add_file_state(onnx, state, keep, out);
  // every width from the tensors themselves
  c.input_dim = (int)dim_of(state, "encoder.in_linear1.linear.weight", 1, "the VAD input");
  c.affine = (int)dim_of(state, "encoder.in_linear1.linear.weight", 0, "the VAD input affine");
  c.linear = (int)dim_of(state, "encoder.in_linear2.linear.weight", 0, "the VAD linear width");
  c.proj = (int)dim_of(state, "encoder.fsmn.0.linear.linear.weight", 0, "the VAD projection");
  c.lorder = (int)dim_of(state, "encoder.fsmn.0.fsmn_block.conv_left.weight", 2, "the VAD memory order");
  c.out_affine = (int)dim_of(state, "encoder.out_linear1.linear.weight", 0, "the VAD output affine");
  c.n_out = (int)dim_of(state, "encoder.out_linear2.linear.weight", 0, "the VAD classes");
  std::vector<float> shift, rescale;
  parse_am_mvn(slurp(cmvn), shift, rescale);
  out.sources.push_back(cmvn);
  out.sources.push_back(config);
  const std::string cj = "{\"model\": \"fsmn_vad\", \"n_mels\": " + std::to_string(c.n_mels) + ", \"lfr_m\": " + std::to_string(c.lfr_m) +
                         ", \"lfr_n\": " + std::to_string(c.lfr_n) + ", \"input_dim\": " + std::to_string(c.input_dim) + ", \"affine\": " + std::to_string(c.affine) +
                         ", \"linear\": " + std::to_string(c.linear) + ", \"proj\": " + std::to_string(c.proj) + ", \"lorder\": " + std::to_string(c.lorder) +
                         ", \"layers\": " + std::to_string(c.layers) + ", \"out_affine\": " + std::to_string(c.out_affine) + ", \"n_out\": " + std::to_string(c.n_out) +
                         ", \"fs\": " + std::to_string(c.fs) + ", \"max_end_silence_time\": " + std::to_string(c.max_end_silence_time) +
                         ", \"max_single_segment_time\": " + std::to_string(c.max_single_segment_time) + ", \"speech_noise_thres\": " + num(c.speech_noise_thres) + "}";
  fill(vad_specs(c), vad_name_map(c), state, {{"cmvn.mean", shift}, {"cmvn.istd", rescale}}, cj, out);
  finish(prefix, files, out);
}

void load_punc(const std::string& model, const std::string& config, Container& out) {
  out = Container();
  if (load_container_pair(model, config, out) || preconverted(model, "punc", out)) return;
  const std::string onnx = resolve_onnx(model);
  const std::vector<std::string> files = {onnx, config};
  const std::string prefix = cache_prefix(onnx);
  if (try_cache(prefix, files, out)) return;
  // CTokenizer::OpenYaml(config, token_file) reads model_conf.punc_list (tokenizer.cpp:130-183); encoder_conf sizes the SAN-M
  const YNode y = parse_yaml(slurp(config));
  const YNode none;
  auto sect = [&](const YNode& from, const char* k) { const YNode* n = from.get(k); return n && n->kind == YNode::MAP ? *n : none; };
  const YNode enc = sect(y, "encoder_conf"), mc = sect(y, "model_conf");
  PuncConfig c;
  c.d_model = (int)enc.number("output_size", c.d_model);
  c.n_head = (int)enc.number("attention_heads", c.n_head);
  c.ffn = (int)enc.number("linear_units", c.ffn);
  c.layers = (int)enc.number("num_blocks", c.layers);
  c.kernel = (int)enc.number("kernel_size", c.kernel);
  c.sanm_shift = (int)enc.number("sanm_shfit", enc.number("sanm_shift", c.sanm_shift));      // (sic) the upstream key is misspelt
  const YNode* pl = mc.get("punc_list");
  if (!pl) pl = y.get("punc_list");
  if (pl && pl->kind == YNode::SEQ)
    for (const YNode& e : pl->seq)
      if (e.kind == YNode::SCALAR) c.punc_list.push_back(e.s);
  std::vector<OnnxModel> keep;
  keep.reserve(1);
  State state;
  add_file_state(onnx, state, keep, out);
  c.vocab = (int)dim_of(state, "embed.weight", 0, "the punctuation vocabulary");
  c.n_punc = (int)dim_of(state, "decoder.weight", 0, "the punctuation classes");
  out.sources.push_back(config);
  std::string cj = "{\"model\": \"ct_transformer\", \"vocab\": " + std::to_string(c.vocab) + ", \"d_model\": " + std::to_string(c.d_model) +
                   ", \"n_head\": " + std::to_string(c.n_head) + ", \"ffn\": " + std::to_string(c.ffn) + ", \"layers\": " + std::to_string(c.layers) +
                   ", \"kernel\": " + std::to_string(c.kernel) + ", \"n_punc\": " + std::to_string(c.n_punc) + ", \"sanm_shift\": " + std::to_string(c.sanm_shift);
  if (!c.punc_list.empty()) {
    cj += ", \"punc_list\": [";
    for (size_t i = 0; i < c.punc_list.size(); ++i) cj += (i ? ", " : "") + std::string("\"") + json_escape(c.punc_list[i]) + "\"";
    cj += "]";
  }
  cj += "}";
  fill(punc_specs(c), punc_name_map(c), state, {}, cj, out);
  finish(prefix, files, out);
}

}  // namespace pfhip_files

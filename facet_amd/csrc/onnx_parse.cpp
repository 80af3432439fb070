// Protobuf wire-format reader for onnx.ModelProto (field numbers from the public onnx.proto3 schema) -> fe::onnx::Model.
// The files it reads are the reference's own model files (insightface buffalo_l/*.onnx, analyzers/face.py:30-38); every length,
// count and shape in them is checked before it is used. Plain C++ (no HIP): also built under ASan/UBSan and fuzzed on the CPU.
#include "onnx_model.h"

#include <cmath>
#include <cstring>

#include "fe_check.h"

namespace fe {
namespace onnx {
namespace {

struct Rd {
  const uint8_t* p;
  const uint8_t* e;
  bool eof() const { return p >= e; }
  uint64_t varint() {
    uint64_t v = 0;
    int sh = 0;
    for (;;) {
      FE_CHECK(p < e && sh < 64, "onnx: truncated varint");
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) break;
      sh += 7;
    }
    return v;
  }
  uint32_t fixed32() { FE_CHECK(e - p >= 4, "onnx: truncated fixed32"); uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
  uint64_t fixed64() { FE_CHECK(e - p >= 8, "onnx: truncated fixed64"); uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
  Rd sub() {
    const uint64_t n = varint();
    FE_CHECK(n <= (uint64_t)(e - p), "onnx: truncated length-delimited field");
    Rd r{p, p + n};
    p += n;
    return r;
  }
  std::string str() { Rd r = sub(); return std::string((const char*)r.p, (size_t)(r.e - r.p)); }
  void skip(int wt) {
    if (wt == 0) (void)varint();
    else if (wt == 1) (void)fixed64();
    else if (wt == 2) (void)sub();
    else if (wt == 5) (void)fixed32();
    else FE_CHECK(false, "onnx: unsupported wire type %d", wt);
  }
};

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400));
      bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ff) << 13;
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
  else bits = sign | (exp + 112) << 23 | man << 13;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

void parse_tensor(Rd r, TensorData& t) {
  std::vector<float> fdata;
  std::vector<int64_t> idata;
  std::vector<double> ddata;
  const uint8_t* raw = nullptr;
  size_t rawlen = 0;
  while (!r.eof()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    switch (field) {
      case 1:
        if (wt == 2) { Rd s = r.sub(); while (!s.eof()) t.dims.push_back((int64_t)s.varint()); }
        else t.dims.push_back((int64_t)r.varint());
        break;
      case 2: t.dtype = (int)r.varint(); break;
      case 4:
        if (wt == 2) { Rd s = r.sub(); while (!s.eof()) { uint32_t b = s.fixed32(); float f; memcpy(&f, &b, 4); fdata.push_back(f); } }
        else { uint32_t b = r.fixed32(); float f; memcpy(&f, &b, 4); fdata.push_back(f); }
        break;
      case 5: case 7:
        if (wt == 2) { Rd s = r.sub(); while (!s.eof()) idata.push_back((int64_t)s.varint()); }
        else idata.push_back((int64_t)r.varint());
        break;
      case 8: t.name = r.str(); break;
      case 9: { Rd s = r.sub(); raw = s.p; rawlen = (size_t)(s.e - s.p); break; }
      case 10:
        if (wt == 2) { Rd s = r.sub(); while (!s.eof()) { uint64_t b = s.fixed64(); double d; memcpy(&d, &b, 8); ddata.push_back(d); } }
        else { uint64_t b = r.fixed64(); double d; memcpy(&d, &b, 8); ddata.push_back(d); }
        break;
      case 13: FE_CHECK(false, "onnx: tensor '%s' uses external data, which is not supported", t.name.c_str());
      default: r.skip(wt);
    }
  }
  for (auto d : t.dims) FE_CHECK(d >= 0, "onnx: tensor '%s' has a negative dim", t.name.c_str());
  size_t n = 1;
  for (auto d : t.dims) {   // corrupted dims must not turn into a giant allocation
    FE_CHECK(d == 0 || n <= ((size_t)1 << 40) / (size_t)d, "onnx: tensor '%s' has an implausible shape", t.name.c_str());
    n *= (size_t)d;
  }
  FE_CHECK(n <= (size_t)1 << 32, "onnx: tensor '%s' has %zu elements", t.name.c_str(), n);
  auto need_raw = [&](size_t esz) { FE_CHECK(rawlen == n * esz, "onnx: tensor '%s' raw_data is %zu bytes, expected %zu", t.name.c_str(), rawlen, n * esz); };
  switch (t.dtype) {
    case 1:
      if (raw) { need_raw(4); t.f.resize(n); if (n) memcpy(t.f.data(), raw, n * 4); }
      else t.f = std::move(fdata);
      break;
    case 10:
      if (raw) need_raw(2);
      else FE_CHECK(idata.size() == n, "onnx: f16 tensor '%s' size", t.name.c_str());
      t.f.resize(n);
      if (raw) { for (size_t k = 0; k < n; ++k) { uint16_t h; memcpy(&h, raw + 2 * k, 2); t.f[k] = half_to_float(h); } }
      else { FE_CHECK(idata.size() == n, "onnx: f16 tensor '%s' size", t.name.c_str()); for (size_t k = 0; k < n; ++k) t.f[k] = half_to_float((uint16_t)idata[k]); }
      break;
    case 11:
      if (raw) need_raw(8);
      else FE_CHECK(ddata.size() == n, "onnx: f64 tensor '%s' size", t.name.c_str());
      t.f.resize(n);
      if (raw) { for (size_t k = 0; k < n; ++k) { double d; memcpy(&d, raw + 8 * k, 8); t.f[k] = (float)d; } }
      else { FE_CHECK(ddata.size() == n, "onnx: f64 tensor '%s' size", t.name.c_str()); for (size_t k = 0; k < n; ++k) t.f[k] = (float)ddata[k]; }
      break;
    case 7:
      if (raw) { need_raw(8); t.i.resize(n); if (n) memcpy(t.i.data(), raw, n * 8); }
      else t.i = std::move(idata);
      break;
    case 6:
      if (raw) { need_raw(4); t.i.resize(n); for (size_t k = 0; k < n; ++k) { int32_t v; memcpy(&v, raw + 4 * k, 4); t.i[k] = v; } }
      else t.i = std::move(idata);
      break;
    case 2: case 3: case 9:
      if (raw) { need_raw(1); t.i.resize(n); for (size_t k = 0; k < n; ++k) t.i[k] = t.dtype == 3 ? (int64_t)(int8_t)raw[k] : (int64_t)raw[k]; }
      else t.i = std::move(idata);
      break;
    default: FE_CHECK(false, "onnx: tensor '%s' has unsupported data type %d", t.name.c_str(), t.dtype);
  }
  FE_CHECK((t.is_int() ? t.i.size() : t.f.size()) == n, "onnx: tensor '%s' holds %zu values, dims say %zu", t.name.c_str(),
           t.is_int() ? t.i.size() : t.f.size(), n);
}

void parse_attr(Rd r, std::string& name, Attr& a) {
  while (!r.eof()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    switch (field) {
      case 1: name = r.str(); break;
      case 2: { uint32_t b = r.fixed32(); memcpy(&a.f, &b, 4); break; }
      case 3: a.i = (int64_t)r.varint(); break;
      case 4: a.s = r.str(); break;
      case 5: parse_tensor(r.sub(), a.t); break;
      case 7:
        if (wt == 2) { Rd s = r.sub(); while (!s.eof()) { uint32_t b = s.fixed32(); float f; memcpy(&f, &b, 4); a.floats.push_back(f); } }
        else { uint32_t b = r.fixed32(); float f; memcpy(&f, &b, 4); a.floats.push_back(f); }
        break;
      case 8:
        if (wt == 2) { Rd s = r.sub(); while (!s.eof()) a.ints.push_back((int64_t)s.varint()); }
        else a.ints.push_back((int64_t)r.varint());
        break;
      case 20: a.type = (int)r.varint(); break;
      default: r.skip(wt);
    }
  }
}

void parse_node(Rd r, Node& n) {
  while (!r.eof()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    switch (field) {
      case 1: n.in.push_back(r.str()); break;
      case 2: n.out.push_back(r.str()); break;
      case 3: n.name = r.str(); break;
      case 4: n.op = r.str(); break;
      case 5: { std::string nm; Attr a; parse_attr(r.sub(), nm, a); n.attr[nm] = std::move(a); break; }
      default: r.skip(wt);
    }
  }
}

void parse_value_info(Rd r, ValueInfo& v) {
  while (!r.eof()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == 1) v.name = r.str();
    else if (field == 2) {
      Rd ty = r.sub();
      while (!ty.eof()) {
        const uint64_t k2 = ty.varint();
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) {   // tensor_type
          Rd tt = ty.sub();
          while (!tt.eof()) {
            const uint64_t k3 = tt.varint();
            if ((k3 >> 3) == 2 && (k3 & 7) == 2) {   // shape
              Rd sh = tt.sub();
              while (!sh.eof()) {
                const uint64_t k4 = sh.varint();
                if ((k4 >> 3) == 1 && (k4 & 7) == 2) {   // dim
                  Rd dm = sh.sub();
                  int64_t val = -1;
                  while (!dm.eof()) {
                    const uint64_t k5 = dm.varint();
                    if ((k5 >> 3) == 1 && (k5 & 7) == 0) val = (int64_t)dm.varint();
                    else dm.skip((int)(k5 & 7));
                  }
                  v.dims.push_back(val);
                } else sh.skip((int)(k4 & 7));
              }
            } else tt.skip((int)(k3 & 7));
          }
        } else ty.skip((int)(k2 & 7));
      }
    } else r.skip(wt);
  }
}

void parse_graph(Rd r, Model& m) {
  std::vector<ValueInfo> ins;
  while (!r.eof()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    switch (field) {
      case 1: { Node n; parse_node(r.sub(), n); m.nodes.push_back(std::move(n)); break; }
      case 5: { TensorData t; parse_tensor(r.sub(), t); std::string nm = t.name; m.init[nm] = std::move(t); break; }
      case 11: { ValueInfo v; parse_value_info(r.sub(), v); ins.push_back(std::move(v)); break; }
      case 12: { ValueInfo v; parse_value_info(r.sub(), v); m.outputs.push_back(std::move(v)); break; }
      default: r.skip(wt);
    }
  }
  for (auto& v : ins)
    if (!m.init.count(v.name)) m.inputs.push_back(v);   // IR < 4 lists initializers among the inputs too
}

}  // namespace

void parse_model(const uint8_t* data, size_t len, Model& m) {
  FE_CHECK(data && len > 0, "onnx: empty model");
  Rd r{data, data + len};
  bool have_graph = false;
  while (!r.eof()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == 7 && wt == 2) { parse_graph(r.sub(), m); have_graph = true; }
    else if (field == 2 && wt == 2) m.producer = r.str();
    else if (field == 8 && wt == 2) {
      Rd o = r.sub();
      std::string dom;
      int64_t ver = 0;
      while (!o.eof()) {
        const uint64_t k = o.varint();
        if ((k >> 3) == 1) dom = o.str();
        else if ((k >> 3) == 2) ver = (int64_t)o.varint();
        else o.skip((int)(k & 7));
      }
      if (dom.empty() || dom == "ai.onnx") m.opset = ver;
    } else r.skip(wt);
  }
  FE_CHECK(have_graph, "onnx: no graph in model");
  FE_CHECK(!m.inputs.empty() && !m.outputs.empty(), "onnx: graph has no inputs or no outputs");
}

}  // namespace onnx
}  // namespace fe

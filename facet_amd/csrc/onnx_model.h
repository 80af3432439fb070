// In-memory form of an onnx.ModelProto (the subset the graph runtime reads) and its dependency-free protobuf wire reader
// (onnx_parse.cpp). No HIP dependency: the parser is also built with -fsanitize=address,undefined and fuzzed on the CPU
// (tests/test_onnx_host.py).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace fe {
namespace onnx {

struct TensorData {
  std::string name;
  std::vector<int64_t> dims;
  int dtype = 0;             // TensorProto.DataType: 1 f32, 2 u8, 3 i8, 6 i32, 7 i64, 9 bool, 10 f16, 11 f64
  std::vector<float> f;      // floating payloads, widened/narrowed to fp32
  std::vector<int64_t> i;    // integer payloads
  bool is_int() const { return dtype == 2 || dtype == 3 || dtype == 6 || dtype == 7 || dtype == 9; }
  size_t numel() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
  double at(size_t k) const { return is_int() ? (double)i[k] : (double)f[k]; }
};

struct Attr {
  int type = 0;
  float f = 0.f;
  int64_t i = 0;
  std::string s;
  std::vector<float> floats;
  std::vector<int64_t> ints;
  TensorData t;
};

struct Node {
  std::string op, name;
  std::vector<std::string> in, out;
  std::map<std::string, Attr> attr;
  int64_t geti(const char* k, int64_t def) const { auto it = attr.find(k); return it == attr.end() ? def : it->second.i; }
  float getf(const char* k, float def) const { auto it = attr.find(k); return it == attr.end() ? def : it->second.f; }
  std::string gets(const char* k, const char* def) const { auto it = attr.find(k); return it == attr.end() ? def : it->second.s; }
  std::vector<int64_t> getints(const char* k) const { auto it = attr.find(k); return it == attr.end() ? std::vector<int64_t>() : it->second.ints; }
  bool has(const char* k) const { return attr.count(k) != 0; }
};

struct ValueInfo { std::string name; std::vector<int64_t> dims; };   // dynamic dims are -1

struct Model {
  std::vector<Node> nodes;
  std::map<std::string, TensorData> init;
  std::vector<ValueInfo> inputs, outputs;   // inputs exclude initializers
  int64_t opset = 0;
  std::string producer;
};

void parse_model(const uint8_t* data, size_t len, Model& m);

}  // namespace onnx
}  // namespace fe

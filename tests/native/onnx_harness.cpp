// CPU harness for facet_amd/csrc/onnx_parse.cpp (built with -fsanitize=address,undefined by tests/test_onnx_host.py): parses every
// file named on the command line and prints one line per file: "ok <nodes> <initializers> <inputs> <outputs>" or "error <message>".
#include <cstdio>
#include <exception>
#include <vector>

#include "onnx_model.h"

int main(int argc, char** argv) {
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) return 3;
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    try {
      fe::onnx::Model m;
      fe::onnx::parse_model(buf.data(), buf.size(), m);
      size_t elems = 0;
      for (auto& kv : m.init) elems += kv.second.f.size() + kv.second.i.size();
      printf("ok %zu %zu %zu %zu %zu\n", m.nodes.size(), m.init.size(), m.inputs.size(), m.outputs.size(), elems);
    } catch (const std::exception& e) {
      printf("error %s\n", e.what());
    }
  }
  return 0;
}

// fe::Error and FE_CHECK - no HIP dependency, so host-only translation units (onnx_parse.cpp, lines_host.cpp) can be built with a
// plain C++ compiler and run under sanitizers on the CPU.
#pragma once
#include <cstdio>
#include <stdexcept>

namespace fe {
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
}  // namespace fe

#define FE_CHECK(cond, ...)                                                               \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      char _b[512];                                                                       \
      int _n = snprintf(_b, sizeof _b, "%s:%d: check failed (%s): ", __FILE__, __LINE__,  \
                        #cond);                                                           \
      snprintf(_b + _n, sizeof _b - _n, __VA_ARGS__);                                     \
      throw fe::Error(_b);                                                                \
    }                                                                                     \
  } while (0)

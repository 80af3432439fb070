// 2-byte implicit-GEMM convolution: the eight-wave 256x256 tile of the long-K GEMMs (see conv_bf16_kernel.h).
// FE_E = element type of this translation unit: bf16 here, f16 through kernels_conv_f16_wide4.hip, which includes this file.
#include "conv_bf16_kernel.h"
#ifndef FE_E
#define FE_E bf16
#endif

namespace fe {

template void launch_bf16_variant<FE_E, 2, 4, 4, 2, 1, 0, true>(const ConvParamsT<FE_E>&, hipStream_t);

}  // namespace fe

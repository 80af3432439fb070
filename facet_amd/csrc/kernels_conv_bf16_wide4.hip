// bf16 implicit-GEMM convolution: the eight-wave 256x256 tile of the long-K GEMMs (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace fe {

template void launch_bf16_variant<2, 4, 4, 2, 1, 0, true>(const ConvParamsH&, hipStream_t);

}  // namespace fe

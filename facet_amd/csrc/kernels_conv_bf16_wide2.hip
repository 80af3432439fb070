// bf16 implicit-GEMM convolution: wide-tile instantiations (a share of them per translation unit: see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace fe {

template void launch_bf16_variant<2, 2, 2, 2, 1, 0, false>(const ConvParamsH&, hipStream_t);
template void launch_bf16_variant<2, 2, 4, 2, 1, 0, false>(const ConvParamsH&, hipStream_t);

}  // namespace fe

// 2-byte implicit-GEMM convolution: the wide tiles in their fp32-stream form (p.res32 / p.y32; see conv_bf16_kernel.h).
// FE_E = element type of this translation unit: bf16 here, f16 through kernels_conv_f16_s32.hip, which includes this file.
#include "conv_bf16_kernel.h"
#ifndef FE_E
#define FE_E bf16
#endif

namespace fe {

FE_WIDE_TILES_S32(, FE_E)

}  // namespace fe

// fp16 instantiations of kernels_conv_bf16_s32.hip (same source, element type f16).
#define FE_E f16
#include "kernels_conv_bf16_s32.hip"

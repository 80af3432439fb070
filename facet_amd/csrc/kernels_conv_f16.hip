// fp16 instantiation of the 2-byte convolution launcher and its narrow tiles (same source as kernels_conv_bf16.hip, element type f16).
#define FE_E f16
#include "kernels_conv_bf16.hip"

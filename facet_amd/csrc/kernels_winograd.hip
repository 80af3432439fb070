// Winograd F(2x2, 3x3) transforms for the deep 3x3 / stride-1 convolutions (ResNet-50 layer3 / layer4 conv2: Cin = Cout = 256 / 512).
//   Y = A^T [ (G g G^T) . (B^T d B) ] A      - 16 multiplies per 2x2 outputs and channel pair instead of 36 (2.25x fewer MACs)
// The element-wise product summed over channels is 16 independent GEMMs [tiles x Cin] x [Cin x Cout], which run as ONE batched
// launch of the LDS-DMA GEMM kernel (gridDim.y = 16). This file holds the two HBM-bound transforms around it:
//   wino_input : NHWC activation -> V[16][tiles][Cin]      (B^T d B, additions only; zero padding applied while loading)
//   wino_output: M[16][tiles][Cout] -> NHWC output          (A^T m A, per-channel scale / shift, optional ReLU)
// Worth it only where the GEMM's K (= Cin) is long enough to run near the kernel's plateau and the transforms' extra HBM round
// trip (~5 B moved per original activation byte) is small next to it: Cin >= 256 here (measured).
#include "fe_common.h"

namespace fe {

static inline int grid_for_w(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// one thread: one tile x 4 channels
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, int N, int H, int W, int C4, int th, int tw,
                                                         float* __restrict__ V, size_t plane) {
  const size_t total = (size_t)N * th * tw * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t t = i / C4;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int img = (int)(t / ((size_t)tw * th));
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iy = 2 * ty - 1 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ix = 2 * tx - 1 + q;
        d[r][q] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                      ? *reinterpret_cast<const float4*>(x + (((size_t)img * H + iy) * W + ix) * ldx + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 tmid[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // B^T d
      tmid[0][q] = f4sub(d[0][q], d[2][q]);
      tmid[1][q] = f4add(d[1][q], d[2][q]);
      tmid[2][q] = f4sub(d[2][q], d[1][q]);
      tmid[3][q] = f4sub(d[1][q], d[3][q]);
    }
    float* dst = V + t * (size_t)(C4 * 4) + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // (B^T d) B
      *reinterpret_cast<float4*>(dst + (size_t)(4 * r + 0) * plane) = f4sub(tmid[r][0], tmid[r][2]);
      *reinterpret_cast<float4*>(dst + (size_t)(4 * r + 1) * plane) = f4add(tmid[r][1], tmid[r][2]);
      *reinterpret_cast<float4*>(dst + (size_t)(4 * r + 2) * plane) = f4sub(tmid[r][2], tmid[r][1]);
      *reinterpret_cast<float4*>(dst + (size_t)(4 * r + 3) * plane) = f4sub(tmid[r][1], tmid[r][3]);
    }
  }
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, size_t plane, int N, int H, int W, int C4, int th, int tw,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                          float* __restrict__ y, int ldy) {
  const size_t total = (size_t)N * th * tw * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t t = i / C4;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int img = (int)(t / ((size_t)tw * th));
    const float* src = Mb + t * (size_t)(C4 * 4) + c;
    float4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) m[r][q] = *reinterpret_cast<const float4*>(src + (size_t)(4 * r + q) * plane);
    float4 s[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // A^T m
      s[0][q] = f4add(f4add(m[0][q], m[1][q]), m[2][q]);
      s[1][q] = f4sub(f4sub(m[1][q], m[2][q]), m[3][q]);
    }
    const float4 sf = shift ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = 2 * ty + r;
      if (oy >= H) continue;
      float4 o[2] = {f4add(f4add(s[r][0], s[r][1]), s[r][2]), f4sub(f4sub(s[r][1], s[r][2]), s[r][3])};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ox = 2 * tx + q;
        if (ox >= W) continue;
        float4 v = make_float4(o[q].x * sc.x + sf.x, o[q].y * sc.y + sf.y, o[q].z * sc.z + sf.z, o[q].w * sc.w + sf.w);
        if (relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        *reinterpret_cast<float4*>(y + (((size_t)img * H + oy) * W + ox) * ldy + c) = v;
      }
    }
  }
}

void launch_wino_input(const Tensor& x, int th, int tw, float* V, hipStream_t s) {
  FE_CHECK(x.c % 4 == 0 && x.ld % 4 == 0, "wino_input: channels");
  const size_t tiles = (size_t)x.n * th * tw;
  hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for_w(tiles * (x.c / 4))), dim3(256), 0, s, x.p, x.ld, x.n, x.h, x.w, x.c / 4, th, tw, V, tiles * x.c);
  FE_HIP(hipGetLastError());
}
void launch_wino_output(const float* Mb, const Tensor& y, int th, int tw, const float* scale, const float* shift, int relu, hipStream_t s) {
  FE_CHECK(y.c % 4 == 0 && y.ld % 4 == 0, "wino_output: channels");
  const size_t tiles = (size_t)y.n * th * tw;
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for_w(tiles * (y.c / 4))), dim3(256), 0, s, Mb, tiles * y.c, y.n, y.h, y.w, y.c / 4, th, tw, scale, shift, relu, y.p,
                     y.ld);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

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

// ---- F(4x4, 3x3): 36 multiplies per 4x4 outputs (4x fewer MACs than direct), 6x6 input tiles with stride 4 ------------------------
//   B^T rows: [4 0 -5 0 1 0] [0 -4 -4 1 1 0] [0 4 -4 -1 1 0] [0 -2 -1 2 1 0] [0 2 -1 -2 1 0] [0 4 0 -5 0 1]
//   A^T rows: [1 1 1 1 1 0] [0 1 -1 2 -2 0] [0 1 1 4 4 0] [0 1 -1 8 -8 1]            (Lavin & Gray 2015)
// One thread = one tile x ONE channel (36 live values); consecutive lanes take consecutive channels, so every access of a
// wavefront is a contiguous 256-B run.
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5, float* o) {
  o[0] = 4.f * d0 - 5.f * d2 + d4;
  o[1] = -4.f * d1 - 4.f * d2 + d3 + d4;
  o[2] = 4.f * d1 - 4.f * d2 - d3 + d4;
  o[3] = -2.f * d1 - d2 + 2.f * d3 + d4;
  o[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
  o[5] = 4.f * d1 - 5.f * d3 + d5;
}
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, int ldx, int N, int H, int W, int C, int th, int tw,
                                                          float* __restrict__ V, size_t plane) {
  const size_t total = (size_t)N * th * tw * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int img = (int)(t / ((size_t)tw * th));
    float tm[6][6];   // B^T d, built column by column
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int ix = 4 * tx - 1 + q;
      float col[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int iy = 4 * ty - 1 + r;
        col[r] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? x[(((size_t)img * H + iy) * W + ix) * ldx + c] : 0.f;
      }
      float o[6];
      bt6(col[0], col[1], col[2], col[3], col[4], col[5], o);
#pragma unroll
      for (int r = 0; r < 6; ++r) tm[r][q] = o[r];
    }
    float* dst = V + t * (size_t)C + c;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float o[6];
      bt6(tm[r][0], tm[r][1], tm[r][2], tm[r][3], tm[r][4], tm[r][5], o);
#pragma unroll
      for (int q = 0; q < 6; ++q) dst[(size_t)(6 * r + q) * plane] = o[q];
    }
  }
}
__device__ __forceinline__ void at4(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float* o) {
  o[0] = m0 + m1 + m2 + m3 + m4;
  o[1] = m1 - m2 + 2.f * m3 - 2.f * m4;
  o[2] = m1 + m2 + 4.f * m3 + 4.f * m4;
  o[3] = m1 - m2 + 8.f * m3 - 8.f * m4 + m5;
}
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Mb, size_t plane, int N, int H, int W, int C, int th, int tw,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                           const float* __restrict__ slope, const float* __restrict__ res, int ldr,
                                                           int res_after_act, float* __restrict__ y, int ldy) {
  const size_t total = (size_t)N * th * tw * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int img = (int)(t / ((size_t)tw * th));
    const float* src = Mb + t * (size_t)C + c;
    float s[4][6];   // A^T m, column by column
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float o[4];
      at4(src[(size_t)(0 * 6 + q) * plane], src[(size_t)(1 * 6 + q) * plane], src[(size_t)(2 * 6 + q) * plane], src[(size_t)(3 * 6 + q) * plane],
          src[(size_t)(4 * 6 + q) * plane], src[(size_t)(5 * 6 + q) * plane], o);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r][q] = o[r];
    }
    const float sc = scale ? scale[c] : 1.f, sf = shift ? shift[c] : 0.f, sl = (act == ACT_PRELU) ? slope[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oy = 4 * ty + r;
      if (oy >= H) continue;
      float o[4];
      at4(s[r][0], s[r][1], s[r][2], s[r][3], s[r][4], s[r][5], o);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ox = 4 * tx + q;
        if (ox >= W) continue;
        const size_t pix = ((size_t)img * H + oy) * W + ox;
        float v = o[q] * sc + sf;
        const float rr = res ? res[pix * ldr + c] : 0.f;
        if (!res_after_act) v += rr;
        if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == ACT_PRELU) v = v > 0.f ? v : v * sl;
        if (res_after_act) v += rr;
        y[pix * ldy + c] = v;
      }
    }
  }
}
void launch_wino4_input(const Tensor& x, int th, int tw, float* V, hipStream_t s) {
  const size_t tiles = (size_t)x.n * th * tw;
  hipLaunchKernelGGL(wino4_input_kernel, dim3(grid_for_w(tiles * x.c)), dim3(256), 0, s, x.p, x.ld, x.n, x.h, x.w, x.c, th, tw, V, tiles * x.c);
  FE_HIP(hipGetLastError());
}
void launch_wino4_output(const float* Mb, const Tensor& y, int th, int tw, const float* scale, const float* shift, int act, const float* slope,
                         const Tensor* res, int res_after_act, hipStream_t s) {
  FE_CHECK(act == ACT_NONE || act == ACT_RELU || (act == ACT_PRELU && slope), "wino4_output: activation");
  FE_CHECK(!res || (res->c == y.c && res->pixels() == y.pixels()), "wino4_output: residual shape");
  const size_t tiles = (size_t)y.n * th * tw;
  hipLaunchKernelGGL(wino4_output_kernel, dim3(grid_for_w(tiles * y.c)), dim3(256), 0, s, Mb, tiles * y.c, y.n, y.h, y.w, y.c, th, tw, scale, shift, act,
                     slope, res ? res->p : nullptr, res ? res->ld : 0, res_after_act, y.p, y.ld);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

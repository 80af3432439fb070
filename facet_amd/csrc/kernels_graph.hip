// HBM-bound kernels used by the ONNX-subset graph runtime (onnx_graph.hip): per-channel affine + activation,
// binary elementwise, strided gather (transpose / compaction / channel-slice copies), nearest upsampling,
// average pooling and depthwise convolution. NHWC fp32 views with channel counts that are multiples of 4, so every
// thread moves 16-byte vectors; grids are capped and grid-strided like kernels_misc.hip.
#include "fe_common.h"

namespace fe {

static inline int grid_for_g(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float act_one(float v, int act, float sl) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_PRELU) return v > 0.f ? v : v * sl;
  if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

// ---- y = act(x * scale[c] + shift[c]) -------------------------------------------------------------------------
__global__ void affine_act_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, size_t pixels, int c4,
                                  const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                  const float* __restrict__ slope) {
  const size_t total = pixels * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / c4;
    const int ch = (int)(i - pix * c4) * 4;
    float4 v = *reinterpret_cast<const float4*>(x + pix * ldx + ch);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f), sl = sf;
    if (scale) sc = *reinterpret_cast<const float4*>(scale + ch);
    if (shift) sf = *reinterpret_cast<const float4*>(shift + ch);
    if (slope) sl = *reinterpret_cast<const float4*>(slope + ch);
    v.x = act_one(v.x * sc.x + sf.x, act, sl.x);
    v.y = act_one(v.y * sc.y + sf.y, act, sl.y);
    v.z = act_one(v.z * sc.z + sf.z, act, sl.z);
    v.w = act_one(v.w * sc.w + sf.w, act, sl.w);
    *reinterpret_cast<float4*>(y + pix * ldy + ch) = v;
  }
}
void launch_affine_act(const Tensor& x, const Tensor& y, const float* scale, const float* shift, int act, const float* slope,
                       hipStream_t s) {
  FE_CHECK(x.c == y.c && x.pixels() == y.pixels() && x.c % 4 == 0 && x.ld % 4 == 0 && y.ld % 4 == 0, "affine_act: shapes");
  FE_CHECK(act != ACT_PRELU || slope, "affine_act: PReLU without slopes");
  const size_t work = x.pixels() * (x.c / 4);
  if (!work) return;
  hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for_g(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.pixels(), x.c / 4,
                     scale, shift, act, slope);
  FE_HIP(hipGetLastError());
}

// ---- y = act(a (op) b), same shape ------------------------------------------------------------------------------
__global__ void binary_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ y,
                              int ldy, size_t pixels, int c4, int op, int act) {
  const size_t total = pixels * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / c4;
    const int ch = (int)(i - pix * c4) * 4;
    const float4 u = *reinterpret_cast<const float4*>(a + pix * lda + ch);
    const float4 w = *reinterpret_cast<const float4*>(b + pix * ldb + ch);
    float4 v;
    if (op == 0) v = make_float4(u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w);
    else if (op == 1) v = make_float4(u.x - w.x, u.y - w.y, u.z - w.z, u.w - w.w);
    else if (op == 2) v = make_float4(u.x * w.x, u.y * w.y, u.z * w.z, u.w * w.w);
    else v = make_float4(u.x / w.x, u.y / w.y, u.z / w.z, u.w / w.w);
    v.x = act_one(v.x, act, 0.f); v.y = act_one(v.y, act, 0.f); v.z = act_one(v.z, act, 0.f); v.w = act_one(v.w, act, 0.f);
    *reinterpret_cast<float4*>(y + pix * ldy + ch) = v;
  }
}
void launch_binary(const Tensor& a, const Tensor& b, const Tensor& y, int op, int act, hipStream_t s) {
  FE_CHECK(a.c == b.c && a.c == y.c && a.pixels() == b.pixels() && a.pixels() == y.pixels() && a.c % 4 == 0 &&
               a.ld % 4 == 0 && b.ld % 4 == 0 && y.ld % 4 == 0, "binary: shapes");
  FE_CHECK(act != ACT_PRELU, "binary: PReLU is not fusable here");
  const size_t work = a.pixels() * (a.c / 4);
  if (!work) return;
  hipLaunchKernelGGL(binary_kernel, dim3(grid_for_g(work)), dim3(256), 0, s, a.p, a.ld, b.p, b.ld, y.p, y.ld, a.pixels(),
                     a.c / 4, op, act);
  FE_HIP(hipGetLastError());
}

// ---- dst (dense, up to 6 dims) gathered from src through element strides -------------------------------------------
struct GatherDims { long long d[6]; long long s[6]; };
__global__ void gather_strided_kernel(const float* __restrict__ src, float* __restrict__ dst, GatherDims g, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    long long off = 0;
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      const long long q = (long long)(r % (size_t)g.d[k]);
      r /= (size_t)g.d[k];
      off += q * g.s[k];
    }
    dst[i] = src[off];
  }
}
void launch_gather_strided(const float* src, float* dst, const long long dims[6], const long long sstr[6], hipStream_t s) {
  GatherDims g;
  size_t total = 1;
  for (int k = 0; k < 6; ++k) { g.d[k] = dims[k]; g.s[k] = sstr[k]; FE_CHECK(dims[k] >= 1, "gather: dim"); total *= (size_t)dims[k]; }
  hipLaunchKernelGGL(gather_strided_kernel, dim3(grid_for_g(total)), dim3(256), 0, s, src, dst, g, total);
  FE_HIP(hipGetLastError());
}

// ---- nearest-neighbour resize, ONNX 'asymmetric' coordinates + floor (what torch exports for nn.Upsample/F.interpolate) ----
__global__ void nearest_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int n, int h, int w, int ho,
                               int wo, int c4, float sy, float sx) {
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ox = (int)(t % wo); t /= wo;
    const int oy = (int)(t % ho);
    const int img = (int)(t / ho);
    int iy = (int)floorf((float)oy * sy), ix = (int)floorf((float)ox * sx);
    iy = iy < h - 1 ? iy : h - 1;
    ix = ix < w - 1 ? ix : w - 1;
    *reinterpret_cast<float4*>(y + (((size_t)img * ho + oy) * wo + ox) * ldy + ch) =
        *reinterpret_cast<const float4*>(x + (((size_t)img * h + iy) * w + ix) * ldx + ch);
  }
}
void launch_nearest(const Tensor& x, const Tensor& y, hipStream_t s) {
  FE_CHECK(x.c == y.c && x.n == y.n && x.c % 4 == 0 && x.ld % 4 == 0 && y.ld % 4 == 0, "nearest: shapes");
  const size_t work = y.pixels() * (y.c / 4);
  if (!work) return;
  hipLaunchKernelGGL(nearest_kernel, dim3(grid_for_g(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, y.h, y.w,
                     x.c / 4, (float)x.h / (float)y.h, (float)x.w / (float)y.w);
  FE_HIP(hipGetLastError());
}

// ---- average pooling (window clipped to the image; divisor = k*k when count_include_pad and the window only overlaps
//      explicit padding, else the number of in-image taps - ONNX AveragePool) ------------------------------------------
__global__ void avgpool_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int n, int h, int w, int ho,
                               int wo, int c4, int k, int stride, int pad, int include_pad) {
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ox = (int)(t % wo); t /= wo;
    const int oy = (int)(t % ho);
    const int img = (int)(t / ho);
    const int y0 = oy * stride - pad, x0 = ox * stride - pad;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0, cnt_pad = 0;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const int iy = y0 + dy, ix = x0 + dx;
        if (iy < -pad || iy >= h + pad || ix < -pad || ix >= w + pad) continue;   // beyond the padded image (ceil_mode)
        ++cnt_pad;
        if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)img * h + iy) * w + ix) * ldx + ch);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        ++cnt;
      }
    const float d = (float)(include_pad ? cnt_pad : cnt);
    const float inv = d > 0.f ? 1.f / d : 0.f;
    *reinterpret_cast<float4*>(y + (((size_t)img * ho + oy) * wo + ox) * ldy + ch) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}
void launch_avgpool(const Tensor& x, const Tensor& y, int k, int stride, int pad, int include_pad, hipStream_t s) {
  FE_CHECK(x.c == y.c && x.n == y.n && x.c % 4 == 0 && x.ld % 4 == 0 && y.ld % 4 == 0, "avgpool: shapes");
  const size_t work = y.pixels() * (y.c / 4);
  if (!work) return;
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for_g(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, y.h, y.w,
                     x.c / 4, k, stride, pad, include_pad);
  FE_HIP(hipGetLastError());
}

// ---- depthwise convolution: y[n,oy,ox,c] = act((sum_t x[n,iy,ix,c] * w[t][c]) * scale[c] + shift[c] (+res)) ---------
__global__ void dwconv_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, const float* __restrict__ wt,
                              int n, int h, int w, int ho, int wo, int c4, int kh, int kw, int sh, int sw, int ph, int pw,
                              const float* __restrict__ scale, const float* __restrict__ shift, int act,
                              const float* __restrict__ slope, const float* __restrict__ res, int ldr) {
  const size_t total = (size_t)n * ho * wo * c4;
  const int C = c4 * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ox = (int)(t % wo); t /= wo;
    const int oy = (int)(t % ho);
    const int img = (int)(t / ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < kh; ++dy) {
      const int iy = oy * sh - ph + dy;
      if (iy < 0 || iy >= h) continue;
      for (int dx = 0; dx < kw; ++dx) {
        const int ix = ox * sw - pw + dx;
        if (ix < 0 || ix >= w) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)img * h + iy) * w + ix) * ldx + ch);
        const float4 k4 = *reinterpret_cast<const float4*>(wt + (size_t)(dy * kw + dx) * C + ch);
        acc.x += v.x * k4.x; acc.y += v.y * k4.y; acc.z += v.z * k4.z; acc.w += v.w * k4.w;
      }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f), sl = sf;
    if (scale) sc = *reinterpret_cast<const float4*>(scale + ch);
    if (shift) sf = *reinterpret_cast<const float4*>(shift + ch);
    if (slope) sl = *reinterpret_cast<const float4*>(slope + ch);
    const size_t opix = ((size_t)img * ho + oy) * wo + ox;
    float4 v = make_float4(acc.x * sc.x + sf.x, acc.y * sc.y + sf.y, acc.z * sc.z + sf.z, acc.w * sc.w + sf.w);
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + opix * ldr + ch);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    v.x = act_one(v.x, act, sl.x); v.y = act_one(v.y, act, sl.y); v.z = act_one(v.z, act, sl.z); v.w = act_one(v.w, act, sl.w);
    *reinterpret_cast<float4*>(y + opix * ldy + ch) = v;
  }
}
void launch_dwconv(const Tensor& x, const Tensor& y, const float* wt, int kh, int kw, int sh, int sw, int ph, int pw,
                   const float* scale, const float* shift, int act, const float* slope, const Tensor* res, hipStream_t s) {
  FE_CHECK(x.c == y.c && x.n == y.n && x.c % 4 == 0 && x.ld % 4 == 0 && y.ld % 4 == 0, "dwconv: shapes");
  FE_CHECK(!res || (res->c == y.c && res->pixels() == y.pixels() && res->ld % 4 == 0), "dwconv: residual shape");
  FE_CHECK(act != ACT_PRELU || slope, "dwconv: PReLU without slopes");
  const size_t work = y.pixels() * (y.c / 4);
  if (!work) return;
  hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for_g(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, wt, x.n, x.h, x.w, y.h, y.w,
                     x.c / 4, kh, kw, sh, sw, ph, pw, scale, shift, act, slope, res ? res->p : nullptr, res ? res->ld : 0);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

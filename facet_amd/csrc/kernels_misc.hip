// HBM-bound helper kernels of the scoring hot path: pixel preprocessing, pooling, resampling,
// LayerNorm / softmax rows. All operate on NHWC fp32 views (see fe::Tensor) with 16-byte vector
// accesses where the channel count allows; grid capped at 2048 blocks with grid-stride loops.
#include "fe_common.h"

namespace fe {

template <class T> static inline bool vec4_ok(const T* a, const T* b) { return (((uintptr_t)a | (uintptr_t)b) & (4 * sizeof(T) - 1)) == 0; }

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- u8 HWC (RGB or BGR) -> fp32 NHWC4, ((v/255) - mean)/std --------------------------------------
// Mirrors reference models/pyiqa_scorer.py:155-158 (array/255, HWC->CHW) followed by the model's own
// ImageNet normalisation; channel 3 is zero so the stem conv can use 16-B pixel loads.
__global__ void u8_to_nhwc4_kernel(const uint8_t* __restrict__ src, float4* __restrict__ dst, size_t pixels,
                                   float m0, float m1, float m2, float s0, float s1, float s2, int bgr) {
  // 4 pixels (12 bytes = 3 dwords) per thread
  const size_t groups = pixels / 4;
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
    const uint32_t a = s32[g * 3 + 0], b = s32[g * 3 + 1], c = s32[g * 3 + 2];
    uint8_t px[12];
    px[0] = a; px[1] = a >> 8; px[2] = a >> 16; px[3] = a >> 24;
    px[4] = b; px[5] = b >> 8; px[6] = b >> 16; px[7] = b >> 24;
    px[8] = c; px[9] = c >> 8; px[10] = c >> 16; px[11] = c >> 24;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float c0 = (float)px[k * 3 + 0] / 255.0f, c1 = (float)px[k * 3 + 1] / 255.0f, c2 = (float)px[k * 3 + 2] / 255.0f;
      if (bgr) { float tmp = c0; c0 = c2; c2 = tmp; }
      dst[g * 4 + k] = make_float4((c0 - m0) / s0, (c1 - m1) / s1, (c2 - m2) / s2, 0.f);
    }
  }
  // tail (pixels % 4)
  if (blockIdx.x == 0 && threadIdx.x < (pixels & 3)) {
    const size_t i = groups * 4 + threadIdx.x;
    float c0 = (float)src[i * 3 + 0] / 255.0f, c1 = (float)src[i * 3 + 1] / 255.0f, c2 = (float)src[i * 3 + 2] / 255.0f;
    if (bgr) { float tmp = c0; c0 = c2; c2 = tmp; }
    dst[i] = make_float4((c0 - m0) / s0, (c1 - m1) / s1, (c2 - m2) / s2, 0.f);
  }
}

void launch_u8_to_nhwc4_norm(const uint8_t* src, float* dst, size_t pixels, const float mean[3],
                             const float stdv[3], int bgr, hipStream_t s) {
  FE_CHECK(((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 15) == 0, "u8_to_nhwc4: alignment");
  hipLaunchKernelGGL(u8_to_nhwc4_kernel, dim3(grid_for(pixels / 4 + 1)), dim3(256), 0, s, src, (float4*)dst, pixels,
                     mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], bgr);
  FE_HIP(hipGetLastError());
}

// ---- layout shuffles at the C-ABI boundary (host tensors arrive NCHW like the reference's) ----------
template <class T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int c, int h, int w, int cpad) {
  const size_t total = (size_t)n * h * w * cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % cpad;
    const size_t pix = i / cpad;
    const size_t hw = (size_t)h * w;
    const size_t img = pix / hw, rem = pix - img * hw;
    stf(dst + i, ch < c ? src[(img * c + ch) * hw + rem] : 0.f);
  }
}
template <class T>
void launch_nchw_to_nhwc(const float* src, T* dst, int n, int c, int h, int w, int cpad, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for((size_t)n * h * w * cpad)), dim3(256), 0, s, src, dst, n, c, h, w, cpad);
  FE_HIP(hipGetLastError());
}
template <class T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int ld, float* __restrict__ dst, int n, int c, int h, int w) {
  const size_t total = (size_t)n * c * h * w;
  const size_t hw = (size_t)h * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t rem = i % hw;
    const size_t t = i / hw;
    const int ch = t % c;
    const size_t img = t / c;
    dst[i] = ldf(src + (img * hw + rem) * ld + ch);
  }
}
template <class T>
void launch_nhwc_to_nchw(const T* src, int ld, float* dst, int n, int c, int h, int w, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for((size_t)n * c * h * w)), dim3(256), 0, s, src, ld, dst, n, c, h, w);
  FE_HIP(hipGetLastError());
}

// ---- max pool (NHWC, implicit -inf padding; ceil_mode windows are clipped to the input) ------------
template <class T, int VEC>
__global__ void maxpool_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int h, int w,
                               int c, int ho, int wo, int k, int stride, int pad) {
  const int cv = c / VEC;
  const size_t total = (size_t)n * ho * wo * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = i % cv;
    size_t pix = i / cv;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    float best[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) best[v] = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
      const int ih = oh * stride - pad + dy;
      if ((unsigned)ih >= (unsigned)h) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int iw = ow * stride - pad + dx;
        if ((unsigned)iw >= (unsigned)w) continue;
        const T* p = x + ((img * h + ih) * w + iw) * ldx + cg * VEC;
        if (VEC == 4) {
          const float4 q = ld4(p);
          best[0] = fmaxf(best[0], q.x); best[1 % VEC] = fmaxf(best[1 % VEC], q.y);
          best[2 % VEC] = fmaxf(best[2 % VEC], q.z); best[3 % VEC] = fmaxf(best[3 % VEC], q.w);
        } else {
          best[0] = fmaxf(best[0], ldf(p));
        }
      }
    }
    T* o = y + ((img * ho + oh) * wo + ow) * ldy + cg * VEC;
    if (VEC == 4) st4(o, make_float4(best[0], best[1 % VEC], best[2 % VEC], best[3 % VEC]));
    else stf(o, best[0]);
  }
}
// 3x3 windows (every max pool of the hot path: ResNet stems, U2-Net-P) with 16 bytes per thread and tap, all nine loads issued before the
// first use (clamped addresses, a select instead of a branch per tap): the general kernel above walks its window behind two nested
// `continue`s with one 8-byte load in flight per thread and ran the 512^2 x 64 -> 256^2 pool of a TOPIQ micro-batch at 1.9 TB/s.
template <class T>
__global__ void maxpool3_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int h, int w, int c, int ho, int wo, int stride, int pad) {
  constexpr int VEC = 16 / sizeof(T);
  const int cv = c / VEC;
  const size_t total = (size_t)n * ho * wo * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = i % cv;
    size_t pix = i / cv;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    uint4 q[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh * stride - pad + t / 3, iw = ow * stride - pad + t % 3;
      ok[t] = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
      q[t] = *reinterpret_cast<const uint4*>(x + ((img * h + (ok[t] ? ih : 0)) * w + (ok[t] ? iw : 0)) * ldx + cg * VEC);
    }
    float best[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) best[v] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float f[VEC];
      if constexpr (sizeof(T) == 4) {
        f[0] = __uint_as_float(q[t].x); f[1] = __uint_as_float(q[t].y); f[2] = __uint_as_float(q[t].z); f[3] = __uint_as_float(q[t].w);
      } else {
        fe_unpack2((const T*)nullptr, q[t].x, f[0], f[1]); fe_unpack2((const T*)nullptr, q[t].y, f[2], f[3]);
        fe_unpack2((const T*)nullptr, q[t].z, f[4 % VEC], f[5 % VEC]); fe_unpack2((const T*)nullptr, q[t].w, f[6 % VEC], f[7 % VEC]);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) best[v] = ok[t] ? fmaxf(best[v], f[v]) : best[v];
    }
    T* o = y + ((img * ho + oh) * wo + ow) * ldy + cg * VEC;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(best[0], best[1], best[2], best[3]);
    } else {      // (a maximum of stored values is representable: no rounding, no saturation needed)
      *reinterpret_cast<uint4*>(o) = make_uint4(fe_pack2((const T*)nullptr, best[0], best[1]), fe_pack2((const T*)nullptr, best[2], best[3]),
                                                fe_pack2((const T*)nullptr, best[4 % VEC], best[5 % VEC]), fe_pack2((const T*)nullptr, best[6 % VEC], best[7 % VEC]));
    }
  }
}

template <class T>
void launch_maxpool(const TensorT<T>& x, const TensorT<T>& y, int k, int stride, int pad, hipStream_t s) {
  FE_CHECK(x.c == y.c && x.n == y.n, "maxpool: shape mismatch");
  constexpr int V16 = 16 / (int)sizeof(T);
  if (k == 3 && x.c % V16 == 0 && x.ld % V16 == 0 && y.ld % V16 == 0 && (((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0 && !getenv("FE_NO_MAXPOOL3")) {
    const size_t work = y.pixels() * (x.c / V16);
    hipLaunchKernelGGL((maxpool3_kernel<T>), dim3(grid_for(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c, y.h, y.w, stride, pad);
    FE_HIP(hipGetLastError());
    return;
  }
  const bool v4 = (x.c % 4 == 0) && (x.ld % 4 == 0) && (y.ld % 4 == 0) && vec4_ok(x.p, y.p);
  const size_t work = y.pixels() * (v4 ? x.c / 4 : x.c);
  if (v4)
    hipLaunchKernelGGL((maxpool_kernel<T, 4>), dim3(grid_for(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c, y.h, y.w, k, stride, pad);
  else
    hipLaunchKernelGGL((maxpool_kernel<T, 1>), dim3(grid_for(work)), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c, y.h, y.w, k, stride, pad);
  FE_HIP(hipGetLastError());
}

// ---- bilinear resize, align_corners=False (torch F.interpolate semantics, reference samp_net.py:59) --
template <class T>
__global__ void bilinear_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int h, int w,
                                int c, int ho, int wo, float sy, float sx) {
  const size_t total = (size_t)n * ho * wo * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % c;
    size_t pix = i / c;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    float fy = ((float)oh + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = ((float)ow + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* b = x + img * h * w * ldx + ch;
    const float v00 = ldf(b + ((size_t)y0 * w + x0) * ldx), v01 = ldf(b + ((size_t)y0 * w + x1) * ldx);
    const float v10 = ldf(b + ((size_t)y1 * w + x0) * ldx), v11 = ldf(b + ((size_t)y1 * w + x1) * ldx);
    stf(y + ((img * ho + oh) * wo + ow) * ldy + ch, hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
  }
}
// 4 channels per thread (8- / 16-byte accesses); same arithmetic per element as the scalar kernel
template <class T>
__global__ void bilinear_vec4_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int h, int w,
                                     int c4, int ho, int wo, float sy, float sx) {
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c4) * 4;
    size_t pix = i / c4;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    float fy = ((float)oh + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = ((float)ow + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* b = x + img * h * w * ldx + ch;
    const float4 v00 = ld4(b + ((size_t)y0 * w + x0) * ldx), v01 = ld4(b + ((size_t)y0 * w + x1) * ldx);
    const float4 v10 = ld4(b + ((size_t)y1 * w + x0) * ldx), v11 = ld4(b + ((size_t)y1 * w + x1) * ldx);
    st4(y + ((img * ho + oh) * wo + ow) * ldy + ch,
        make_float4(hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x), hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y),
                    hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z), hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w)));
  }
}
template <class T>
void launch_bilinear(const TensorT<T>& x, const TensorT<T>& y, hipStream_t s) {
  FE_CHECK(x.c == y.c && x.n == y.n, "bilinear: shape mismatch");
  const float sy = (float)x.h / (float)y.h, sx = (float)x.w / (float)y.w;
  if (x.c % 4 == 0 && x.ld % 4 == 0 && y.ld % 4 == 0 && vec4_ok(x.p, y.p)) {
    hipLaunchKernelGGL(bilinear_vec4_kernel<T>, dim3(grid_for(y.pixels() * (size_t)(y.c / 4))), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c / 4, y.h, y.w, sy, sx);
    FE_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(bilinear_kernel<T>, dim3(grid_for(y.numel())), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c, y.h, y.w, sy, sx);
  FE_HIP(hipGetLastError());
}

// ---- adaptive average pool (torch semantics) ---------------------------------------------------------
template <class T>
__global__ void adaptive_avgpool_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int h,
                                        int w, int c, int ho, int wo) {
  const size_t total = (size_t)n * ho * wo * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % c;
    size_t pix = i / c;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    const int hs = (oh * h) / ho, he = ((oh + 1) * h + ho - 1) / ho;
    const int ws = (ow * w) / wo, we = ((ow + 1) * w + wo - 1) / wo;
    float acc = 0.f;
    for (int iy = hs; iy < he; ++iy)
      for (int ix = ws; ix < we; ++ix) acc += ldf(x + ((img * h + iy) * w + ix) * ldx + ch);
    stf(y + ((img * ho + oh) * wo + ow) * ldy + ch, acc / (float)((he - hs) * (we - ws)));
  }
}
// 4 channels per thread (16-B loads), two independent accumulators per row pair so the adds do not serialise the loads.
// The summation order over a window stays row-major like the scalar kernel's (rows alternate between the accumulators).
template <class T>
__global__ void adaptive_avgpool_vec4_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int h,
                                             int w, int c4, int ho, int wo) {
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c4) * 4;
    size_t pix = i / c4;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    const int hs = (oh * h) / ho, he = ((oh + 1) * h + ho - 1) / ho;
    const int ws = (ow * w) / wo, we = ((ow + 1) * w + wo - 1) / wo;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = hs; iy < he; ++iy) {
      const T* row = x + ((img * h + iy) * w + ws) * (size_t)ldx + ch;
      int ix = 0;
      const int cnt = we - ws;
      for (; ix + 4 <= cnt; ix += 4) {
        const float4 a = ld4(row + (size_t)ix * ldx);
        const float4 b = ld4(row + (size_t)(ix + 1) * ldx);
        const float4 d = ld4(row + (size_t)(ix + 2) * ldx);
        const float4 e = ld4(row + (size_t)(ix + 3) * ldx);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
      }
      for (; ix < cnt; ++ix) {
        const float4 a = ld4(row + (size_t)ix * ldx);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
      }
    }
    const float d = (float)((he - hs) * (we - ws));
    st4(y + ((img * ho + oh) * wo + ow) * (size_t)ldy + ch, make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d));
  }
}
template <class T>
void launch_adaptive_avgpool(const TensorT<T>& x, const TensorT<T>& y, hipStream_t s) {
  FE_CHECK(x.c == y.c && x.n == y.n, "adaptive_avgpool: shape mismatch");
  if (x.c % 4 == 0 && x.ld % 4 == 0 && y.ld % 4 == 0 && vec4_ok(x.p, y.p)) {
    const size_t work = y.pixels() * (size_t)(y.c / 4);
    size_t g = (work + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(adaptive_avgpool_vec4_kernel<T>, dim3((unsigned)g), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c / 4, y.h, y.w);
    FE_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(adaptive_avgpool_kernel<T>, dim3(grid_for(y.numel())), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.n, x.h, x.w, x.c, y.h, y.w);
  FE_HIP(hipGetLastError());
}

// ---- elementwise ---------------------------------------------------------------------------------------
__global__ void sigmoid_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, size_t pixels, int c) {
  const size_t total = pixels * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / c; const int ch = i % c;
    y[pix * ldy + ch] = 1.f / (1.f + expf(-x[pix * ldx + ch]));
  }
}
void launch_sigmoid(const Tensor& x, const Tensor& y, hipStream_t s) {
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(x.numel())), dim3(256), 0, s, x.p, x.ld, y.p, y.ld, x.pixels(), x.c);
  FE_HIP(hipGetLastError());
}
__global__ void add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ y,
                           int ldy, size_t pixels, int c) {
  const size_t total = pixels * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / c; const int ch = i % c;
    y[pix * ldy + ch] = a[pix * lda + ch] + b[pix * ldb + ch];
  }
}
void launch_add(const Tensor& a, const Tensor& b, const Tensor& y, hipStream_t s) {
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(a.numel())), dim3(256), 0, s, a.p, a.ld, b.p, b.ld, y.p, y.ld, a.pixels(), a.c);
  FE_HIP(hipGetLastError());
}

// ---- narrow-output convolution (Cout <= 4): HBM-bound dot products, no matrix cores -------------------------
// 16 lanes share one output pixel (each owns 4 input channels per 64-channel chunk), 4 pixels per wave instruction;
// partial sums are reduced with xor-shuffles. Used for the Cout=1 gate / side / fuse convolutions (TOPIQ GatedConv
// weight_blk[4], U2-Net-P side1-6 + outconv) where a 32-wide MFMA tile would waste 31/32 of the matrix work.
template <int CO>
__global__ void conv_narrow_kernel(const ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 15, grp = lane >> 4;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const int HoWo = p.Ho * p.Wo;
  for (long long m4 = wave * 4; m4 < p.M; m4 += nwaves * 4) {
    const long long m = m4 + grp;
    const bool mv = m < p.M;
    const int mm = mv ? (int)m : 0;
    const int nimg = mm / HoWo;
    const int rem = mm - nimg * HoWo;
    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = oh * p.sh - p.ph + kh * p.dh;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = ow * p.sw - p.pw + kw * p.dw;
        const bool ok = mv && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const float* xp = p.x + (((size_t)nimg * p.H + (ok ? ih : 0)) * p.W + (ok ? iw : 0)) * p.ldx;
        const float* wp = p.w + (size_t)(kh * p.KW + kw) * p.Cin;
        for (int c = sub * 4; c < p.Cin; c += 64) {
          float4 xv = *reinterpret_cast<const float4*>(xp + c);
          if (!ok) xv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const float4 wv = *reinterpret_cast<const float4*>(wp + (size_t)o * p.ldw + c);
            acc[o] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      float v = acc[o];
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
      if (sub == 0 && mv && o < p.Cout) {
        v = v * (p.scale ? p.scale[o] : 1.f) + (p.shift ? p.shift[o] : 0.f);
        if (p.res && !p.res_after_act) v += p.res[(size_t)m * p.ldr + o];
        if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        else if (p.act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
        else if (p.act == ACT_PRELU) v = v > 0.f ? v : v * p.slope[o];
        if (p.res && p.res_after_act) v += p.res[(size_t)m * p.ldr + o];
        p.y[(size_t)m * p.ldy + o] = v;
      }
    }
  }
}
void launch_conv_narrow(const ConvParams& p, hipStream_t s) {
  FE_CHECK(p.Cout <= 4 && p.Cin % 4 == 0 && p.batch <= 1 && !p.gate, "conv_narrow: unsupported problem");
  const int blocks = grid_for((size_t)((p.M + 3) / 4) * 64);
  if (p.Cout == 1) hipLaunchKernelGGL(conv_narrow_kernel<1>, dim3(blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(conv_narrow_kernel<4>, dim3(blocks), dim3(256), 0, s, p);
  FE_HIP(hipGetLastError());
}

// ---- second half of the tap-decomposed narrow convolution: sum the per-tap partial products of the neighbours ----
template <class T>
__global__ void tap_gather_kernel(const T* __restrict__ z, int ldz, int n, int h, int w, int kh, int kw, int ph, int pw,
                                  int dh, int dw, int cout, const float* __restrict__ scale, const float* __restrict__ shift,
                                  int act, T* __restrict__ y, int ldy, int ho, int wo) {
  const size_t total = (size_t)n * ho * wo * cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % cout;
    size_t pix = i / cout;
    const int ow = pix % wo; pix /= wo;
    const int oh = pix % ho;
    const size_t img = pix / ho;
    float acc = 0.f;
    for (int a = 0; a < kh; ++a) {
      const int ih = oh - ph + a * dh;
      if ((unsigned)ih >= (unsigned)h) continue;
      for (int b = 0; b < kw; ++b) {
        const int iw = ow - pw + b * dw;
        if ((unsigned)iw >= (unsigned)w) continue;
        acc += ldf(z + ((img * h + ih) * w + iw) * ldz + (a * kw + b) * cout + co);
      }
    }
    float v = acc * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
    v = fe_apply_act(v, act);
    stf(y + ((img * ho + oh) * wo + ow) * ldy + co, v);
  }
}
template <class T>
void launch_tap_gather(const T* z, int ldz, int n, int h, int w, int kh, int kw, int ph, int pw, int dh, int dw,
                       int cout, const float* scale, const float* shift, int act, T* y, int ldy, int ho, int wo,
                       hipStream_t s) {
  hipLaunchKernelGGL(tap_gather_kernel<T>, dim3(grid_for((size_t)n * ho * wo * cout)), dim3(256), 0, s, z, ldz, n, h, w, kh, kw,
                     ph, pw, dh, dw, cout, scale, shift, act, y, ldy, ho, wo);
  FE_HIP(hipGetLastError());
}

// ---- skinny GEMM (M <= 32 rows): y[m][n] = act(sum_k x[m][k] w[n][k] * scale[n] + shift[n]) -----------------------
// One wave per output column streams its weight row once (16 B per lane); the <= 32 activation rows come from L1/L2.
// Used for the per-image vectors of the heads (SAMP pattern "convs" K up to 7524, score MLPs, CLIP projection) where a
// 128-row MFMA tile would leave 255 of 256 CUs idle behind a K loop thousands of steps long.
// TI / TW / TO: element types of the activations, the weight rows and the outputs (bf16 activations with bf16 weights can
// still emit fp32: the last layer of every head does).
template <int MR, class TI, class TW, class TO>
__global__ void gemm_skinny_kernel(const TI* __restrict__ x, int ldx, const TW* __restrict__ w, int ldw,
                                   const float* __restrict__ scale, const float* __restrict__ shift, TO* __restrict__ y,
                                   int ldy, int M, int N, int K, int act) {
  const int lane = threadIdx.x & 63;
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (n >= N) return;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const TW* wr = w + (size_t)n * ldw;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 wv = ld4(wr + k);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < M) {
        const float4 xv = ld4(x + (size_t)m * ldx + k);
        acc[m] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float v = acc[m];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    acc[m] = v;
  }
  if (lane == 0) {
    const float sc = scale ? scale[n] : 1.f, sf = shift ? shift[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < M) stf(y + (size_t)m * ldy + n, fe_apply_act(acc[m] * sc + sf, act));
    }
  }
}
template <class TI, class TW, class TO>
void launch_gemm_skinny(const TI* x, int ldx, const TW* w, int ldw, const float* scale, const float* shift, TO* y,
                        int ldy, int M, int N, int K, int act, hipStream_t s) {
  FE_CHECK(M >= 1 && M <= 32 && K % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0, "gemm_skinny: M=%d K=%d", M, K);
  FE_CHECK(((uintptr_t)x & (4 * sizeof(TI) - 1)) == 0 && ((uintptr_t)w & (4 * sizeof(TW) - 1)) == 0, "gemm_skinny: alignment");
  const int blocks = (N * 64 + 255) / 256;
  if (M <= 8) hipLaunchKernelGGL((gemm_skinny_kernel<8, TI, TW, TO>), dim3(blocks), dim3(256), 0, s, x, ldx, w, ldw, scale, shift, y, ldy, M, N, K, act);
  else hipLaunchKernelGGL((gemm_skinny_kernel<32, TI, TW, TO>), dim3(blocks), dim3(256), 0, s, x, ldx, w, ldw, scale, shift, y, ldy, M, N, K, act);
  FE_HIP(hipGetLastError());
}

// ---- LayerNorm: one wave per row, two-pass (mean, then centred variance) in registers ------------------
template <class T, class TO>
__global__ void layernorm_kernel(const T* __restrict__ x, int ldx, TO* __restrict__ y, int ldy,
                                 const float* __restrict__ g, const float* __restrict__ b, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    const T* xr = x + (size_t)row * ldx;
    float sum = 0.f;
    for (int i = lane; i < d; i += 64) sum += ldf(xr + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)d;
    float var = 0.f;
    for (int i = lane; i < d; i += 64) { const float t = ldf(xr + i) - mean; var += t * t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)d + eps);
    TO* yr = y + (size_t)row * ldy;
    for (int i = lane; i < d; i += 64) stf(yr + i, (ldf(xr + i) - mean) * rstd * g[i] + b[i]);
  }
}
// Register-resident form for d = 256 * NV (the ViT / CFANet widths 256, 768, 1024): one wave per row, every lane holds 4 * NV
// consecutive-by-256 elements (8- or 16-byte loads: whole 512-B / 1-KiB segments per wave instruction), ONE read and one write of
// the row; mean, then centred variance, both in fp32 registers.
template <class T, class TO, int NV>
__global__ void layernorm_reg_kernel(const T* __restrict__ x, int ldx, TO* __restrict__ y, int ldy,
                                     const float* __restrict__ g, const float* __restrict__ b, int rows, float eps) {
  constexpr int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float4 gv[NV], bv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    gv[k] = *reinterpret_cast<const float4*>(g + 256 * k + 4 * lane);
    bv[k] = *reinterpret_cast<const float4*>(b + 256 * k + 4 * lane);
  }
  for (int row = wave; row < rows; row += nwaves) {
    const T* xr = x + (size_t)row * ldx;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { v[k] = ld4(xr + 256 * k + 4 * lane); sum += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)d;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
      var += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)d + eps);
    TO* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      st4(yr + 256 * k + 4 * lane, make_float4(v[k].x * rstd * gv[k].x + bv[k].x, v[k].y * rstd * gv[k].y + bv[k].y,
                                               v[k].z * rstd * gv[k].z + bv[k].z, v[k].w * rstd * gv[k].w + bv[k].w));
  }
}
template <class T, class TO>
void launch_layernorm(const T* x, int ldx, TO* y, int ldy, const float* g, const float* b, int rows, int d,
                      float eps, hipStream_t s) {
  const int blocks = grid_for((size_t)rows * 64);
  const bool vec = d % 256 == 0 && d <= 1024 && ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x & (4 * sizeof(T) - 1)) == 0 && ((uintptr_t)y & (4 * sizeof(TO) - 1)) == 0 &&
                   ((((uintptr_t)g | (uintptr_t)b) & 15) == 0);
  if (vec && d == 256) hipLaunchKernelGGL((layernorm_reg_kernel<T, TO, 1>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else if (vec && d == 512) hipLaunchKernelGGL((layernorm_reg_kernel<T, TO, 2>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else if (vec && d == 768) hipLaunchKernelGGL((layernorm_reg_kernel<T, TO, 3>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else if (vec && d == 1024) hipLaunchKernelGGL((layernorm_reg_kernel<T, TO, 4>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else hipLaunchKernelGGL((layernorm_kernel<T, TO>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, d, eps);
  FE_HIP(hipGetLastError());
}

// LayerNorm of an fp32 row written as an fp16 pair per element (split-operand GEMMs, ClipModel::split3): one wave per row, the row in
// registers (d = 256 * NV), statistics in fp32 as layernorm_reg_kernel
template <int NV>
__global__ void layernorm_split_kernel(const float* __restrict__ x, int ldx, f16* __restrict__ y, int ldy, const float* __restrict__ g, const float* __restrict__ b,
                                       int rows, float eps) {
  constexpr int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    const float* xr = x + (size_t)row * ldx;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { v[k] = *reinterpret_cast<const float4*>(xr + 256 * k + 4 * lane); sum += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)d;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
      var += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)d + eps);
    f16* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float4 gv = *reinterpret_cast<const float4*>(g + 256 * k + 4 * lane), bv = *reinterpret_cast<const float4*>(b + 256 * k + 4 * lane);
      const float4 o = make_float4(v[k].x * rstd * gv.x + bv.x, v[k].y * rstd * gv.y + bv.y, v[k].z * rstd * gv.z + bv.z, v[k].w * rstd * gv.w + bv.w);
      const float4 hi = make_float4((float)fe_to_f16(o.x), (float)fe_to_f16(o.y), (float)fe_to_f16(o.z), (float)fe_to_f16(o.w));
      st4(yr + 256 * k + 4 * lane, hi);
      st4(yr + d + 256 * k + 4 * lane, make_float4(o.x - hi.x, o.y - hi.y, o.z - hi.z, o.w - hi.w));
    }
  }
}
void launch_layernorm_split(const float* x, int ldx, f16* y, int ldy, const float* g, const float* b, int rows, int d, float eps, hipStream_t s) {
  FE_CHECK((d == 1024 || d == 768 || d == 512 || d == 256) && ldx % 4 == 0 && ldy % 4 == 0 && ldy >= 2 * d && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 &&
           (((uintptr_t)g | (uintptr_t)b) & 15) == 0, "layernorm_split: d = %d / strides / alignment", d);
  const int blocks = grid_for((size_t)rows * 64);
  if (d == 1024) hipLaunchKernelGGL((layernorm_split_kernel<4>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else if (d == 768) hipLaunchKernelGGL((layernorm_split_kernel<3>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else if (d == 512) hipLaunchKernelGGL((layernorm_split_kernel<2>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  else hipLaunchKernelGGL((layernorm_split_kernel<1>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows, eps);
  FE_HIP(hipGetLastError());
}
__global__ void split_hi_lo_kernel(const float* __restrict__ x, f16* __restrict__ y, size_t rows, int cols) {
  const size_t total = rows * (size_t)(cols / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (cols / 4);
    const int c = (int)(i - row * (cols / 4)) * 4;
    const float4 o = *reinterpret_cast<const float4*>(x + row * cols + c);
    const float4 hi = make_float4((float)fe_to_f16(o.x), (float)fe_to_f16(o.y), (float)fe_to_f16(o.z), (float)fe_to_f16(o.w));
    f16* yr = y + row * 2 * cols;
    st4(yr + c, hi);
    st4(yr + cols + c, make_float4(o.x - hi.x, o.y - hi.y, o.z - hi.z, o.w - hi.w));
  }
}
void launch_split_hi_lo(const float* x, f16* y, size_t rows, int cols, hipStream_t s) {
  FE_CHECK(cols % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, "split_hi_lo: alignment");
  hipLaunchKernelGGL(split_hi_lo_kernel, dim3(grid_for(rows * (size_t)(cols / 4))), dim3(256), 0, s, x, y, rows, cols);
  FE_HIP(hipGetLastError());
}

// ---- row softmax, one wave per row ---------------------------------------------------------------------
__global__ void softmax_rows_kernel(float* __restrict__ x, int ld, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    float* xr = x + (size_t)row * ld;
    float mx = -INFINITY;
    for (int i = lane; i < d; i += 64) mx = fmaxf(mx, xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int i = lane; i < d; i += 64) { const float e = expf(xr[i] - mx); xr[i] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int i = lane; i < d; i += 64) xr[i] *= inv;
  }
}
__global__ void softmax_rows_pad_kernel(float* __restrict__ x, int ld, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    float* xr = x + (size_t)row * ld;
    float mx = -INFINITY;
    for (int i = lane; i < d; i += 64) mx = fmaxf(mx, xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int i = lane; i < d; i += 64) { const float e = expf(xr[i] - mx); xr[i] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int i = lane; i < ld; i += 64) xr[i] = i < d ? xr[i] * inv : 0.f;
  }
}
// Register-resident variant (ld <= 64*NV*4, ld % 4 == 0): one 16-B read and one 16-B write per element group.
template <int NV>
__global__ void softmax_rows_pad_reg_kernel(float* __restrict__ x, int ld, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * ld);
    float4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < ld) {
        v[k] = xr[lane + 64 * k];
        if (c + 0 < d) mx = fmaxf(mx, v[k].x);
        if (c + 1 < d) mx = fmaxf(mx, v[k].y);
        if (c + 2 < d) mx = fmaxf(mx, v[k].z);
        if (c + 3 < d) mx = fmaxf(mx, v[k].w);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < ld) {
        v[k].x = c + 0 < d ? expf(v[k].x - mx) : 0.f;
        v[k].y = c + 1 < d ? expf(v[k].y - mx) : 0.f;
        v[k].z = c + 2 < d ? expf(v[k].z - mx) : 0.f;
        v[k].w = c + 3 < d ? expf(v[k].w - mx) : 0.f;
        sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < ld) xr[lane + 64 * k] = make_float4(v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);
    }
  }
}
void launch_softmax_rows_pad(float* x, int ld, int rows, int d, hipStream_t s) {
  const bool vec = (ld % 4 == 0) && (((uintptr_t)x & 15) == 0);
  const int blocks = grid_for((size_t)rows * 64);
  if (vec && ld <= 64 * 4 * 2)
    hipLaunchKernelGGL(softmax_rows_pad_reg_kernel<2>, dim3(blocks), dim3(256), 0, s, x, ld, rows, d);
  else if (vec && ld <= 64 * 4 * 8)
    hipLaunchKernelGGL(softmax_rows_pad_reg_kernel<8>, dim3(blocks), dim3(256), 0, s, x, ld, rows, d);
  else
    hipLaunchKernelGGL(softmax_rows_pad_kernel, dim3(blocks), dim3(256), 0, s, x, ld, rows, d);
  FE_HIP(hipGetLastError());
}
template <class T>
__global__ void add_rows_bcast_kernel(T* __restrict__ y, int ldy, const float* __restrict__ pos, size_t rows, int L, int d) {
  const size_t total = rows * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / d; const int ch = i % d;
    stf(y + row * ldy + ch, ldf(y + row * ldy + ch) + pos[(row % L) * d + ch]);
  }
}
template <class T>
void launch_add_rows_bcast(T* y, int ldy, const float* pos, int rows, int L, int d, hipStream_t s) {
  hipLaunchKernelGGL(add_rows_bcast_kernel<T>, dim3(grid_for((size_t)rows * d)), dim3(256), 0, s, y, ldy, pos, (size_t)rows, L, d);
  FE_HIP(hipGetLastError());
}
void launch_softmax_rows(float* x, int ld, int rows, int d, hipStream_t s) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid_for((size_t)rows * 64)), dim3(256), 0, s, x, ld, rows, d);
  FE_HIP(hipGetLastError());
}

// Sum of split-K partial products + the convolution epilogue: part [splits][M][N] (dense) -> y [M][ldy].
// Fixed summation order (split 0, 1, 2, ...), so results do not depend on scheduling. act: ACT_NONE / ACT_RELU / ACT_PRELU.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const float* __restrict__ slope, int act, const float* __restrict__ res,
                                     int ldr, int res_after_act, float* __restrict__ y, int ldy) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) acc += part[(size_t)s * M * N + i];
  float v = acc * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f);
  if (res && !res_after_act) v += res[(size_t)m * ldr + n];
  if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
  else if (act == ACT_PRELU) v = v > 0.f ? v : v * slope[n];
  if (res && res_after_act) v += res[(size_t)m * ldr + n];
  y[(size_t)m * ldy + n] = v;
}

void launch_splitk_reduce(const float* part, int splits, int M, int N, const float* scale, const float* shift, const float* slope, int act,
                          const float* res, int ldr, int res_after_act, float* y, int ldy, hipStream_t s) {
  FE_CHECK(act == ACT_NONE || act == ACT_RELU || (act == ACT_PRELU && slope), "splitk_reduce: unsupported activation %d", act);
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, splits, M, N, scale, shift, slope, act, res,
                     ldr, res_after_act, y, ldy);
  FE_HIP(hipGetLastError());
}

// ---- fp32 <-> bf16 copies (boundary of the bf16 path: fp32 first-layer outputs, fp32 host tensors) -------------------------------
template <class TI, class TO>
__global__ void convert_kernel(const TI* __restrict__ x, TO* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) stf(y + i, ldf(x + i));
}
void launch_convert(const float* x, bf16* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL((convert_kernel<float, bf16>), dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  FE_HIP(hipGetLastError());
}
void launch_convert(const bf16* x, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL((convert_kernel<bf16, float>), dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  FE_HIP(hipGetLastError());
}
void launch_convert(const float* x, f16* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL((convert_kernel<float, f16>), dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  FE_HIP(hipGetLastError());
}
void launch_convert(const f16* x, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL((convert_kernel<f16, float>), dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  FE_HIP(hipGetLastError());
}

#define FE_INST_T(T)                                                                                                              \
  template void launch_nchw_to_nhwc<T>(const float*, T*, int, int, int, int, int, hipStream_t);                                  \
  template void launch_nhwc_to_nchw<T>(const T*, int, float*, int, int, int, int, hipStream_t);                                  \
  template void launch_maxpool<T>(const TensorT<T>&, const TensorT<T>&, int, int, int, hipStream_t);                             \
  template void launch_bilinear<T>(const TensorT<T>&, const TensorT<T>&, hipStream_t);                                           \
  template void launch_adaptive_avgpool<T>(const TensorT<T>&, const TensorT<T>&, hipStream_t);                                   \
  template void launch_tap_gather<T>(const T*, int, int, int, int, int, int, int, int, int, int, int, const float*, const float*, int, T*, int, int, int, hipStream_t); \
  template void launch_layernorm<T, T>(const T*, int, T*, int, const float*, const float*, int, int, float, hipStream_t);        \
  template void launch_add_rows_bcast<T>(T*, int, const float*, int, int, int, hipStream_t);
FE_INST_T(float)
FE_INST_T(bf16)
FE_INST_T(f16)
#undef FE_INST_T
// fp32 residual stream in front of a 2-byte GEMM operand (precision option FE_PRECISION_RES32): LayerNorm reads fp32, writes 2 bytes
template void launch_layernorm<float, bf16>(const float*, int, bf16*, int, const float*, const float*, int, int, float, hipStream_t);
template void launch_layernorm<float, f16>(const float*, int, f16*, int, const float*, const float*, int, int, float, hipStream_t);
template void launch_gemm_skinny<float, float, float>(const float*, int, const float*, int, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template void launch_gemm_skinny<bf16, bf16, bf16>(const bf16*, int, const bf16*, int, const float*, const float*, bf16*, int, int, int, int, int, hipStream_t);
template void launch_gemm_skinny<bf16, bf16, float>(const bf16*, int, const bf16*, int, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template void launch_gemm_skinny<f16, f16, f16>(const f16*, int, const f16*, int, const float*, const float*, f16*, int, int, int, int, int, hipStream_t);
template void launch_gemm_skinny<f16, f16, float>(const f16*, int, const f16*, int, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template void launch_gemm_skinny<float, bf16, float>(const float*, int, const bf16*, int, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template void launch_gemm_skinny<float, f16, float>(const float*, int, const f16*, int, const float*, const float*, float*, int, int, int, int, int, hipStream_t);

// RGB <-> BGR of a packed uint8 image batch (the reference keeps a PIL RGB and a cv2 BGR copy of every image,
// processing/batch_processor.py:200-215; here the second one is made on the device from the resident first one).
// Four pixels = three aligned 32-bit words per thread.
__global__ void swap_rb_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t pixels) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // group of 4 pixels
  const size_t p0 = q * 4;
  if (p0 >= pixels) return;
  if (p0 + 4 <= pixels && ((((uintptr_t)src) | ((uintptr_t)dst)) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src) + q * 3;
    const uint32_t a = s4[0], b = s4[1], c = s4[2];
    uint8_t v[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (a >> (8 * k)) & 255; v[4 + k] = (b >> (8 * k)) & 255; v[8 + k] = (c >> (8 * k)) & 255; }
    auto pack = [&](int i0, int i1, int i2, int i3) { return (uint32_t)v[i0] | ((uint32_t)v[i1] << 8) | ((uint32_t)v[i2] << 16) | ((uint32_t)v[i3] << 24); };
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst) + q * 3;
    d4[0] = pack(2, 1, 0, 5);
    d4[1] = pack(4, 3, 8, 7);
    d4[2] = pack(6, 11, 10, 9);
  } else {
    for (size_t p = p0; p < pixels && p < p0 + 4; ++p) {
      const uint8_t r = src[p * 3], g = src[p * 3 + 1], b = src[p * 3 + 2];
      dst[p * 3] = b; dst[p * 3 + 1] = g; dst[p * 3 + 2] = r;
    }
  }
}

void launch_swap_rb_u8(const uint8_t* src, uint8_t* dst, size_t pixels, hipStream_t s) {
  if (!pixels) return;
  const size_t groups = (pixels + 3) / 4;
  FE_CHECK(groups < (1ull << 31) * 256, "swap_rb: too many pixels");
  hipLaunchKernelGGL(swap_rb_u8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, src, dst, pixels);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

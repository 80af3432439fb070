// PIL-exact uint8 image resampling on the GPU (bit-for-bit `PIL.Image.resize` for RGB 8-bit images).
//
// The reference preprocesses with PIL on the CPU:
//   * TOPIQ: `image.resize((int(w*s), int(h*s)), Image.LANCZOS)` when the long edge > 1024 (models/pyiqa_scorer.py:150-153)
//   * SAMP-Net: torchvision `transforms.Resize((224,224))` = PIL BILINEAR (models/samp_net.py:823-830)
//   * CLIP: open_clip eval transform = PIL BICUBIC shorter-side resize to 224 + center crop [DEP-KNOWLEDGE]
// PIL's resampler (libImaging/Resample.c) is separable (horizontal pass, then vertical pass over the uint8
// intermediate), antialiased (support scaled by the reduction factor) and integer: coefficients are normalised in double,
// converted to fixed point with 22 fractional bits, accumulated in int32 from 1<<21 and shifted/clipped to uint8.
// Here the coefficient tables are built on the host with the same double arithmetic and the two passes run as HIP
// kernels with the same int32 arithmetic, so results are bit-exact (tests/test_resize_gpu.py compares with PIL itself).
#include "engine.h"
#include <cmath>

namespace fe {

static double filt_bilinear(double x) { if (x < 0.0) x = -x; return x < 1.0 ? 1.0 - x : 0.0; }
static double filt_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static double sinc_filter(double x) { if (x == 0.0) return 1.0; x = x * M_PI; return std::sin(x) / x; }
static double filt_lanczos(double x) { return (-3.0 <= x && x < 3.0) ? sinc_filter(x) * sinc_filter(x / 3) : 0.0; }

constexpr int PRECISION_BITS = 32 - 8 - 2;

void build_resize_coeffs(int in_size, int out_size, int filter, ResizeCoeffs& rc) {
  double (*f)(double);
  double fsupport;
  switch (filter) {
    case FE_FILTER_BILINEAR: f = filt_bilinear; fsupport = 1.0; break;
    case FE_FILTER_BICUBIC: f = filt_bicubic; fsupport = 2.0; break;
    case FE_FILTER_LANCZOS: f = filt_lanczos; fsupport = 3.0; break;
    default: throw Error("resize: unknown filter " + std::to_string(filter));
  }
  const double in0 = 0.0, in1 = (double)in_size;
  double scale, filterscale;
  filterscale = scale = (in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = fsupport * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  rc.ksize = ksize;
  rc.out = out_size;
  rc.kk.assign((size_t)out_size * ksize, 0);
  rc.bounds.assign((size_t)out_size * 2, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = f((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < xmax; ++x) {
      const double v = k[x];
      rc.kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    rc.bounds[xx * 2] = xmin;
    rc.bounds[xx * 2 + 1] = xmax;
  }
}

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [n][h][w][3] -> dst [n][h][ow][3]
__global__ void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ kk,
                                const int* __restrict__ bounds, int ksize, size_t rows, int w, int ow) {
  const size_t total = rows * ow;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = i % ow;
    const size_t row = i / ow;
    const int xmin = bounds[xx * 2], xmax = bounds[xx * 2 + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* p = src + (row * w + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
      const int c = k[x];
      s0 += p[x * 3 + 0] * c; s1 += p[x * 3 + 1] * c; s2 += p[x * 3 + 2] * c;
    }
    uint8_t* o = dst + i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
}

// src [n][h][w][3] -> dst [n][oh][w][3], optionally cropped to columns [x0, x0+cw) and rows [y0, y0+ch) of the output
__global__ void resize_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ kk,
                                const int* __restrict__ bounds, int ksize, int n, int h, int w, int y0, int ch, int x0,
                                int cw) {
  const size_t total = (size_t)n * ch * cw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % cw;
    const int yy = (i / cw) % ch;
    const size_t img = i / ((size_t)cw * ch);
    const int oy = yy + y0;
    const int ymin = bounds[oy * 2], ymax = bounds[oy * 2 + 1];
    const int* k = kk + (size_t)oy * ksize;
    const uint8_t* p = src + ((img * h + ymin) * w + (x + x0)) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) {
      const int c = k[y];
      const uint8_t* q = p + (size_t)y * w * 3;
      s0 += q[0] * c; s1 += q[1] * c; s2 += q[2] * c;
    }
    uint8_t* o = dst + i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
}

__global__ void crop_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w, int y0,
                               int ch, int x0, int cw) {
  const size_t total = (size_t)n * ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % 3;
    const int x = (i / 3) % cw;
    const int y = (i / (3 * (size_t)cw)) % ch;
    const size_t img = i / ((size_t)3 * cw * ch);
    dst[i] = src[((img * h + y + y0) * w + x + x0) * 3 + c];
  }
}

static const ResizeCoeffsDev& coeffs_dev(Ctx& c, int in_size, int out_size, int filter) {
  const auto key = std::make_tuple(in_size, out_size, filter);
  auto it = c.resize_cache.find(key);
  if (it != c.resize_cache.end()) return it->second;
  ResizeCoeffs rc;
  build_resize_coeffs(in_size, out_size, filter, rc);
  ResizeCoeffsDev d;
  d.ksize = rc.ksize;
  FE_HIP(hipMalloc((void**)&d.kk, rc.kk.size() * sizeof(int)));
  FE_HIP(hipMalloc((void**)&d.bounds, rc.bounds.size() * sizeof(int)));
  FE_HIP(hipMemcpy(d.kk, rc.kk.data(), rc.kk.size() * sizeof(int), hipMemcpyHostToDevice));
  FE_HIP(hipMemcpy(d.bounds, rc.bounds.data(), rc.bounds.size() * sizeof(int), hipMemcpyHostToDevice));
  return c.resize_cache.emplace(key, d).first->second;
}

static inline int grid_sz(size_t work) { size_t g = (work + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

// d_src [n][h][w][3] u8 (device) -> d_dst [n][ch][cw][3]: PIL resize to (ow, oh) then crop (x0,y0,cw,ch).
// Both buffers and the intermediate come from the caller / arena; PIL order: horizontal pass first.
void resize_u8(Ctx& c, const uint8_t* d_src, int n, int h, int w, int oh, int ow, int filter, int y0, int ch, int x0,
               int cw, uint8_t* d_dst) {
  FE_CHECK(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && y0 >= 0 && x0 >= 0 && y0 + ch <= oh && x0 + cw <= ow, "resize: bad geometry");
  const bool need_h = ow != w, need_v = oh != h;
  const uint8_t* cur = d_src;
  int cur_w = w;
  const size_t mark = c.arena.mark();
  if (need_h) {
    const ResizeCoeffsDev& ch_ = coeffs_dev(c, w, ow, filter);
    uint8_t* tmp = (uint8_t*)c.arena.alloc((size_t)n * h * ow * 3);
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_sz((size_t)n * h * ow)), dim3(256), 0, c.stream, cur, tmp, ch_.kk,
                       ch_.bounds, ch_.ksize, (size_t)n * h, w, ow);
    FE_HIP(hipGetLastError());
    cur = tmp; cur_w = ow;
  }
  if (need_v) {
    const ResizeCoeffsDev& cv = coeffs_dev(c, h, oh, filter);
    hipLaunchKernelGGL(resize_v_kernel, dim3(grid_sz((size_t)n * ch * cw)), dim3(256), 0, c.stream, cur, d_dst, cv.kk,
                       cv.bounds, cv.ksize, n, h, cur_w, y0, ch, x0, cw);
  } else {
    hipLaunchKernelGGL(crop_u8_kernel, dim3(grid_sz((size_t)n * ch * cw * 3)), dim3(256), 0, c.stream, cur, d_dst, n, h,
                       cur_w, y0, ch, x0, cw);
  }
  FE_HIP(hipGetLastError());
  c.arena.rewind(mark);  // the intermediate is only read by the kernel just queued on this same stream
}

}  // namespace fe

// Leading-line detection (reference analyzers/composition.py:191-261: cv2.GaussianBlur(gray, (5,5), 0) ->
// cv2.Canny(50, 150) -> cv2.HoughLinesP(1, pi/180, 80, minLineLength, maxLineGap=20)).
//
// The pixel scans run on the GPU over the resident BGR batch (three small HBM-bound kernels, ~10 bytes of traffic per pixel):
//   lines_blur_kernel    BGR -> gray (OpenCV's 15-bit weights) -> 5x5 binomial blur, the fixed-point path OpenCV takes for
//                        8-bit images: kernel [1 4 6 4 1]/16 per axis, (sum + 128) >> 8, BORDER_REFLECT_101
//   lines_sobel_kernel   3x3 Sobel dx, dy (BORDER_REPLICATE, as cv::Canny asks for) and the L1 magnitude |dx| + |dy|
//   lines_nms_kernel     non-maximum suppression along the quantised gradient direction (tan 22.5 in 15-bit fixed point) and
//                        the two thresholds -> map: 2 = edge for sure, 0 = edge if connected to a 2, 1 = not an edge
// The rest (lines_host.cpp) is inherently sequential and runs on the host, one image per thread: hysteresis (flood fill from the 2s) and the
// progressive probabilistic Hough transform, whose result depends on the order a fixed pseudo-random generator visits the
// edge points in (vote, extract a segment as soon as a bin reaches the threshold, erase its points and un-vote them).
// [DEP-KNOWLEDGE] OpenCV is not available offline: these restate its documented algorithms; parity with cv2 is unpinned.
#include "fe_common.h"

namespace fe {

__device__ __forceinline__ int lines_gray(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }
__device__ __forceinline__ int lines_refl101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}
__device__ __forceinline__ int lines_clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

constexpr int LB_W = 32, LB_H = 8;

__global__ __launch_bounds__(LB_W * LB_H) void lines_blur_kernel(const uint8_t* __restrict__ bgr, int h, int w, uint8_t* __restrict__ blur) {
  __shared__ int g[LB_H + 4][LB_W + 4];
  const size_t img = blockIdx.z;
  const uint8_t* src = bgr + img * (size_t)h * w * 3;
  const int x0 = blockIdx.x * LB_W - 2, y0 = blockIdx.y * LB_H - 2;
  for (int i = threadIdx.y * LB_W + threadIdx.x; i < (LB_H + 4) * (LB_W + 4); i += LB_W * LB_H) {
    const int ty = i / (LB_W + 4), tx = i - ty * (LB_W + 4);
    const int y = lines_refl101(y0 + ty, h), x = lines_refl101(x0 + tx, w);
    const uint8_t* p = src + ((size_t)y * w + x) * 3;
    g[ty][tx] = lines_gray(p[0], p[1], p[2]);
  }
  __syncthreads();
  const int x = blockIdx.x * LB_W + threadIdx.x, y = blockIdx.y * LB_H + threadIdx.y;
  if (x >= w || y >= h) return;
  const int k[5] = {1, 4, 6, 4, 1};
  int acc = 0;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    int row = 0;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) row += k[dx] * g[threadIdx.y + dy][threadIdx.x + dx];
    acc += k[dy] * row;
  }
  blur[img * (size_t)h * w + (size_t)y * w + x] = (uint8_t)((acc + 128) >> 8);
}

__global__ void lines_sobel_kernel(const uint8_t* __restrict__ blur, int h, int w, short2* __restrict__ grad, unsigned short* __restrict__ mag) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const size_t base = (size_t)blockIdx.z * h * w;
  const uint8_t* s = blur + base;
  const int xm = lines_clampi(x - 1, w), xp = lines_clampi(x + 1, w), ym = lines_clampi(y - 1, h), yp = lines_clampi(y + 1, h);
  const int a = s[(size_t)ym * w + xm], b = s[(size_t)ym * w + x], c = s[(size_t)ym * w + xp];
  const int d = s[(size_t)y * w + xm], f = s[(size_t)y * w + xp];
  const int g = s[(size_t)yp * w + xm], hh = s[(size_t)yp * w + x], i = s[(size_t)yp * w + xp];
  const int dx = (c + 2 * f + i) - (a + 2 * d + g), dy = (g + 2 * hh + i) - (a + 2 * b + c);
  grad[base + (size_t)y * w + x] = make_short2((short)dx, (short)dy);
  mag[base + (size_t)y * w + x] = (unsigned short)(abs(dx) + abs(dy));
}

__global__ void lines_nms_kernel(const short2* __restrict__ grad, const unsigned short* __restrict__ mag, int h, int w, int low, int high,
                                 uint8_t* __restrict__ map) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const size_t base = (size_t)blockIdx.z * h * w;
  const unsigned short* mg = mag + base;
  auto M = [&](int yy, int xx) -> int { return (xx < 0 || xx >= w || yy < 0 || yy >= h) ? 0 : (int)mg[(size_t)yy * w + xx]; };   // zero outside
  const int m = mg[(size_t)y * w + x];
  uint8_t out = 1;
  if (m > low) {
    const short2 gr = grad[base + (size_t)y * w + x];
    const int xs = gr.x, ys = gr.y;
    const int ax = abs(xs), ay = abs(ys) << 15;
    const int tg22x = ax * 13573;                      // tan(22.5 deg) * 2^15, rounded
    bool peak;
    if (ay < tg22x) peak = m > M(y, x - 1) && m >= M(y, x + 1);                      // gradient ~horizontal
    else {
      const int tg67x = tg22x + (ax << 16);
      if (ay > tg67x) peak = m > M(y - 1, x) && m >= M(y + 1, x);                    // ~vertical
      else {
        const int s = (xs ^ ys) < 0 ? -1 : 1;                                        // diagonal the gradient points along
        peak = m > M(y - 1, x - s) && m > M(y + 1, x + s);
      }
    }
    if (peak) out = m > high ? 2 : 0;
  }
  map[base + (size_t)y * w + x] = out;
}

// d_bgr [n][h][w][3] -> d_map [n][h][w]; d_blur / d_grad / d_mag: scratch of n*h*w elements each.
void launch_canny_map(const uint8_t* d_bgr, int n, int h, int w, int low, int high, uint8_t* d_blur, void* d_grad, void* d_mag, uint8_t* d_map,
                      hipStream_t s) {
  FE_CHECK(n > 0 && h > 0 && w > 0 && (size_t)h * w < (1ull << 31) && n <= 65535, "canny: bad shape");
  hipLaunchKernelGGL(lines_blur_kernel, dim3((w + LB_W - 1) / LB_W, (h + LB_H - 1) / LB_H, n), dim3(LB_W, LB_H), 0, s, d_bgr, h, w, d_blur);
  FE_HIP(hipGetLastError());
  const dim3 blk(64, 4), grid((w + 63) / 64, (h + 3) / 4, n);
  hipLaunchKernelGGL(lines_sobel_kernel, grid, blk, 0, s, d_blur, h, w, (short2*)d_grad, (unsigned short*)d_mag);
  FE_HIP(hipGetLastError());
  hipLaunchKernelGGL(lines_nms_kernel, grid, blk, 0, s, (const short2*)d_grad, (const unsigned short*)d_mag, h, w, low, high, d_map);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

// Leading-line detection (reference analyzers/composition.py:191-261: cv2.GaussianBlur(gray, (5,5), 0) ->
// cv2.Canny(50, 150) -> cv2.HoughLinesP(1, pi/180, 80, minLineLength, maxLineGap=20)).
//
// The pixel scans run on the GPU over the resident BGR batch (three small HBM-bound kernels, ~10 bytes of traffic per pixel):
//   lines_blur_kernel    BGR -> gray (OpenCV's 15-bit weights) -> 5x5 binomial blur, the fixed-point path OpenCV takes for
//                        8-bit images: kernel [1 4 6 4 1]/16 per axis, (sum + 128) >> 8, BORDER_REFLECT_101
//   lines_sobel_kernel   3x3 Sobel dx, dy (BORDER_REPLICATE, as cv::Canny asks for) and the L1 magnitude |dx| + |dy|
//   lines_nms_kernel     non-maximum suppression along the quantised gradient direction (tan 22.5 in 15-bit fixed point) and
//                        the two thresholds -> map: 2 = edge for sure, 0 = edge if connected to a 2, 1 = not an edge
// The rest is inherently sequential and runs on the host, one image per thread: hysteresis (flood fill from the 2s) and the
// progressive probabilistic Hough transform, whose result depends on the order a fixed pseudo-random generator visits the
// edge points in (vote, extract a segment as soon as a bin reaches the threshold, erase its points and un-vote them).
// [DEP-KNOWLEDGE] OpenCV is not available offline: these restate its documented algorithms; parity with cv2 is unpinned.
#include <cmath>
#include <thread>

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

// ---- host: hysteresis -------------------------------------------------------------------------------------------------------
// map (2 / 0 / 1 as above) -> edges in place: 255 where a pixel is a 2 or an 8-connected chain of 0s reaches a 2, else 0.
void canny_hysteresis(uint8_t* map, int h, int w) {
  std::vector<int> stack;
  const size_t npx = (size_t)h * w;
  for (size_t i = 0; i < npx; ++i)
    if (map[i] == 2) stack.push_back((int)i);
  while (!stack.empty()) {
    const int i = stack.back();
    stack.pop_back();
    const int y = i / w, x = i - y * w;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if ((dy | dx) == 0 || yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
        uint8_t& m = map[(size_t)yy * w + xx];
        if (m == 0) { m = 2; stack.push_back(yy * w + xx); }
      }
  }
  for (size_t i = 0; i < npx; ++i) map[i] = map[i] == 2 ? 255 : 0;
}

// ---- host: progressive probabilistic Hough transform (Matas et al.), rho = 1 pixel, theta = 1 degree ----------------------------
namespace {
struct MwcRng {               // OpenCV's generator: multiply-with-carry, seeded with all ones
  uint64_t state = ~0ull;
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
inline int round_half_even(float v) { return (int)std::nearbyintf(v); }   // default rounding mode = to nearest even
// rho bin of pixel (x, y) for angle n: two rounded float products and one rounded float sum (never a fused multiply-add)
inline int rho_bin(const float* trig, int n, int x, int y, int half) {
#pragma clang fp contract(off)
  const float a = (float)x * trig[2 * n];
  const float b = (float)y * trig[2 * n + 1];
  return round_half_even(a + b) + half;
}
}  // namespace

// edges [h][w] (non-zero = edge). Writes the first max_lines segments (x1,y1,x2,y2) to `lines` and returns the number found,
// which may be larger (the caller re-runs with more room).
int hough_lines_p(const uint8_t* edges, int h, int w, int threshold, int min_len, int max_gap, int max_lines, int* lines) {
  const float theta = (float)(M_PI / 180.0);
  const int numangle = 180;
  const int numrho = (int)std::nearbyint((double)((w + h) * 2 + 1));
  std::vector<int> accum((size_t)numangle * numrho, 0);
  std::vector<uint8_t> mask((size_t)h * w);
  std::vector<float> trig(numangle * 2);
  for (int n = 0; n < numangle; ++n) {
    trig[2 * n] = (float)std::cos((double)n * theta);
    trig[2 * n + 1] = (float)std::sin((double)n * theta);
  }
  std::vector<int> nz;        // y * w + x
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const bool on = edges[(size_t)y * w + x] != 0;
      mask[(size_t)y * w + x] = on;
      if (on) nz.push_back(y * w + x);
    }
  MwcRng rng;
  int found = 0;
  const int shift = 16, half = (numrho - 1) / 2;
  auto rho_of = [&](int n, int x, int y) { return rho_bin(trig.data(), n, x, y, half); };
  for (int count = (int)nz.size(); count > 0; --count) {
    const int idx = rng.uniform(0, count);
    const int pt = nz[idx];
    nz[idx] = nz[count - 1];
    const int i = pt / w, j = pt - i * w;
    if (!mask[pt]) continue;                                 // already part of an extracted segment
    int max_val = threshold - 1, max_n = 0;
    for (int n = 0; n < numangle; ++n) {
      const int val = ++accum[(size_t)n * numrho + rho_of(n, j, i)];
      if (max_val < val) { max_val = val; max_n = n; }
    }
    if (max_val < threshold) continue;
    // walk from the point in both directions along the winning line (16.16 fixed point on the minor axis)
    const float a = -trig[2 * max_n + 1], b = trig[2 * max_n];
    int x0 = j, y0 = i, dx0, dy0;
    bool xflag;
    if (std::fabs(a) > std::fabs(b)) {
      xflag = true;
      dx0 = a > 0 ? 1 : -1;
      dy0 = round_half_even(b * (float)(1 << shift) / std::fabs(a));
      y0 = (y0 << shift) + (1 << (shift - 1));
    } else {
      xflag = false;
      dy0 = b > 0 ? 1 : -1;
      dx0 = round_half_even(a * (float)(1 << shift) / std::fabs(b));
      x0 = (x0 << shift) + (1 << (shift - 1));
    }
    int end_x[2] = {j, j}, end_y[2] = {i, i};
    for (int k = 0; k < 2; ++k) {
      int gap = 0, x = x0, y = y0;
      const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
      for (;; x += dx, y += dy) {
        const int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
        if (j1 < 0 || j1 >= w || i1 < 0 || i1 >= h) break;
        if (mask[(size_t)i1 * w + j1]) { gap = 0; end_y[k] = i1; end_x[k] = j1; }
        else if (++gap > max_gap) break;
      }
    }
    const bool good = std::abs(end_x[1] - end_x[0]) >= min_len || std::abs(end_y[1] - end_y[0]) >= min_len;
    for (int k = 0; k < 2; ++k) {                            // erase the segment's points; un-vote them if the segment is kept
      int x = x0, y = y0;
      const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
      for (;; x += dx, y += dy) {
        const int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
        uint8_t& m = mask[(size_t)i1 * w + j1];
        if (m) {
          if (good)
            for (int n = 0; n < numangle; ++n) --accum[(size_t)n * numrho + rho_of(n, j1, i1)];
          m = 0;
        }
        if (i1 == end_y[k] && j1 == end_x[k]) break;
      }
    }
    if (good) {
      if (found < max_lines) {
        int* l = lines + (size_t)found * 4;
        l[0] = end_x[0]; l[1] = end_y[0]; l[2] = end_x[1]; l[3] = end_y[1];
      }
      ++found;
    }
  }
  return found;
}

// maps [n][h][w] (NMS output, host) -> edges in place and, when lines != nullptr, the segments of every image; one image per
// host thread (at most `threads`).
void lines_host_stage(uint8_t* maps, int n, int h, int w, int threshold, int min_len, int max_gap, int max_lines, int* lines, int* counts, int threads) {
  const size_t npx = (size_t)h * w;
  auto work = [&](int first, int step) {
    for (int i = first; i < n; i += step) {
      canny_hysteresis(maps + (size_t)i * npx, h, w);
      if (lines) counts[i] = hough_lines_p(maps + (size_t)i * npx, h, w, threshold, min_len, max_gap, max_lines, lines + (size_t)i * max_lines * 4);
    }
  };
  const int T = std::max(1, std::min(n, threads));
  if (T == 1) { work(0, 1); return; }
  std::atomic<bool> failed{false};
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t] {
      try { work(t, T); } catch (...) { failed = true; }
    });
  for (auto& th : pool) th.join();
  FE_CHECK(!failed, "leading lines: a host worker failed (out of memory?)");
}

}  // namespace fe

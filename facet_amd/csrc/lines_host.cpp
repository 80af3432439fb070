// Host stage of leading-line detection (see kernels_lines.hip for the GPU stage and the reference citations): Canny hysteresis
// and the progressive probabilistic Hough transform. Plain C++ with no HIP dependency, so the same file is also built with
// -fsanitize=address,undefined and fuzzed on the CPU (tests/test_lines_host.py).
#include "lines_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <thread>
#include <vector>

namespace fe {

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
inline int round_half_even(float v) { return (int)lrintf(v); }   // default rounding mode = to nearest even; one cvtss2si on x86-64
// rho bins of pixel (x, y): two rounded float products and one rounded float sum per angle (never a fused multiply-add);
// the 180 bins of one pixel at once (a straight loop the compiler turns into cvtps2dq lanes)
inline void rho_bins(const float* cos_t, const float* sin_t, int numangle, int x, int y, int half, int* out) {
#pragma clang fp contract(off)
  const float fx = (float)x, fy = (float)y;
  for (int n = 0; n < numangle; ++n) out[n] = (int)lrintf(fx * cos_t[n] + fy * sin_t[n]) + half;
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
  std::vector<float> cos_t(numangle), sin_t(numangle);      // planar copies for the vector loop
  for (int n = 0; n < numangle; ++n) { cos_t[n] = trig[2 * n]; sin_t[n] = trig[2 * n + 1]; }
  std::vector<int> bins(numangle);
  for (int count = (int)nz.size(); count > 0; --count) {
    const int idx = rng.uniform(0, count);
    const int pt = nz[idx];
    nz[idx] = nz[count - 1];
    const int i = pt / w, j = pt - i * w;
    if (!mask[pt]) continue;                                 // already part of an extracted segment
    int max_val = threshold - 1, max_n = 0;
    rho_bins(cos_t.data(), sin_t.data(), numangle, j, i, half, bins.data());
    for (int n = 0; n < numangle; ++n) {
      const int val = ++accum[(size_t)n * numrho + bins[n]];
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
          if (good) {
            rho_bins(cos_t.data(), sin_t.data(), numangle, j1, i1, half, bins.data());
            for (int n = 0; n < numangle; ++n) --accum[(size_t)n * numrho + bins[n]];
          }
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
  if (failed) throw std::runtime_error("leading lines: a host worker failed (out of memory?)");
}

}  // namespace fe

// Per-image technical statistics: the OpenCV/numpy scans that reference analyzers/image_cache.py:28-33 and
// analyzers/technical.py:39-342 run per image on the CPU main thread, restated as two HBM-bound passes over the resident batch.
//   pass 1 (one block per image and hue half): BGR -> gray (cv2 COLOR_BGR2GRAY, 15-bit fixed point) written out for pass 2,
//          256-bin gray histogram, BGR -> HSV (cv2 COLOR_BGR2HSV 8-bit: sdiv / hdiv180 tables, 12-bit fixed point), sum of S,
//          and the 180x256 hue-saturation histogram kept ENTIRELY in LDS (2 x 92 KB halves of the 160 KB/CU) so the entropy
//          term sum(c*log2 c) leaves the block as one number - the 46 080-bin histogram never touches HBM.
//   pass 2 (row strips): 4-neighbour Laplacian (cv2.Laplacian ksize=1, BORDER_REFLECT_101) sum and sum of squares, and
//          sum |Immerkaer 3x3| (cv2.filter2D, same border) - all exact integers in 64-bit accumulators.
// Everything derived (variance, percentiles, spread, scores) is host arithmetic on these few numbers (facet_amd/image_stats.py).
// [DEP-KNOWLEDGE: OpenCV color_rgb / color_hsv fixed-point definitions; cv2 is not importable offline -> parity unpinned.]
#include "fe_common.h"

namespace fe {

// record layout (doubles) per image, mirrored in include/facet_engine.h
enum { ST_HIST = 0, ST_LAP_SUM = 256, ST_LAP_SUMSQ = 257, ST_NOISE_ABS = 258, ST_SAT_SUM = 259, ST_HS_CLOG2C = 260, ST_COUNT = 264 };

struct StatsAccum {            // device accumulators per image
  unsigned int hist[256];
  long long lap_sum, lap_sumsq, noise_abs, sat_sum;
  double hs_clog2c[2];         // one term per half of the hue range (block of pass 1), each summed in a fixed order
};

__device__ __forceinline__ int gray_of(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }

__global__ __launch_bounds__(1024) void stats_pass1_kernel(const uint8_t* __restrict__ bgr, int h, int w, uint8_t* __restrict__ gray,
                                                           uint8_t* __restrict__ hsv_out, const int* __restrict__ sdiv,
                                                           const int* __restrict__ hdiv, StatsAccum* __restrict__ acc) {
  extern __shared__ unsigned int lds[];          // [90*256] hue-sat counts of this half, [256] gray histogram, 2 x [256] tables
  const int half = blockIdx.x, img = blockIdx.y;
  unsigned int* hs = lds;
  unsigned int* gh = lds + 90 * 256;
  int* sdiv_l = reinterpret_cast<int*>(lds + 90 * 256 + 256);   // the two division tables, copied next to the histograms:
  int* hdiv_l = sdiv_l + 256;                                   // per-pixel lookups are scattered, LDS serves them cheaply
  for (int i = threadIdx.x; i < 90 * 256 + 256; i += blockDim.x) lds[i] = 0;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) { sdiv_l[i] = sdiv[i]; hdiv_l[i] = hdiv[i]; }
  __syncthreads();
  const size_t npx = (size_t)h * w;
  const uint8_t* src = bgr + (size_t)img * npx * 3;
  long long sat = 0;
  // gray / saturation / plane output are shared between the two blocks of an image by loop-step parity, so both blocks carry
  // the same number of LDS atomics
  auto one_pixel = [&](bool do_gray, int b, int g, int r, int* gray_v, int* h_v, int* s_v, int* v_v) {
    int v = b > g ? b : g; v = v > r ? v : r;
    int vmin = b < g ? b : g; vmin = vmin < r ? vmin : r;
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * sdiv_l[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv_l[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    hh = hh > 255 ? 255 : hh;
    *h_v = hh; *s_v = s; *v_v = v;
    if (do_gray) {
      const int y = gray_of(b, g, r);
      *gray_v = y;
      atomicAdd(&gh[y], 1u);
      sat += s;
    }
    const int hl = hh - half * 90;
    const bool mine = hl >= 0 && hl < 90;
    const int bin = mine ? hl * 256 + s : -1;
    // flat regions put whole wavefronts into one bin: one add for all lanes that share the first lane's bin
    const int lead = __builtin_amdgcn_readfirstlane(bin);
    const unsigned long long same = __ballot(bin == lead);
    if (bin == lead) {
      if (bin >= 0 && (__ffsll((long long)same) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&hs[bin], (unsigned)__popcll(same));
    } else if (bin >= 0) {
      atomicAdd(&hs[bin], 1u);
    }
  };
  if ((npx & 3) == 0) {
    // 4 pixels = 12 bytes = 3 dwords per thread and step (image bases stay 4-byte aligned because npx % 4 == 0); the next
    // step's dwords are requested before this step's pixels are processed
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    const size_t groups = npx >> 2;
    size_t q = threadIdx.x;
    uint32_t d0 = 0, d1 = 0, d2 = 0;
    if (q < groups) { d0 = s32[q * 3]; d1 = s32[q * 3 + 1]; d2 = s32[q * 3 + 2]; }
    while (q < groups) {
      const size_t qn = q + blockDim.x;
      uint32_t e0 = 0, e1 = 0, e2 = 0;
      if (qn < groups) { e0 = s32[qn * 3]; e1 = s32[qn * 3 + 1]; e2 = s32[qn * 3 + 2]; }
      const int px[12] = {(int)(d0 & 255), (int)((d0 >> 8) & 255), (int)((d0 >> 16) & 255), (int)(d0 >> 24),
                          (int)(d1 & 255), (int)((d1 >> 8) & 255), (int)((d1 >> 16) & 255), (int)(d1 >> 24),
                          (int)(d2 & 255), (int)((d2 >> 8) & 255), (int)((d2 >> 16) & 255), (int)(d2 >> 24)};
      int gy[4] = {0, 0, 0, 0}, hv[4], sv[4], vv[4];
      const bool mine_g = (int)((q / blockDim.x) & 1) == half;   // block-uniform per step
#pragma unroll
      for (int k = 0; k < 4; ++k) one_pixel(mine_g, px[3 * k], px[3 * k + 1], px[3 * k + 2], &gy[k], &hv[k], &sv[k], &vv[k]);
      if (mine_g) {
        reinterpret_cast<uint32_t*>(gray + (size_t)img * npx)[q] = (uint32_t)gy[0] | ((uint32_t)gy[1] << 8) | ((uint32_t)gy[2] << 16) | ((uint32_t)gy[3] << 24);
        if (hsv_out) {
          uint32_t* o = reinterpret_cast<uint32_t*>(hsv_out + (size_t)img * npx * 3) + q * 3;
          o[0] = (uint32_t)hv[0] | ((uint32_t)sv[0] << 8) | ((uint32_t)vv[0] << 16) | ((uint32_t)hv[1] << 24);
          o[1] = (uint32_t)sv[1] | ((uint32_t)vv[1] << 8) | ((uint32_t)hv[2] << 16) | ((uint32_t)sv[2] << 24);
          o[2] = (uint32_t)vv[2] | ((uint32_t)hv[3] << 8) | ((uint32_t)sv[3] << 16) | ((uint32_t)vv[3] << 24);
        }
      }
      d0 = e0; d1 = e1; d2 = e2;
      q = qn;
    }
  } else {
    for (size_t p = threadIdx.x; p < npx; p += blockDim.x) {
      int gy = 0, hv, sv, vv;
      const bool mine_g = (int)((p / blockDim.x) & 1) == half;
      one_pixel(mine_g, src[p * 3], src[p * 3 + 1], src[p * 3 + 2], &gy, &hv, &sv, &vv);
      if (mine_g) {
        gray[(size_t)img * npx + p] = (uint8_t)gy;
        if (hsv_out) { uint8_t* o = hsv_out + ((size_t)img * npx + p) * 3; o[0] = (uint8_t)hv; o[1] = (uint8_t)sv; o[2] = (uint8_t)vv; }
      }
    }
  }
  __syncthreads();
  // entropy term of this half: sum c*log2(c); wave-reduced, one atomic per wave
  double part = 0.0;
  for (int i = threadIdx.x; i < 90 * 256; i += blockDim.x) {
    const unsigned c = hs[i];
    if (c) part += (double)c * log2((double)c);
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  // the wave terms are added in wave order by one thread: a floating-point atomic per wave made the last bit of the entropy depend on
  // which wave came first (two runs of the same image could differ by one ulp)
  double* wave_part = reinterpret_cast<double*>(sdiv_l);          // the division tables are dead after the pixel loop
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot += wave_part[wv];
    acc[img].hs_clog2c[half] = tot;
  }
  for (int o = 32; o > 0; o >>= 1) sat += __shfl_xor(sat, o);
  if ((threadIdx.x & 63) == 0 && sat) atomicAdd((unsigned long long*)&acc[img].sat_sum, (unsigned long long)sat);
  for (int i = threadIdx.x; i < 256; i += blockDim.x)
    if (gh[i]) atomicAdd(&acc[img].hist[i], gh[i]);
}

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }   // BORDER_REFLECT_101

__global__ __launch_bounds__(256) void stats_pass2_kernel(const uint8_t* __restrict__ gray, int h, int w, int rows_per_block,
                                                          StatsAccum* __restrict__ acc) {
  const int img = blockIdx.y;
  const uint8_t* g = gray + (size_t)img * h * w;
  const int y0 = blockIdx.x * rows_per_block, y1 = min(h, y0 + rows_per_block);
  long long lsum = 0, lsq = 0, nabs = 0;
  for (int y = y0; y < y1; ++y) {
    const uint8_t* rm = g + (size_t)(h > 1 ? refl(y - 1, h) : 0) * w;
    const uint8_t* rc = g + (size_t)y * w;
    const uint8_t* rp = g + (size_t)(h > 1 ? refl(y + 1, h) : 0) * w;
    for (int x = threadIdx.x; x < w; x += blockDim.x) {
      const int xm = w > 1 ? refl(x - 1, w) : 0, xp = w > 1 ? refl(x + 1, w) : 0;
      const int c = rc[x];
      const int cross = rm[x] + rp[x] + rc[xm] + rc[xp];
      const int diag = rm[xm] + rm[xp] + rp[xm] + rp[xp];
      const int lap = cross - 4 * c;
      const int imm = diag - 2 * cross + 4 * c;            // [[1,-2,1],[-2,4,-2],[1,-2,1]]
      lsum += lap;
      lsq += lap * lap;
      nabs += imm < 0 ? -imm : imm;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { lsum += __shfl_xor(lsum, o); lsq += __shfl_xor(lsq, o); nabs += __shfl_xor(nabs, o); }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd((unsigned long long*)&acc[img].lap_sum, (unsigned long long)lsum);
    atomicAdd((unsigned long long*)&acc[img].lap_sumsq, (unsigned long long)lsq);
    atomicAdd((unsigned long long*)&acc[img].noise_abs, (unsigned long long)nabs);
  }
}

__global__ void stats_pack_kernel(const StatsAccum* __restrict__ acc, int n, double* __restrict__ out) {
  const int img = blockIdx.x;
  if (img >= n) return;
  double* o = out + (size_t)img * ST_COUNT;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) o[ST_HIST + i] = (double)acc[img].hist[i];
  if (threadIdx.x == 0) {
    o[ST_LAP_SUM] = (double)acc[img].lap_sum;
    o[ST_LAP_SUMSQ] = (double)acc[img].lap_sumsq;
    o[ST_NOISE_ABS] = (double)acc[img].noise_abs;
    o[ST_SAT_SUM] = (double)acc[img].sat_sum;
    o[ST_HS_CLOG2C] = acc[img].hs_clog2c[0] + acc[img].hs_clog2c[1];
    o[261] = o[262] = o[263] = 0.0;
  }
}

// ---- Laplacian statistics of rectangular ROIs (face post-processing) ----------------------------------------
// Reference analyzers/face.py:160-176 (eye ROIs) and :272-279 (face crop): cv2.cvtColor(roi, BGR2GRAY) then
// cv2.Laplacian(gray, CV_64F).var() and np.mean(gray) on a numpy slice - the border rule (reflect-101) applies at the ROI
// edge, not the image edge. One block per ROI; gray is recomputed from BGR on the fly (ROIs are small).
// out per ROI: [0] sum lap, [1] sum lap^2, [2] sum gray, [3] pixel count (exact integers in doubles).
__global__ __launch_bounds__(256) void roi_laplacian_kernel(const uint8_t* __restrict__ bgr, int h, int w, const int* __restrict__ img_of,
                                                            const int* __restrict__ rois, int m, double* __restrict__ out) {
  const int f = blockIdx.x;
  if (f >= m) return;
  const int x1 = rois[4 * f], y1 = rois[4 * f + 1], x2 = rois[4 * f + 2], y2 = rois[4 * f + 3];
  const int rw = x2 - x1, rh = y2 - y1;
  __shared__ long long red[3][4];
  long long lsum = 0, lsq = 0, gsum = 0;
  if (rw > 0 && rh > 0) {
    const uint8_t* src = bgr + (size_t)img_of[f] * h * w * 3;
    auto G = [&](int yy, int xx) {
      const uint8_t* p = src + ((size_t)(y1 + yy) * w + (x1 + xx)) * 3;
      return gray_of(p[0], p[1], p[2]);
    };
    const int total = rw * rh;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int y = i / rw, x = i - y * rw;
      const int ym = rh > 1 ? (y - 1 < 0 ? 1 : y - 1) : 0, yp = rh > 1 ? (y + 1 >= rh ? rh - 2 : y + 1) : 0;
      const int xm = rw > 1 ? (x - 1 < 0 ? 1 : x - 1) : 0, xp = rw > 1 ? (x + 1 >= rw ? rw - 2 : x + 1) : 0;
      const int c = G(y, x);
      const int lap = G(ym, x) + G(yp, x) + G(y, xm) + G(y, xp) - 4 * c;
      lsum += lap; lsq += lap * lap; gsum += c;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { lsum += __shfl_xor(lsum, o); lsq += __shfl_xor(lsq, o); gsum += __shfl_xor(gsum, o); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = lsum; red[1][wv] = lsq; red[2][wv] = gsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = out + (size_t)f * 4;
    o[0] = (double)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    o[1] = (double)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    o[2] = (double)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    o[3] = (rw > 0 && rh > 0) ? (double)rw * rh : 0.0;
  }
}
void launch_roi_laplacian(const uint8_t* bgr, int h, int w, const int* img_of, const int* rois, int m, double* out, hipStream_t s) {
  if (m <= 0) return;
  hipLaunchKernelGGL(roi_laplacian_kernel, dim3(m), dim3(256), 0, s, bgr, h, w, img_of, rois, m, out);
  FE_HIP(hipGetLastError());
}

size_t stats_accum_bytes(int n) { return (size_t)n * sizeof(StatsAccum); }

// gray: device [n][h][w] scratch (always written); hsv_out: nullable device [n][h][w][3]; out: device [n][ST_COUNT] doubles
void launch_image_stats(const uint8_t* bgr, int n, int h, int w, uint8_t* gray, uint8_t* hsv_out, const int* sdiv, const int* hdiv, void* accum,
                        double* out, hipStream_t s) {
  FE_CHECK(n > 0 && h > 0 && w > 0 && (size_t)h * w < (1ull << 31), "image_stats: bad shape");
  StatsAccum* acc = (StatsAccum*)accum;
  FE_HIP(hipMemsetAsync(acc, 0, stats_accum_bytes(n), s));
  constexpr size_t lds = (size_t)(90 * 256 + 256 + 512) * sizeof(unsigned int);
  static std::atomic<uint64_t> lds_set{0};
  ensure_dynamic_lds((const void*)stats_pass1_kernel, lds, lds_set);
  hipLaunchKernelGGL(stats_pass1_kernel, dim3(2, n), dim3(1024), lds, s, bgr, h, w, gray, hsv_out, sdiv, hdiv, acc);
  FE_HIP(hipGetLastError());
  const int rpb = 16;
  hipLaunchKernelGGL(stats_pass2_kernel, dim3((h + rpb - 1) / rpb, n), dim3(256), 0, s, gray, h, w, rpb, acc);
  FE_HIP(hipGetLastError());
  hipLaunchKernelGGL(stats_pack_kernel, dim3(n), dim3(256), 0, s, acc, n, out);
  FE_HIP(hipGetLastError());
}

// cv2's 8-bit HSV division tables: sdiv[i] = round((255 << 12) / i), hdiv180[i] = round((180 << 12) / (6 i)), [0] = 0
void cv_hsv_tables(std::vector<int>& sdiv, std::vector<int>& hdiv) {
  sdiv.assign(256, 0);
  hdiv.assign(256, 0);
  for (int i = 1; i < 256; ++i) {
    sdiv[i] = (int)lrint((double)(255 << 12) / (1.0 * i));
    hdiv[i] = (int)lrint((double)(180 << 12) / (6.0 * i));
  }
}

}  // namespace fe

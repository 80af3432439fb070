// Face-path pixel kernels: the OpenCV steps insightface performs around its three networks, restated on the GPU so a whole
// batch (and all faces of a batch) moves through them without leaving HBM. [DEP-KNOWLEDGE: OpenCV imgproc / insightface
// model_zoo; neither is importable offline, so parity with cv2 itself is UNPINNED - the numpy oracle in oracle/face_ref.py
// restates the same fixed-point arithmetic and the GPU results are bit-exact against it.]
//   cv_resize_linear_u8   cv2.resize(img, (nw, nh)) INTER_LINEAR, 8-bit: 11-bit coefficients, horizontal pass to int,
//                         vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2; exact 2x downscale = 2x2 box average.
//                         (SCRFD.detect: aspect-preserving resize into the top-left of a zero det_size canvas)
//   warp_affine_u8        cv2.warpAffine(img, M, (S,S), borderValue=0) INTER_LINEAR: 10-bit fixed-point source coordinates,
//                         5-bit sub-pixel, 15-bit bilinear weights (face_align.norm_crop / face_align.transform)
//   u8_blob               cv2.dnn.blobFromImage(s): (pixel - mean) * scale, optional R<->B swap, to NHWC4 fp32
//   scrfd_decode          score threshold + distance2bbox / distance2kps + compaction into per-image candidate lists
#include "fe_common.h"

namespace fe {

static inline int grid_for_f(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- cv2.resize INTER_LINEAR (u8, 3 channels) into a canvas --------------------------------------------------------
// xofs/yofs: source index per destination column/row; ialpha/ibeta: the two 11-bit weights (short) per column/row.
__global__ void cv_resize_linear_kernel(const uint8_t* __restrict__ src, int n, int h, int w, uint8_t* __restrict__ dst, int ch, int cw,
                                        int nh, int nw, const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                        const int* __restrict__ yofs, const short* __restrict__ ibeta, int area2) {
  const size_t total = (size_t)n * nh * nw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int dx = (int)(i % nw);
    const int dy = (int)((i / nw) % nh);
    const int img = (int)(i / ((size_t)nw * nh));
    const uint8_t* s = src + (size_t)img * h * w * 3;
    uint8_t* d = dst + (((size_t)img * ch + dy) * cw + dx) * 3;
    if (area2) {   // scale exactly 2 in both directions: OpenCV switches INTER_LINEAR to the 2x2 box average
      const uint8_t* p = s + ((size_t)(2 * dy) * w + 2 * dx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c] = (uint8_t)((p[c] + p[3 + c] + p[(size_t)w * 3 + c] + p[(size_t)w * 3 + 3 + c] + 2) >> 2);
      continue;
    }
    const int sx = xofs[dx], sy = yofs[dy];
    const int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1], b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    // columns: the table already folds the border rule (weight 2048 on the last column); rows: OpenCV clamps each of the
    // two source rows into the image separately
    const int sx1 = sx + 1 < w ? sx + 1 : sx;
    const int sy0 = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy), sy1 = sy + 1 < 0 ? 0 : (sy + 1 > h - 1 ? h - 1 : sy + 1);
    const uint8_t* r0 = s + (size_t)sy0 * w * 3;
    const uint8_t* r1 = s + (size_t)sy1 * w * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
      const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
      const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}
void launch_cv_resize_linear(const uint8_t* src, int n, int h, int w, uint8_t* dst, int ch, int cw, int nh, int nw, const int* xofs,
                             const short* ialpha, const int* yofs, const short* ibeta, int area2, hipStream_t s) {
  FE_CHECK(nh <= ch && nw <= cw && nh > 0 && nw > 0, "cv_resize: target %dx%d does not fit the %dx%d canvas", nh, nw, ch, cw);
  hipLaunchKernelGGL(cv_resize_linear_kernel, dim3(grid_for_f((size_t)n * nh * nw)), dim3(256), 0, s, src, n, h, w, dst, ch, cw, nh, nw,
                     xofs, ialpha, yofs, ibeta, area2);
  FE_HIP(hipGetLastError());
}

// ---- cv2.warpAffine INTER_LINEAR, BORDER_CONSTANT(0) -------------------------------------------------------------------
// Minv: per crop the inverted 2x3 matrix (dst -> src), doubles, as OpenCV computes it before the fixed-point walk.
// wtab: the 32x32x4 table of 15-bit bilinear weights (host-built, sums forced to 32768 like initInterTab2D).
__global__ void warp_affine_kernel(const uint8_t* __restrict__ src, int h, int w, const int* __restrict__ img_of, const double* __restrict__ Minv,
                                   int m, int S, const short* __restrict__ wtab, uint8_t* __restrict__ dst) {
  const size_t total = (size_t)m * S * S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % S);
    const int y = (int)((i / S) % S);
    const int f = (int)(i / ((size_t)S * S));
    const double* M = Minv + (size_t)f * 6;
    const int adelta = __double2int_rn(M[0] * x * 1024.0), bdelta = __double2int_rn(M[3] * x * 1024.0);
    const int X0 = __double2int_rn((M[1] * y + M[2]) * 1024.0) + 16, Y0 = __double2int_rn((M[4] * y + M[5]) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const short* wt = wtab + (size_t)((Y & 31) * 32 + (X & 31)) * 4;
    const uint8_t* s = src + (size_t)img_of[f] * h * w * 3;
    uint8_t* d = dst + i * 3;
    const bool x0ok = sx >= 0 && sx < w, x1ok = sx + 1 >= 0 && sx + 1 < w, y0ok = sy >= 0 && sy < h, y1ok = sy + 1 >= 0 && sy + 1 < h;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int p00 = (x0ok && y0ok) ? s[((size_t)sy * w + sx) * 3 + c] : 0;
      const int p01 = (x1ok && y0ok) ? s[((size_t)sy * w + sx + 1) * 3 + c] : 0;
      const int p10 = (x0ok && y1ok) ? s[((size_t)(sy + 1) * w + sx) * 3 + c] : 0;
      const int p11 = (x1ok && y1ok) ? s[((size_t)(sy + 1) * w + sx + 1) * 3 + c] : 0;
      const int v = (p00 * wt[0] + p01 * wt[1] + p10 * wt[2] + p11 * wt[3] + (1 << 14)) >> 15;
      d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}
void launch_warp_affine(const uint8_t* src, int h, int w, const int* img_of, const double* Minv, int m, int S, const short* wtab,
                        uint8_t* dst, hipStream_t s) {
  if (m <= 0) return;
  hipLaunchKernelGGL(warp_affine_kernel, dim3(grid_for_f((size_t)m * S * S)), dim3(256), 0, s, src, h, w, img_of, Minv, m, S, wtab, dst);
  FE_HIP(hipGetLastError());
}

// ---- blobFromImage: (pixel - mean) * scale -> NHWC4 fp32 (4th channel 0) -----------------------------------------------
__global__ void u8_blob_kernel(const uint8_t* __restrict__ src, float4* __restrict__ dst, size_t pixels, float mean, float scale, int swap_rb) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (size_t)gridDim.x * blockDim.x) {
    float c0 = (float)src[i * 3 + 0], c1 = (float)src[i * 3 + 1], c2 = (float)src[i * 3 + 2];
    if (swap_rb) { const float t = c0; c0 = c2; c2 = t; }
    dst[i] = make_float4((c0 - mean) * scale, (c1 - mean) * scale, (c2 - mean) * scale, 0.f);
  }
}
void launch_u8_blob(const uint8_t* src, float* dst, size_t pixels, float mean, float scale, int swap_rb, hipStream_t s) {
  if (!pixels) return;
  hipLaunchKernelGGL(u8_blob_kernel, dim3(grid_for_f(pixels)), dim3(256), 0, s, src, (float4*)dst, pixels, mean, scale, swap_rb);
  FE_HIP(hipGetLastError());
}

// ---- SCRFD decode: one stride level of a batch --------------------------------------------------------------------------
// scores [n*hw*A], bbox [n*hw*A][4], kps [n*hw*A][2*K] (distance units of one stride). Candidate record (16 floats):
// score, x1, y1, x2, y2, kps x,y * 5, level index. Coordinates are divided by det_scale like SCRFD.detect does.
__global__ void scrfd_decode_kernel(const float* __restrict__ scores, const float* __restrict__ bbox, const float* __restrict__ kps, int n,
                                    int fh, int fw, int A, int K, int stride, float thresh, float det_scale, int level,
                                    float* __restrict__ cand, int* __restrict__ counts, int max_cand) {
  const size_t per = (size_t)fh * fw * A;
  const size_t total = (size_t)n * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float sc = scores[i];
    if (!(sc >= thresh)) continue;
    const int img = (int)(i / per);
    const size_t r = i - (size_t)img * per;
    const size_t cell = r / A;
    const float px = (float)((int)(cell % fw) * stride), py = (float)((int)(cell / fw) * stride);
    const int slot = atomicAdd(&counts[img], 1);
    if (slot >= max_cand) continue;
    float* o = cand + ((size_t)img * max_cand + slot) * 16;
    const float* b = bbox + i * 4;
    o[0] = sc;
    const float fs = (float)stride;
    o[1] = (px - b[0] * fs) / det_scale;
    o[2] = (py - b[1] * fs) / det_scale;
    o[3] = (px + b[2] * fs) / det_scale;
    o[4] = (py + b[3] * fs) / det_scale;
    for (int k = 0; k < 5; ++k) {
      o[5 + 2 * k] = k < K ? (px + kps[i * 2 * K + 2 * k] * fs) / det_scale : 0.f;
      o[6 + 2 * k] = k < K ? (py + kps[i * 2 * K + 2 * k + 1] * fs) / det_scale : 0.f;
    }
    o[15] = (float)level;
  }
}
void launch_scrfd_decode(const float* scores, const float* bbox, const float* kps, int n, int fh, int fw, int A, int K, int stride,
                         float thresh, float det_scale, int level, float* cand, int* counts, int max_cand, hipStream_t s) {
  const size_t total = (size_t)n * fh * fw * A;
  if (!total) return;
  hipLaunchKernelGGL(scrfd_decode_kernel, dim3(grid_for_f(total)), dim3(256), 0, s, scores, bbox, kps, n, fh, fw, A, K, stride, thresh,
                     det_scale, level, cand, counts, max_cand);
  FE_HIP(hipGetLastError());
}

// ---- host-side tables ---------------------------------------------------------------------------------------------------
static inline short sat_short_round(float v) {
  const long r = lrintf(v);   // cvRound: round half to even
  return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}
// cv::resize's coefficient precomputation for INTER_LINEAR. `clamp_fx`: the horizontal pass folds the borders into the
// weights (fx = 0 at the first/last source column); the vertical pass keeps its weights and clamps source rows instead.
void cv_resize_tables(int src, int dst, bool clamp_fx, std::vector<int>& ofs, std::vector<short>& coef) {
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  ofs.resize(dst);
  coef.resize((size_t)dst * 2);
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_fx) {
      if (s < 0) { f = 0.f; s = 0; }
      if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    ofs[d] = s;
    coef[d * 2] = sat_short_round((1.f - f) * 2048.f);
    coef[d * 2 + 1] = sat_short_round(f * 2048.f);
  }
}
// initInterTab2D(INTER_LINEAR, fixpt): products of (1-x, x), x = i/32, times 32768, saturated to short; the one entry that
// saturates (weight 1.0 at integer positions -> 32767) gets its missing unit added to the (1,1) tap, as OpenCV's sum fix does.
void cv_warp_weight_table(std::vector<short>& wtab) {
  wtab.assign(32 * 32 * 4, 0);
  for (int fy = 0; fy < 32; ++fy)
    for (int fx = 0; fx < 32; ++fx) {
      const float ty[2] = {1.f - fy / 32.f, fy / 32.f}, tx[2] = {1.f - fx / 32.f, fx / 32.f};
      short* t = &wtab[(size_t)(fy * 32 + fx) * 4];
      int sum = 0;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) { t[a * 2 + b] = sat_short_round(ty[a] * tx[b] * 32768.f); sum += t[a * 2 + b]; }
      if (sum != 32768) t[3] = (short)(t[3] - (sum - 32768));
    }
}

}  // namespace fe

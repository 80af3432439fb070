// Stem convolutions (3(+1 zero) input channels -> 32 or 64 output channels): 7x7 stride 2 = the first layer of ResNet-50 (TOPIQ,
// pyiqa/timm resnet50 conv1) and ResNet-18 (SAMP-Net, reference models/samp_net.py:652-662); 3x3 stride 1 / 2 = the first
// layer of the face graphs (ArcFace IResNet, SCRFD). With Cin = 4 the generic
// implicit-GEMM kernel spends its time gathering 16-byte taps from HBM/L2 (K = 196 -> 59 TFLOP/s at 1024^2); here the input
// patch of an 8x32 output tile (21 x 69 pixels x 16 B = 23 KB) and the whole weight tensor (Cout x 50 taps x 16 B) sit in LDS
// and the MFMA operands are read straight out of them:
//   * one ds_read_b128 = the 4 channels of one tap of one pixel = the A operand of the MFMAs of that tap (k-permutation:
//     lanes 0-31 feed tap 2p, lanes 32-63 tap 2p+1 of a tap pair); the zero 4th channel is skipped -> 3 MFMAs per tap pair,
//     K_eff = 150 for 147 algorithmic,
//   * operands are swapped (weights as A, pixels as B) so every lane ends up holding 4 consecutive output channels of ITS
//     pixel: the epilogue moves float4 groups (no scalar shuffles) through a wave-private LDS transpose to full-row stores.
#include "fe_common.h"
#include <cstdlib>

namespace fe {

typedef float stem_f32x16 __attribute__((ext_vector_type(16)));
typedef float stem_v4f __attribute__((ext_vector_type(4)));

template <class TO>              // TO: output element type (float, or bf16 when the stem feeds the bf16 path)
struct StemParamsT {
  const float* x; int ldx;      // NHWC fp32, >= 4 floats per pixel (channel 3 is ignored)
  const float* w;               // [TAPS2][Cout][4]: tap-major so the 32 lanes of a fragment read consecutive 16-B words; channel 3 and the padding tap are zero
  const float* scale; const float* shift;
  TO* y; int ldy;
  const float* slope;           // PReLU slopes (act == 2)
  int N, H, W, Ho, Wo, Cout, act;   // act: 0 none, 1 relu, 2 prelu
};

constexpr int STEM_TH = 8, STEM_TW = 32;

template <int TN, int STEM_K, int STEM_S, class TO>   // TN = Cout / 32; square kernel STEM_K, stride STEM_S, padding STEM_K / 2
__global__ __launch_bounds__(256, 2) void stem_kernel(StemParamsT<TO> p) {
  constexpr int STEM_P = STEM_K / 2;
  constexpr int STEM_PH = (STEM_TH - 1) * STEM_S + STEM_K;   // 7x7/2: 21 patch rows
  constexpr int STEM_PW = (STEM_TW - 1) * STEM_S + STEM_K;   // 7x7/2: 69 patch columns
  constexpr int STEM_TAPS2 = (STEM_K * STEM_K + 1) & ~1;     // taps padded to an even count (49 -> 50)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                  // [PH][PW][4]
  float* wl = smem + STEM_PH * STEM_PW * 4;             // [TAPS2][TN*32][4]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ox0 = blockIdx.x * STEM_TW, oy0 = blockIdx.y * STEM_TH, img = blockIdx.z;

  // ---- stage the input patch (zero outside the image = the convolution's zero padding) and the weights --------------
  const int iy0 = oy0 * STEM_S - STEM_P, ix0 = ox0 * STEM_S - STEM_P;
  for (int i = t; i < STEM_PH * STEM_PW; i += 256) {
    const int py = i / STEM_PW, px = i - py * STEM_PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      v = *reinterpret_cast<const float4*>(p.x + (((size_t)img * p.H + iy) * p.W + ix) * p.ldx);
    v.w = 0.f;
    *reinterpret_cast<float4*>(patch + (size_t)i * 4) = v;
  }
  for (int i = t; i < TN * 32 * STEM_TAPS2; i += 256)
    *reinterpret_cast<float4*>(wl + (size_t)i * 4) = *reinterpret_cast<const float4*>(p.w + (size_t)i * 4);
  __syncthreads();

  // ---- MFMA over 25 tap pairs; wave w owns output rows 2w, 2w+1 of the tile (two 32-pixel B tiles) ----------------------
  stem_f32x16 acc[TN][2];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
  const float* prow[2] = {patch + ((size_t)((2 * wave + 0) * STEM_S) * STEM_PW + r * STEM_S) * 4,
                          patch + ((size_t)((2 * wave + 1) * STEM_S) * STEM_PW + r * STEM_S) * 4};
#pragma unroll
  for (int pr = 0; pr < STEM_TAPS2 / 2; ++pr) {
    int tap = 2 * pr + h;
    int ky = tap / STEM_K, kx = tap - ky * STEM_K;
    if (tap >= STEM_K * STEM_K) { ky = 0; kx = 0; }      // padding tap: any valid address, its weights are zero
    const int poff = (ky * STEM_PW + kx) * 4;
    stem_v4f a[TN], b[2];
#pragma unroll
    for (int j = 0; j < TN; ++j) a[j] = *reinterpret_cast<const stem_v4f*>(wl + ((size_t)tap * (TN * 32) + j * 32 + r) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const stem_v4f*>(prow[i] + poff);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b[i].x, acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b[i].y, acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b[i].z, acc[j][i], 0, 0, 0);
      }
  }

  // ---- epilogue: D[row = cout][col = pixel]; lane (r, h) holds couts (e&3) + 8(e>>2) + 4h of pixel r. Each wave transposes its
  // 64 pixels x Cout tile through LDS (the patch / weight area is free now) so that 16 (8) lanes write one pixel's whole 256-B
  // (128-B) channel row and a wave stores 1 KiB contiguous runs - the direct form stored 32-byte pieces of 64 different rows.
  __syncthreads();
  constexpr int CO = TN * 32, ES = CO + 4, LPP = CO / 4;         // floats per pixel row in LDS (padded), lanes per pixel
  float* E = smem + (size_t)wave * 64 * ES;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(E + (size_t)(i * 32 + r) * ES + j * 32 + 8 * g + 4 * h) =
            make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
  // wave-private region: the wave's own LDS writes are complete before its reads (same wave, lgkmcnt) - no block barrier needed
  const int c4 = (lane % LPP) * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f), sl = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c4);
  if (p.shift) sf = *reinterpret_cast<const float4*>(p.shift + c4);
  if (p.act == 2) sl = *reinterpret_cast<const float4*>(p.slope + c4);
  constexpr int PPI = 64 / LPP;                                  // pixels per iteration of the wave
#pragma unroll
  for (int it = 0; it < 64 / PPI; ++it) {
    const int px = it * PPI + lane / LPP;                        // 0..63: row i = px / 32 of the wave's two rows, column px % 32
    const int oy = oy0 + 2 * wave + (px >> 5), ox = ox0 + (px & 31);
    float4 v = *reinterpret_cast<const float4*>(E + (size_t)px * ES + c4);
    v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
    if (p.act == 1) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
    else if (p.act == 2) { v.x = v.x > 0.f ? v.x : v.x * sl.x; v.y = v.y > 0.f ? v.y : v.y * sl.y; v.z = v.z > 0.f ? v.z : v.z * sl.z; v.w = v.w > 0.f ? v.w : v.w * sl.w; }
    if (oy < p.Ho && ox < p.Wo) st4(p.y + (((size_t)img * p.Ho + oy) * p.Wo + ox) * p.ldy + c4, v);
  }
}

// The 7x7 stride-2 stem of the bf16 path (the ResNet-50 / ResNet-18 stems of a model committed under bf16): the same tile, patch
// and epilogue, with the products on the bf16 matrix cores. The fp32 NHWC4 input patch and the fp32 weights are rounded to bf16 on
// their way into LDS (the same rounding every later layer of that path applies to its input). A tap row is padded to 8 taps (the
// 8th has zero weights) so taps pair up as (ky, 2t), (ky, 2t+1): one pair = two neighbouring pixels x 4 channels = 16 contiguous,
// 16-byte aligned bytes of the patch = one ds_read_b128 = a lane's 8 k-values of one v_mfma_f32_32x32x16_bf16 (lanes 0-31 feed
// pair 2u, lanes 32-63 pair 2u+1). 28 pairs = 14 MFMAs of 32 cycles per 32x32 tile instead of 75 fp32 MFMAs of 64.
// TO = bf16 | f16: the element type of the model the stem feeds (same kernel, the other matrix instruction).
template <int TN, class TO>
__global__ __launch_bounds__(256, 2) void stem7_bf16_kernel(StemParamsT<TO> p) {
  constexpr int K = 7, S = 2, P = 3;
  constexpr int PH = (STEM_TH - 1) * S + K;            // 21 patch rows
  constexpr int PW = (STEM_TW - 1) * S + K + 1;        // 69 patch columns + 1: rows stay 16-byte aligned at 8 bytes per pixel
  constexpr int PAIRS = K * 4;                         // 28 tap pairs
  constexpr int CO = TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint2* patch = reinterpret_cast<uint2*>(smem);                              // [PH][PW] pixels of 4 bf16
  uint4* wl = reinterpret_cast<uint4*>(smem) + (PH * PW + 1) / 2;             // [PAIRS][CO] entries of 8 bf16 (2 taps x 4 channels)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ox0 = blockIdx.x * STEM_TW, oy0 = blockIdx.y * STEM_TH, img = blockIdx.z;
  auto pack4 = [](const float4 v) -> uint2 {
    return make_uint2(fe_pack2((const TO*)nullptr, v.x, v.y), fe_pack2((const TO*)nullptr, v.z, v.w));
  };
  const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
  for (int i = t; i < PH * PW; i += 256) {
    const int py = i / PW, px = i - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      v = *reinterpret_cast<const float4*>(p.x + (((size_t)img * p.H + iy) * p.W + ix) * p.ldx);
    v.w = 0.f;
    patch[i] = pack4(v);
  }
  for (int i = t; i < PAIRS * CO; i += 256) {          // p.w: [50 taps][CO][4] fp32 (the fp32 stem's layout)
    const int pair = i / CO, co = i - pair * CO;
    const int ky = pair >> 2, kx = (pair & 3) * 2;
    const float4 w0 = *reinterpret_cast<const float4*>(p.w + ((size_t)(ky * K + kx) * CO + co) * 4);
    float4 w1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kx + 1 < K) w1 = *reinterpret_cast<const float4*>(p.w + ((size_t)(ky * K + kx + 1) * CO + co) * 4);
    const uint2 a = pack4(w0), b = pack4(w1);
    wl[i] = make_uint4(a.x, a.y, b.x, b.y);
  }
  __syncthreads();

  stem_f32x16 acc[TN][2];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
  const uint2* prow[2] = {patch + (size_t)((2 * wave + 0) * S) * PW + r * S, patch + (size_t)((2 * wave + 1) * S) * PW + r * S};
#pragma unroll
  for (int u = 0; u < PAIRS / 2; ++u) {
    const int pair = 2 * u + h;
    const int ky = pair >> 2, kx = (pair & 3) * 2;
    union { uint4 u4; fe_v4f f; } a[TN], b[2];
#pragma unroll
    for (int j = 0; j < TN; ++j) a[j].u4 = wl[pair * CO + j * 32 + r];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i].u4 = *reinterpret_cast<const uint4*>(prow[i] + ky * PW + kx);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = fe_mfma16((const TO*)nullptr, a[j].f, b[i].f, acc[j][i]);
  }

  // ---- epilogue: as stem_kernel (lane (r, h) holds couts (e&3) + 8(e>>2) + 4h of pixel r) --------------------------------
  __syncthreads();
  constexpr int ES = CO + 4, LPP = CO / 4;
  float* E = smem + (size_t)wave * 64 * ES;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(E + (size_t)(i * 32 + r) * ES + j * 32 + 8 * g + 4 * h) =
            make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
  const int c4 = (lane % LPP) * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c4);
  if (p.shift) sf = *reinterpret_cast<const float4*>(p.shift + c4);
  constexpr int PPI = 64 / LPP;
#pragma unroll
  for (int it = 0; it < 64 / PPI; ++it) {
    const int px = it * PPI + lane / LPP;
    const int oy = oy0 + 2 * wave + (px >> 5), ox = ox0 + (px & 31);
    float4 v = *reinterpret_cast<const float4*>(E + (size_t)px * ES + c4);
    v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
    if (p.act == 1) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
    if (oy < p.Ho && ox < p.Wo) st4(p.y + (((size_t)img * p.Ho + oy) * p.Wo + ox) * p.ldy + c4, v);
  }
}

template <int TN, class TO>
static void launch_stem7_bf16(const StemParamsT<TO>& p, hipStream_t s) {
  constexpr int PH = (STEM_TH - 1) * 2 + 7, PW = (STEM_TW - 1) * 2 + 7 + 1;
  constexpr size_t main_lds = (size_t)((PH * PW + 1) / 2) * 16 + (size_t)28 * TN * 32 * 16;
  constexpr size_t epi_lds = (size_t)4 * 64 * (TN * 32 + 4) * sizeof(float);
  constexpr size_t lds = main_lds > epi_lds ? main_lds : epi_lds;
  static std::atomic<uint64_t> lds_set{0};
  ensure_dynamic_lds((const void*)stem7_bf16_kernel<TN, TO>, lds, lds_set);
  const dim3 grid((p.Wo + STEM_TW - 1) / STEM_TW, (p.Ho + STEM_TH - 1) / STEM_TH, p.N);
  hipLaunchKernelGGL((stem7_bf16_kernel<TN, TO>), grid, dim3(256), lds, s, p);
  FE_HIP(hipGetLastError());
}

template <int TN, int K, int S, class TO>
static void launch_stem_t(const StemParamsT<TO>& p, hipStream_t s) {
  constexpr int PH = (STEM_TH - 1) * S + K, PW = (STEM_TW - 1) * S + K, T2 = (K * K + 1) & ~1;
  constexpr size_t main_lds = ((size_t)PH * PW * 4 + (size_t)TN * 32 * T2 * 4) * sizeof(float);
  constexpr size_t epi_lds = (size_t)4 * 64 * (TN * 32 + 4) * sizeof(float);     // four wave-private transpose regions
  constexpr size_t lds = main_lds > epi_lds ? main_lds : epi_lds;
  static std::atomic<uint64_t> lds_set{0};
  ensure_dynamic_lds((const void*)stem_kernel<TN, K, S, TO>, lds, lds_set);
  const dim3 grid((p.Wo + STEM_TW - 1) / STEM_TW, (p.Ho + STEM_TH - 1) / STEM_TH, p.N);
  hipLaunchKernelGGL((stem_kernel<TN, K, S, TO>), grid, dim3(256), lds, s, p);
  FE_HIP(hipGetLastError());
}

// wstem: [taps padded to even][Cout][4]; Cout 32 or 64; (k, stride) one of (7,2) (3,1) (3,2), padding k/2; act 0 none / 1 relu /
// 2 prelu(slope). Returns false when the shape is not one of these (the caller falls back to the generic kernel).
template <class TO>
bool launch_stem(const float* x, int ldx, int N, int H, int W, const float* wstem, const float* scale, const float* shift, const float* slope,
                 int Cout, int k, int stride, int act, TO* y, int ldy, int Ho, int Wo, hipStream_t s) {
  const int pad = k / 2;
  if (!(Cout == 32 || Cout == 64) || ldx < 4 || ldx % 4 || ldy % 4 || Ho != (H + 2 * pad - k) / stride + 1 || Wo != (W + 2 * pad - k) / stride + 1 ||
      (((uintptr_t)x | (uintptr_t)wstem) & 15) || ((uintptr_t)y & (4 * sizeof(TO) - 1)) || (act == 2 && !slope) || act < 0 || act > 2)
    return false;
  StemParamsT<TO> p{x, ldx, wstem, scale, shift, y, ldy, slope, N, H, W, Ho, Wo, Cout, act};
  const int key = k * 10 + stride;
  if constexpr (sizeof(TO) == 2) {      // a stem that feeds the bf16 path: 7x7 / 2 on the bf16 matrix cores
    static const bool f32_stem = getenv("FE_BF16_STEM_F32") != nullptr;      // A/B hook: keep the fp32 products
    if (key == 72 && act != 2 && !f32_stem) {
      if (Cout == 64) launch_stem7_bf16<2, TO>(p, s); else launch_stem7_bf16<1, TO>(p, s);
      return true;
    }
  }
  if (Cout == 64) {
    if (key == 72) launch_stem_t<2, 7, 2, TO>(p, s);
    else if (key == 31) launch_stem_t<2, 3, 1, TO>(p, s);
    else if (key == 32) launch_stem_t<2, 3, 2, TO>(p, s);
    else return false;
  } else {
    if (key == 72) launch_stem_t<1, 7, 2, TO>(p, s);
    else if (key == 31) launch_stem_t<1, 3, 1, TO>(p, s);
    else if (key == 32) launch_stem_t<1, 3, 2, TO>(p, s);
    else return false;
  }
  return true;
}
template bool launch_stem<float>(const float*, int, int, int, int, const float*, const float*, const float*, const float*, int, int, int, int, float*, int, int, int,
                                 hipStream_t);
template bool launch_stem<bf16>(const float*, int, int, int, int, const float*, const float*, const float*, const float*, int, int, int, int, bf16*, int, int, int,
                                hipStream_t);
template bool launch_stem<f16>(const float*, int, int, int, int, const float*, const float*, const float*, const float*, int, int, int, int, f16*, int, int, int,
                               hipStream_t);

}  // namespace fe

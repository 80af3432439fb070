// 3x3 convolution with 16 output channels in fp32, stride 1, padding 1 - U2-Net-P's REBNCONV layers with mid_ch = 16 (64 -> 16, 32 -> 16,
// 16 -> 16 at 224^2 and 112^2: reference models/samp_net.py:45-54, 62-255; a quarter of the SAMP stage's convolution time). The generic
// fp32 kernel serves them with its 256 x 32 tile on v_mfma_f32_32x32x2_f32: half of every matrix instruction multiplies zero columns
// (43 TFLOP/s). Here: v_mfma_f32_16x16x4_f32 (the same FLOP rate, 16 columns wide), TRANSPOSED (A = weights: 16 output
// channels x 4 k, B = pixels: 4 k x 16 pixels), halo-tiled like kernels_c64.hip: one persistent workgroup per CU walks 16 x 16 output
// tiles, the 18 x 18 x Cin input patch is loaded once into LDS (prefetched in registers behind the previous tile's matrix loop), the
// nine taps are offset reads of it, and the weights (36 KB at Cin = 64) sit behind it in LDS in operand order. The contraction order inside a block of
// 16 channels is permuted so that one 16-byte LDS read feeds four matrix instructions (lane (pixel, kq) reads channels 4 kq .. 4 kq + 3;
// instruction i of the four contracts {4 kq' + i}); the weights are stored in that order on the host. A lane's 4 results are 4 consecutive
// output channels of one pixel: one 16-byte store, four lanes = the pixel's 64 bytes.
#include "fe_common.h"
#include "engine.h"

#include <atomic>
#include <vector>

namespace fe {

struct N16Params {
  const float* x; int ldx;             // [B][H][W][Cin] (ld >= Cin: a slice of a concat buffer)
  int B, H, W;
  float* y; int ldy;                   // [B][H][W][16] (ld >= 16)
  const float* wt;                     // [tap 9][Cin / 16][lane 64][i 4]
  const float *scale, *shift;          // [16] or null
  int act;
  int tiles_x, tiles_y, ntiles;
};

typedef float n16_f4 __attribute__((ext_vector_type(4)));

template <int CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_n16_f32_kernel(const N16Params p) {
  constexpr int G = CIN / 16, C4 = CIN / 4, PITCH = CIN * 4 + 16, NPIECE = 324 * C4, PER_T = (NPIECE + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_n[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 15, kq = lane >> 4;

  // weights: [tap][g][lane][i] behind the patch in LDS, one 16-byte read per (tap, 16-channel block) and lane (in registers - 9 Cin / 4
  // per lane - the Cin = 64 form spilled beside the 84 registers of the prefetched patch)
  char* const sWt = smem_n + 324 * PITCH;
  for (int i = t; i < 9 * G * 64; i += 256) reinterpret_cast<n16_f4*>(sWt)[i] = reinterpret_cast<const n16_f4*>(p.wt)[i];
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sc[j] = p.scale ? p.scale[4 * kq + j] : 1.f; sh[j] = p.shift ? p.shift[4 * kq + j] : 0.f; }
  // the thread's pieces of a patch: 324 pixels x Cin / 4 float4s
  int pc_dy[PER_T], pc_dx[PER_T], pc_rel[PER_T], pc_dst[PER_T];
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int pc = t + 256 * i, px = pc / C4, c4 = pc - px * C4;
    pc_dy[i] = px / 18 - 1; pc_dx[i] = px % 18 - 1;
    pc_rel[i] = (pc_dy[i] * p.W + pc_dx[i]) * p.ldx + c4 * 4;
    pc_dst[i] = pc < NPIECE ? px * PITCH + c4 * 16 : -1;
  }
  auto tile_origin = [&](const int v, int& b, int& y0, int& x0) {
    const int per = p.tiles_x * p.tiles_y;
    b = v / per;
    const int rem = v - b * per, ty = rem / p.tiles_x;
    y0 = ty * 16; x0 = (rem - ty * p.tiles_x) * 16;
  };
  n16_f4 pf[PER_T];
  auto fetch_patch = [&](const int v) {
    if (v >= p.ntiles) return;
    int b, y0, x0;
    tile_origin(v, b, y0, x0);
    const float* const org = p.x + (((size_t)b * p.H + y0) * p.W + x0) * p.ldx;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const bool in = pc_dst[i] >= 0 && (unsigned)(y0 + pc_dy[i]) < (unsigned)p.H && (unsigned)(x0 + pc_dx[i]) < (unsigned)p.W;
      const n16_f4 q = *reinterpret_cast<const n16_f4*>(org + (in ? pc_rel[i] : 0));
      pf[i] = in ? q : n16_f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < PER_T; ++i)
      if (pc_dst[i] >= 0) *reinterpret_cast<n16_f4*>(smem_n + pc_dst[i]) = pf[i];
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the weight loads have landed before the tile loop (see kernels_gate.hip)
  fetch_patch(blockIdx.x);
  store_patch();
  __syncthreads();

  for (int v = blockIdx.x; v < p.ntiles; v += gridDim.x) {
    int b, y0, x0;
    tile_origin(v, b, y0, x0);
    fetch_patch(v + (int)gridDim.x);
    // wave w: tile rows 4 w .. 4 w + 3, one 16-pixel row each
    n16_f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = n16_f4{0.f, 0.f, 0.f, 0.f};
    const char* const base = smem_n + ((wave * 4) * 18 + n) * PITCH + kq * 16;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int dy = tp / 3, dx = tp - dy * 3;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const n16_f4 aw = *reinterpret_cast<const n16_f4*>(sWt + ((tp * G + g) * 64 + lane) * 16);
        n16_f4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const n16_f4*>(base + ((q + dy) * 18 + dx) * PITCH + g * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[i], bq[q][i], acc[q], 0, 0, 0);
      }
    }
    __syncthreads();      // every wave is done with this patch
    store_patch();
    // lane (pixel n, channel quad kq): results j = output channels 4 kq + j
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = y0 + wave * 4 + q, ox = x0 + n;
      if (oy < p.H && ox < p.W) {
        n16_f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fe_apply_act(acc[q][j] * sc[j] + sh[j], p.act);
        *reinterpret_cast<n16_f4*>(p.y + (((size_t)b * p.H + oy) * p.W + ox) * p.ldy + 4 * kq) = o;
      }
    }
    __syncthreads();      // the next patch is in LDS
  }
}

// Host: the weights [16][Cin][3][3] in the kernel's order [tap][Cin / 16][lane][i]: lane (m = lane % 16, kq = lane / 16) holds
// W[m][16 g + 4 kq + i][tap].
float* build_n16_weights(DeviceWeights& dw, const float* W, int cin) {
  const int G = cin / 16;
  std::vector<float> f((size_t)9 * G * 4 * 64);
  for (int tp = 0; tp < 9; ++tp)
    for (int g = 0; g < G; ++g)
      for (int i = 0; i < 4; ++i)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, kq = lane >> 4, ch = 16 * g + 4 * kq + i;
          f[((size_t)(tp * G + g) * 64 + lane) * 4 + i] = W[((size_t)m * cin + ch) * 9 + tp];
        }
  return dw.upload(f);
}

void launch_conv3x3_n16_f32(const Tensor& x, const Tensor& y, const float* wt, const float* scale, const float* shift, int act, hipStream_t s) {
  FE_CHECK((x.c == 16 || x.c == 32 || x.c == 64) && y.c == 16 && y.n == x.n && y.h == x.h && y.w == x.w && x.ld % 4 == 0 && y.ld % 4 == 0 &&
               (((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0,
           "conv3x3_n16: %dx%dx%d (ld %d) -> %d channels (ld %d)", x.h, x.w, x.c, x.ld, y.c, y.ld);
  N16Params p{};
  p.x = x.p; p.ldx = x.ld; p.B = x.n; p.H = x.h; p.W = x.w;
  p.y = y.p; p.ldy = y.ld;
  p.wt = wt; p.scale = scale; p.shift = shift; p.act = act;
  p.tiles_x = (x.w + 15) / 16; p.tiles_y = (x.h + 15) / 16; p.ntiles = x.n * p.tiles_x * p.tiles_y;
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
#define N16_LAUNCH(CIN)                                                                                              \
  {                                                                                                                  \
    constexpr int LDS = 324 * (CIN * 4 + 16) + 9 * (CIN / 16) * 64 * 16;                                                                        \
    static std::atomic<uint64_t> done{0};                                                                            \
    ensure_dynamic_lds((const void*)conv3x3_n16_f32_kernel<CIN>, LDS, done);                                         \
    hipLaunchKernelGGL(conv3x3_n16_f32_kernel<CIN>, dim3(grid), dim3(256), LDS, s, p);                               \
  }
  if (x.c == 64) N16_LAUNCH(64)
  else if (x.c == 32) N16_LAUNCH(32)
  else N16_LAUNCH(16)
#undef N16_LAUNCH
  FE_HIP(hipGetLastError());
}

}  // namespace fe

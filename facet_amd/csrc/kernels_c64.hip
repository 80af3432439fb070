// Halo-tiled 3x3 convolution, 64 -> 64 channels, stride 1, padding 1, for the 2-byte models - and, chained behind it in the same launch,
// the 1x1 expand 64 -> 256 + residual + ReLU that ends a ResNet-50 layer1 bottleneck (torchvision / timm Bottleneck: conv2, bn2, relu,
// conv3, bn3, += identity, relu; the reference runs it inside pyiqa's TOPIQ backbone, models/pyiqa_scorer.py; oracle/topiq.py).
//
// The generic implicit-GEMM kernel fetches every 64-channel input row nine times (once per tap, by LDS-DMA) and reached 480 TFLOP/s on
// this shape (K = 576, N = 64: 0.19 of the matrix peak, 1.7 TB/s); the bottleneck's 64-channel intermediate then went to HBM and came
// back for the expand. Here one persistent workgroup owns an output tile of 16 rows x 14 columns: the 18 x 16 input patch is loaded
// ONCE into LDS (pixel rows of 64 channels, pitch 144 B), the nine taps are `base + immediate` reads of it, and the 3x3 weights (72 KB,
// in matrix-fragment order, prepared on the host) stay in LDS for the life of the workgroup. The output region is walked as 16 rows x
// 16 columns = 8 pixel tiles of 32 (two per wave; columns 14 and 15 wrap into the next patch row and are discarded), so a tap is one
// constant offset for the whole tile.
//
// Layout (v_mfma_f32_32x32x16, TRANSPOSED: A = weights, B = pixels): a lane's 16 accumulators are 16 channels of ONE pixel, so after
// scale / shift / activation they pack straight into the B fragments of the expand (K = the 64 mid channels, W3's K index permuted on
// the host to the accumulator order): the 64-channel tensor never leaves the registers. The expand's 256 channels come out 32 at a
// time (one matrix tile), get scale / shift + identity + ReLU and go to HBM as 8-byte pieces (two lanes complete 16 B of a pixel row).
#include "fe_common.h"
#include "engine.h"

#include <atomic>
#include <vector>

namespace fe {

constexpr int C64_PITCH = 144;                         // bytes per patch pixel row (64 channels + 16: conflict-free 16-byte reads)
constexpr int C64_PATCH_ROWS = 18 * 16 + 2;            // reads of the discarded columns reach row 255 + 34
constexpr int C64_W2_BYTES = 9 * 4 * 2 * 1024;         // [tap][k-step][channel half] fragments
constexpr int C64_W3_BYTES = 8 * 4 * 1024;             // [out tile of 32][k-step] fragments
constexpr int C64_SS_FLOATS = 64 * 2 + 256 * 2;
constexpr int C64_LDS = C64_W2_BYTES + C64_PATCH_ROWS * C64_PITCH + C64_W3_BYTES + C64_SS_FLOATS * 4;

template <class E>
struct C64Params {
  const E* x; int ldx;                 // [B][H][W][64]
  int B, H, W;
  E* y; int ldy;                       // chained: [B][H][W][256]; plain: [B][H][W][64]
  const E* res; int ldr;               // chained: the identity
  const E* frag2; const E* frag3;      // fragment blobs
  const float *scale2, *shift2, *scale3, *shift3;   // null: 1 / 0
  int act2;                            // plain form only (-1 template): named activation
  int tiles_x, tiles_y, ntiles;
};

typedef float c64_f2 __attribute__((ext_vector_type(2)));
template <int A>
__device__ __forceinline__ c64_f2 c64_act2(const c64_f2 v, const int act) {
  if constexpr (A == ACT_RELU) {
    return c64_f2{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)};
  } else if constexpr (A == ACT_GELU) {      // fe_gelu_fast on two values (kernels_gate.hip)
    const c64_f2 c2 = {-0.10294324f, -0.10294324f}, c1 = {-2.3022082f, -2.3022082f};
    const c64_f2 z = v * __builtin_elementwise_fma(v * v, c2, c1);
    c64_f2 e;
    e.x = __builtin_amdgcn_exp2f(z.x); e.y = __builtin_amdgcn_exp2f(z.y);
    const c64_f2 d = e + 1.0f;
    c64_f2 rr;
    rr.x = __builtin_amdgcn_rcpf(d.x); rr.y = __builtin_amdgcn_rcpf(d.y);
    return v * rr;
  } else {
    return c64_f2{fe_apply_act_fast(v.x, act), fe_apply_act_fast(v.y, act)};
  }
}

// CHAIN: the expand + identity + ReLU behind the 3x3 (its activation is then ReLU too). A2: the 3x3's activation (plain form).
template <class E, bool CHAIN, int A2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_c64_kernel(const C64Params<E> p) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  char* const sW2 = smem_c;
  char* const sPatch = smem_c + C64_W2_BYTES;
  char* const sW3 = sPatch + C64_PATCH_ROWS * C64_PITCH;
  float* const sSS = reinterpret_cast<float*>(sW3 + C64_W3_BYTES);      // scale2[64] shift2[64] scale3[256] shift3[256]
  const E* const tag = nullptr;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;

  {
    const uint4* src = reinterpret_cast<const uint4*>(p.frag2);
    uint4* dst = reinterpret_cast<uint4*>(sW2);
    for (int i = t; i < C64_W2_BYTES / 16; i += 256) dst[i] = src[i];
    if (CHAIN) {
      const uint4* s3 = reinterpret_cast<const uint4*>(p.frag3);
      uint4* d3 = reinterpret_cast<uint4*>(sW3);
      for (int i = t; i < C64_W3_BYTES / 16; i += 256) d3[i] = s3[i];
    }
    if (t < 64) { sSS[t] = p.scale2 ? p.scale2[t] : 1.f; sSS[64 + t] = p.shift2 ? p.shift2[t] : 0.f; }
    if (CHAIN) { sSS[128 + t] = p.scale3 ? p.scale3[t] : 1.f; sSS[384 + t] = p.shift3 ? p.shift3[t] : 0.f; }
  }
  // the thread's nine 16-byte pieces of a patch (288 pixel rows x 8 pieces): where they come from relative to the tile origin, where they go
  int pc_dy[9], pc_dx[9], pc_rel[9], pc_dst[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int c = t + 256 * i, row = c >> 3, part = c & 7;
    pc_dy[i] = (row >> 4) - 1; pc_dx[i] = (row & 15) - 1;
    pc_rel[i] = (pc_dy[i] * p.W + pc_dx[i]) * p.ldx + part * 8;
    pc_dst[i] = row * C64_PITCH + part * 16;
  }
  auto tile_origin = [&](const int v, int& b, int& y0, int& x0) {
    const int per = p.tiles_x * p.tiles_y;
    b = v / per;
    const int rem = v - b * per, ty = rem / p.tiles_x;
    y0 = ty * 16; x0 = (rem - ty * p.tiles_x) * 14;
  };
  fe_v4f pf[9];
  auto fetch_patch = [&](const int v) {      // a tile's patch into registers (zeros outside the image)
    if (v >= p.ntiles) return;
    int b, y0, x0;
    tile_origin(v, b, y0, x0);
    const E* const org = p.x + (((size_t)b * p.H + y0) * p.W + x0) * p.ldx;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const bool in = (unsigned)(y0 + pc_dy[i]) < (unsigned)p.H && (unsigned)(x0 + pc_dx[i]) < (unsigned)p.W;
      const fe_v4f q = *reinterpret_cast<const fe_v4f*>(org + (in ? pc_rel[i] : 0));
      pf[i] = in ? q : fe_v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<fe_v4f*>(sPatch + pc_dst[i]) = pf[i];
  };
  // the lane's scale / shift quads of the 3x3 (channels mt 32 + 8 g + 4 h + 0..3), in registers
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see kernels_gate.hip (loads in flight at the loop head cost counted waits inside it)
  __syncthreads();
  float4 sc2[2][4], sh2[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sc2[mt][g] = *reinterpret_cast<const float4*>(sSS + mt * 32 + 8 * g + 4 * h);
      sh2[mt][g] = *reinterpret_cast<const float4*>(sSS + 64 + mt * 32 + 8 * g + 4 * h);
    }
  fetch_patch(blockIdx.x);
  store_patch();
  __syncthreads();

  for (int v = blockIdx.x; v < p.ntiles; v += gridDim.x) {
    int b, y0, x0;
    tile_origin(v, b, y0, x0);
    fetch_patch(v + (int)gridDim.x);      // lands behind the matrix loop; stored to LDS once every wave has left the loop

    // ---- the 3x3: 36 K-steps (tap x 16 channels), two pixel tiles x two channel halves per wave --------------------------------------
    const int qb = wave * 64 + r;
    // the lane's two output pixels (its columns of the two pixel tiles)
    bool okn[2]; size_t pixn[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int q = qb + n * 32;
      const int oy = y0 + (q >> 4), ocol = q & 15, ox = x0 + ocol;
      okn[n] = ocol < 14 && oy < p.H && ox < p.W;
      pixn[n] = ((size_t)b * p.H + (okn[n] ? oy : y0)) * p.W + (okn[n] ? ox : x0);
    }
    // chained form: the identity and the result cross HBM as whole pixel rows (16 bytes per lane, 16 lanes = 256 contiguous bytes of one
    // pixel) through a wave-private LDS image, not as the accumulator layout's 8-byte quads: as quads every store instruction touched
    // 32 cache lines 16 bytes at a time and the launch sat at 3.5 TB/s. Piece i (of 8) of a pixel tile's 32 pixels x 128 channels: pixel lane / 16 + 4 i, 16 bytes at lane % 16.
    const int pc_part = lane & 15;
    bool pc_ok[16]; size_t pc_pix[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = i >> 3, px = (lane >> 4) + 4 * (i & 7), q = wave * 64 + n * 32 + px;
      const int oy = y0 + (q >> 4), ocol = q & 15, ox = x0 + ocol;
      pc_ok[i] = ocol < 14 && oy < p.H && ox < p.W;
      pc_pix[i] = ((size_t)b * p.H + (pc_ok[i] ? oy : y0)) * p.W + (pc_ok[i] ? ox : x0);
    }
    fe_v4f idr[2][8];      // the identity rows of one half (128 channels) of one pixel tile, and of the one after it
    auto fetch_identity = [&](const int n, const int half, fe_v4f (&dst)[8]) __attribute__((always_inline)) {
      if constexpr (CHAIN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const fe_v4f*>(p.res + pc_pix[n * 8 + i] * p.ldr + half * 128 + pc_part * 8);
      }
    };
    fetch_identity(0, 0, idr[0]);      // lands behind the matrix loop
    const char* const bB = sPatch + qb * C64_PITCH + h * 16;
    const char* const bA = sW2 + lane * 16;
    fe_f32x16 acc[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][mt][e] = 0.f;
    fe_v4f fr[2][4];
    auto read_step = [&](const int st, fe_v4f (&f)[4]) __attribute__((always_inline)) {
      const int tp = st >> 2, ks = st & 3, dy = tp / 3, dx = tp - dy * 3;
      f[0] = *reinterpret_cast<const fe_v4f*>(bA + (st * 2 + 0) * 1024);
      f[1] = *reinterpret_cast<const fe_v4f*>(bA + (st * 2 + 1) * 1024);
#pragma unroll
      for (int n = 0; n < 2; ++n) f[2 + n] = *reinterpret_cast<const fe_v4f*>(bB + (n * 32 + dy * 16 + dx) * C64_PITCH + ks * 32);
    };
    read_step(0, fr[0]);
#pragma unroll
    for (int st = 0; st < 36; ++st) {
      if (st + 1 < 36) read_step(st + 1, fr[(st + 1) & 1]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        acc[n][0] = fe_mfma16(tag, fr[st & 1][0], fr[st & 1][2 + n], acc[n][0]);
        acc[n][1] = fe_mfma16(tag, fr[st & 1][1], fr[st & 1][2 + n], acc[n][1]);
      }
    }
    __syncthreads();      // every wave is done with this patch

    // ---- epilogue ----------------------------------------------------------------------------------------------------------------------
    if constexpr (!CHAIN) {
      store_patch();        // the next patch (stale data past the last tile)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const bool ok = okn[n];
        E* const yr = p.y + pixn[n] * p.ldy;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 s = sc2[mt][g], f = sh2[mt][g];
            const c64_f2 lo = c64_act2<A2>(__builtin_elementwise_fma(c64_f2{acc[n][mt][4 * g + 0], acc[n][mt][4 * g + 1]}, c64_f2{s.x, s.y}, c64_f2{f.x, f.y}), p.act2);
            const c64_f2 hi = c64_act2<A2>(__builtin_elementwise_fma(c64_f2{acc[n][mt][4 * g + 2], acc[n][mt][4 * g + 3]}, c64_f2{s.z, s.w}, c64_f2{f.z, f.w}), p.act2);
            if (ok) *reinterpret_cast<uint2*>(yr + mt * 32 + 8 * g + 4 * h) = make_uint2(fe_pack2(tag, lo.x, lo.y), fe_pack2(tag, hi.x, hi.y));
          }
      }
    } else {
      // the patch buffer is free until the next patch is stored: 32 pixels x (256 + 16) bytes per wave
      constexpr int SP = 272;
      char* const stg = sPatch + wave * (32 * SP);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        // mid = relu(acc * scale2 + shift2) as the B fragments of the expand: step s = (mt, half) holds quads 2 half, 2 half + 1 of mt
        fe_v4f bq[4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            unsigned u[4];
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              const int g = 2 * half + g2;
              const float4 s = sc2[mt][g], f = sh2[mt][g];
              const c64_f2 lo = c64_act2<ACT_RELU>(__builtin_elementwise_fma(c64_f2{acc[n][mt][4 * g + 0], acc[n][mt][4 * g + 1]}, c64_f2{s.x, s.y}, c64_f2{f.x, f.y}), 0);
              const c64_f2 hi = c64_act2<ACT_RELU>(__builtin_elementwise_fma(c64_f2{acc[n][mt][4 * g + 2], acc[n][mt][4 * g + 3]}, c64_f2{s.z, s.w}, c64_f2{f.z, f.w}), 0);
              u[2 * g2] = fe_pack2(tag, lo.x, lo.y); u[2 * g2 + 1] = fe_pack2(tag, hi.x, hi.y);
            }
            bq[mt * 2 + half] = fe_v4f{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
          }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {      // output channels 128 hf .. + 128
          const int cur = (n * 2 + hf) & 1;
          // identity rows -> LDS (row layout); the next half's rows are requested now
#pragma unroll
          for (int i = 0; i < 8; ++i) *reinterpret_cast<fe_v4f*>(stg + ((lane >> 4) + 4 * i) * SP + pc_part * 16) = idr[cur][i];
          if (hf == 0) fetch_identity(n, 1, idr[cur ^ 1]);
          else if (n == 0) fetch_identity(1, 0, idr[cur ^ 1]);
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) {
            const int m = hf * 4 + mm;
            fe_f32x16 a3;
#pragma unroll
            for (int e = 0; e < 16; ++e) a3[e] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) a3 = fe_mfma16(tag, *reinterpret_cast<const fe_v4f*>(sW3 + (m * 4 + st) * 1024 + lane * 16), bq[st], a3);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 s = *reinterpret_cast<const float4*>(sSS + 128 + m * 32 + 8 * g + 4 * h);
              const float4 f = *reinterpret_cast<const float4*>(sSS + 384 + m * 32 + 8 * g + 4 * h);
              uint2* const cell = reinterpret_cast<uint2*>(stg + r * SP + (mm * 32 + 8 * g + 4 * h) * 2);      // the lane's quad of its pixel
              const uint2 id = *cell;
              float i0, i1, i2, i3;
              fe_unpack2(tag, id.x, i0, i1); fe_unpack2(tag, id.y, i2, i3);
              const c64_f2 lo = c64_act2<ACT_RELU>(__builtin_elementwise_fma(c64_f2{a3[4 * g + 0], a3[4 * g + 1]}, c64_f2{s.x, s.y}, c64_f2{f.x, f.y}) + c64_f2{i0, i1}, 0);
              const c64_f2 hi = c64_act2<ACT_RELU>(__builtin_elementwise_fma(c64_f2{a3[4 * g + 2], a3[4 * g + 3]}, c64_f2{s.z, s.w}, c64_f2{f.z, f.w}) + c64_f2{i2, i3}, 0);
              *cell = make_uint2(fe_pack2(tag, lo.x, lo.y), fe_pack2(tag, hi.x, hi.y));
            }
          }
          // result rows: LDS -> HBM, 256 contiguous bytes per pixel
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const fe_v4f o = *reinterpret_cast<const fe_v4f*>(stg + ((lane >> 4) + 4 * i) * SP + pc_part * 16);
            if (pc_ok[n * 8 + i]) *reinterpret_cast<fe_v4f*>(p.y + pc_pix[n * 8 + i] * p.ldy + hf * 128 + pc_part * 8) = o;
          }
        }
      }
      __syncthreads();      // every wave is done with its image in the patch buffer
      store_patch();        // the next patch (stale data past the last tile)
    }
    __syncthreads();      // the next patch is in LDS
  }
}

// Host: the two fragment blobs. W2 [64][64][3][3] -> [tap][k-step][half][lane][8]; W3 [256][64] -> [out tile][step s][lane][8] with the K
// slot (s = (mt, half), lane half h, j) = mid channel mt 32 + 8 (2 half + j / 4) + 4 h + j % 4 (the accumulator order of the 3x3).
void build_c64_fragments(DeviceWeights& dw, const float* W2, const float* W3, void** frag2, void** frag3) {
  const int prec = dw.prec;
  std::vector<uint16_t> f2((size_t)C64_W2_BYTES / 2), f3((size_t)C64_W3_BYTES / 2, 0);
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 31, h = lane >> 5;
    for (int tp = 0; tp < 9; ++tp)
      for (int ks = 0; ks < 4; ++ks)
        for (int mt = 0; mt < 2; ++mt)
          for (int j = 0; j < 8; ++j)
            f2[(size_t)((tp * 4 + ks) * 2 + mt) * 512 + lane * 8 + j] = f32_to_half_bits(W2[((size_t)(mt * 32 + r) * 64 + ks * 16 + h * 8 + j) * 9 + tp], prec);
    if (W3)
      for (int m = 0; m < 8; ++m)
        for (int s = 0; s < 4; ++s)
          for (int j = 0; j < 8; ++j) {
            const int mt = s >> 1, half = s & 1, ch = mt * 32 + 8 * (2 * half + j / 4) + 4 * h + (j & 3);
            f3[(size_t)(m * 4 + s) * 512 + lane * 8 + j] = f32_to_half_bits(W3[(size_t)(m * 32 + r) * 64 + ch], prec);
          }
  }
  *frag2 = dw.upload_raw(f2.data(), f2.size() * sizeof(uint16_t));
  if (frag3) *frag3 = W3 ? dw.upload_raw(f3.data(), f3.size() * sizeof(uint16_t)) : nullptr;
}

template <class E>
void launch_conv3x3_c64(const TensorT<E>& x, const TensorT<E>& y, const TensorT<E>* res, const void* frag2, const void* frag3, const float* scale2, const float* shift2,
                        const float* scale3, const float* shift3, int act2, hipStream_t s) {
  const bool chain = frag3 != nullptr;
  FE_CHECK(x.c == 64 && y.n == x.n && y.h == x.h && y.w == x.w && y.c == (chain ? 256 : 64) && x.ld % 8 == 0 && y.ld % 4 == 0, "conv3x3_c64: %dx%dx%d -> %dx%dx%d", x.h, x.w, x.c,
           y.h, y.w, y.c);
  FE_CHECK(!chain || (res && res->c == 256 && res->pixels() == y.pixels() && res->ld % 8 == 0 && y.ld % 8 == 0 && act2 == ACT_RELU), "conv3x3_c64: the chained form needs the identity (16-byte rows) and ReLU");
  FE_CHECK(chain || (act2 >= ACT_NONE && act2 <= ACT_SOFTPLUS && act2 != ACT_PRELU), "conv3x3_c64: activation %d", act2);
  FE_CHECK((size_t)x.n * x.h * x.w * (size_t)(x.ld > y.ld ? x.ld : y.ld) < (1ull << 40), "conv3x3_c64: tensor too large");
  C64Params<E> p{};
  p.x = x.p; p.ldx = x.ld; p.B = x.n; p.H = x.h; p.W = x.w;
  p.y = y.p; p.ldy = y.ld;
  p.res = res ? res->p : nullptr; p.ldr = res ? res->ld : 0;
  p.frag2 = (const E*)frag2; p.frag3 = (const E*)frag3;
  p.scale2 = scale2; p.shift2 = shift2; p.scale3 = scale3; p.shift3 = shift3;
  p.act2 = act2;
  p.tiles_x = (x.w + 13) / 14; p.tiles_y = (x.h + 15) / 16; p.ntiles = x.n * p.tiles_x * p.tiles_y;
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
#define C64_LAUNCH(CH, A)                                                                                   \
  {                                                                                                         \
    static std::atomic<uint64_t> done{0};                                                                   \
    ensure_dynamic_lds((const void*)conv3x3_c64_kernel<E, CH, A>, C64_LDS, done);                           \
    hipLaunchKernelGGL((conv3x3_c64_kernel<E, CH, A>), dim3(grid), dim3(256), C64_LDS, s, p);               \
  }
  if (chain) C64_LAUNCH(true, ACT_RELU)
  else if (act2 == ACT_RELU) C64_LAUNCH(false, ACT_RELU)
  else if (act2 == ACT_GELU) C64_LAUNCH(false, ACT_GELU)
  else C64_LAUNCH(false, -1)
#undef C64_LAUNCH
  FE_HIP(hipGetLastError());
}
template void launch_conv3x3_c64<bf16>(const TensorT<bf16>&, const TensorT<bf16>&, const TensorT<bf16>*, const void*, const void*, const float*, const float*, const float*,
                                       const float*, int, hipStream_t);
template void launch_conv3x3_c64<f16>(const TensorT<f16>&, const TensorT<f16>&, const TensorT<f16>*, const void*, const void*, const float*, const float*, const float*,
                                      const float*, int, hipStream_t);

}  // namespace fe

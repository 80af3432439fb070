// TOPIQ GatedConv of the full-resolution pyramid level (64 channels at H/2 x W/2: 8.4 M pixels per 32 images of 1024 x 1024) fused with
// the 16 x 16 average pool that follows it, for the 2-byte models (reference: pyiqa CFANet's GatedConv as facet calls it through
// models/pyiqa_scorer.py; oracle/topiq.py restates it).
//
//   wa = act(W0 f + b0)            1x1, 64 -> 64   (weight_blk[0] composed with the x2 half of splitconv, see build_topiq_head)
//   wb = act(W2 * wa + b2)         3x3, 64 -> 64, zero padding
//   wc = sigmoid(W4 * wb + b4)     3x3, 64 -> 1,  zero padding
//   out = mean_{16x16}( act(Wx f + bx) . wc )      (x1 half of splitconv, gated, adaptive average pool to H/16 x W/16)
//
// As four launches + the pool this level moved ~8.6 GB per 32 images (f read twice, wa / wb / wc / the gated tensor written and read
// back) for 1.07 GB of input: 3.9 ms of the 23 ms a micro-batch of TOPIQ takes in fp16. Here one workgroup owns one pooling window:
// it reads the 20 x 20 halo patch of f once, keeps wa (20 x 20), the nine per-tap partial sums of wc (18 x 18) and wc (16 x 16) in
// LDS, and writes 64 pooled values. W2 (72 KB) stays in LDS for the life of the (persistent) workgroup, pre-arranged on the host
// in matrix-instruction fragment order, so every operand read is `base + immediate`, 1 KiB contiguous per wave, conflict-free.
//
// Matrix layout (v_mfma_f32_32x32x16): stages 1 and 2 run TRANSPOSED - A = weights (rows = output channels), B = pixels (columns), so a
// lane's 16 accumulators are 16 channels of ONE pixel (channels 8 g + 4 h + j): the activation result packs straight into the next
// stage's B fragment (stage 2 -> the 64 -> 9-tap product) or into 8-byte LDS stores (stage 1 -> wa rows). The 3x3 64 -> 1 convolution
// is NOT a second spatial pass over wb: P[t][q] = W4[t] . wb[q] (a 64 x 9 product per pixel, four more matrix instructions per 32
// pixels, W4's K index permuted on the host to the accumulator order) and wc[p] = sigmoid(b4 + sum_t P[t][p + off_t]) - wb never
// leaves the registers. Stage 4 runs untransposed (rows = pixels) so the pool is an in-lane sum over the accumulator rows.
#include "fe_common.h"
#include "engine.h"

#include <atomic>
#include <vector>

namespace fe {

constexpr int G64_WA_PITCH = 144;                       // bytes per wa pixel row: 64 channels + 16 (conflict-free ds_read_b128 at stride 1 pixel)
constexpr int G64_WA_ROWS = 426;                        // 20 x 20 written; reads of the discarded columns reach row 383 + 42
constexpr int G64_W2_BYTES = 9 * 4 * 2 * 1024;          // [tap][k-step][channel half] fragments
constexpr int G64_P_STRIDE = 384;                       // 12 pixel tiles of the 18-row x 20-column wb region
constexpr int G64_LDS = G64_W2_BYTES + G64_WA_ROWS * G64_WA_PITCH + 9 * G64_P_STRIDE * 4 + 256 * 4 + 256 * 4 + 256 * 4;
// fragment blob (elements): W2 | W0 | Wx | W4, then fp32 biases b0[64] b2[64] bx[64] b4
constexpr int G64_OFF_W0 = G64_W2_BYTES / 2, G64_OFF_WX = G64_OFF_W0 + 4 * 2 * 512, G64_OFF_W4 = G64_OFF_WX + 4 * 2 * 512, G64_FRAG_ELEMS = G64_OFF_W4 + 4 * 512;

template <class E>
struct Gate64Params {
  const E* x; int ldx;                 // [B][H][W][64]
  int B, H, W;
  E* out; int ldo;                     // [B][H/16][W/16][64]
  const E* frag;                       // G64_FRAG_ELEMS elements
  const float* bias;                   // 193 floats
  int wblk_act, gate_act;
  int tiles_x, tiles_y, ntiles;
  long long* stamps;                   // developer builds (-DG64_STAMPS): per-stage time of workgroup 0, in s_memtime ticks
};

template <class E>
__device__ __forceinline__ fe_v4f g64_pack8(const float (&v)[8]) {
  fe_v4f o;
  o[0] = __uint_as_float(fe_pack2((const E*)nullptr, v[0], v[1]));
  o[1] = __uint_as_float(fe_pack2((const E*)nullptr, v[2], v[3]));
  o[2] = __uint_as_float(fe_pack2((const E*)nullptr, v[4], v[5]));
  o[3] = __uint_as_float(fe_pack2((const E*)nullptr, v[6], v[7]));
  return o;
}

// AW / AG: the two activations at compile time (ACT_GELU: the tanh form every 2-byte epilogue uses) or -1 = named by the parameters -
// a per-value runtime choice left 1,700 branches in the kernel and serialised the activation arithmetic behind them. The kernel is
// bound by this arithmetic, not by the matrix pipe (264 activations per lane and window against 270 matrix instructions per wave), so
// it runs on PAIRS: v_pk_mul / v_pk_fma / v_pk_add_f32 do two values per 4-cycle issue; only v_exp_f32 and v_rcp_f32 stay scalar.
typedef float g64_f2 __attribute__((ext_vector_type(2)));
template <int A>
__device__ __forceinline__ g64_f2 g64_act2(const g64_f2 v, const int act) {
  if constexpr (A == ACT_GELU) {      // fe_gelu_fast on two values
    const g64_f2 c2 = {-0.10294324f, -0.10294324f}, c1 = {-2.3022082f, -2.3022082f};
    const g64_f2 z = v * __builtin_elementwise_fma(v * v, c2, c1);
    g64_f2 e;
    e.x = __builtin_amdgcn_exp2f(z.x); e.y = __builtin_amdgcn_exp2f(z.y);
    const g64_f2 d = e + 1.0f;
    g64_f2 rr;
    rr.x = __builtin_amdgcn_rcpf(d.x); rr.y = __builtin_amdgcn_rcpf(d.y);
    return v * rr;
  } else {
    g64_f2 o;
    o.x = fe_apply_act_fast(v.x, act); o.y = fe_apply_act_fast(v.y, act);
    return o;
  }
}

template <class E, int AW, int AG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void topiq_gate64_kernel(const Gate64Params<E> p) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  char* const sW2 = smem_g;
  char* const sWA = smem_g + G64_W2_BYTES;
  float* const sP = reinterpret_cast<float*>(sWA + G64_WA_ROWS * G64_WA_PITCH);
  float* const sWC = sP + 9 * G64_P_STRIDE;
  float* const sRed = sWC + 256;
  float* const sB = sRed + 256;
  const E* const tag = nullptr;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- once per workgroup: W2 and the biases to LDS, the small weight sets to registers ------------------------------------------------
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.frag);
    uint4* dst = reinterpret_cast<uint4*>(sW2);
    for (int i = t; i < G64_W2_BYTES / 16; i += 256) dst[i] = src[i];
    if (t < 193) sB[t] = p.bias[t];
  }
  fe_v4f w0f[4][2], wxf[4][2], w4f[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      w0f[ks][mt] = *reinterpret_cast<const fe_v4f*>(p.frag + G64_OFF_W0 + (ks * 2 + mt) * 512 + lane * 8);
      wxf[ks][mt] = *reinterpret_cast<const fe_v4f*>(p.frag + G64_OFF_WX + (ks * 2 + mt) * 512 + lane * 8);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s) w4f[s] = *reinterpret_cast<const fe_v4f*>(p.frag + G64_OFF_W4 + s * 512 + lane * 8);
  // every load above has landed before the window loop: a weight fragment still "in flight" at the loop head makes the compiler guard
  // its uses inside the loop with counted vmcnt waits that, from the second window on, wait for the PREFETCHES issued just before
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  __syncthreads();
  const float b4 = sB[192];
  // the lane's bias quads (channels mt 32 + 8 g + 4 h + 0..3), in registers for the life of the workgroup: read from LDS beside every
  // quad of activations they cost one exposed LDS round trip per four values
  float4 rb0[2][4], rb2[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      rb0[mt][g] = *reinterpret_cast<const float4*>(sB + mt * 32 + 8 * g + 4 * h);
      rb2[mt][g] = *reinterpret_cast<const float4*>(sB + 64 + mt * 32 + 8 * g + 4 * h);
    }
  const float rbx[2] = {sB[128 + r], sB[128 + 32 + r]};

  // The patch of a window arrives as matrix fragments straight from HBM (16 bytes per lane: 8 channels of one pixel). One wave per SIMD
  // hides nothing by itself, so the loads run one stage ahead: the NEXT window's 20 x 20 patch (up to 4 pixel tiles per wave) and this
  // window's 16 x 16 centre (stage 4) are requested right after stage 2's matrix loop and land behind its activation arithmetic.
  // Pixel tiles 0..11 of the patch go three to a wave; the 13th (pixels 384..399) is computed by all four waves, a quarter of its
  // channels each - as a fourth whole tile on one wave it made that wave the 4 : 3 straggler of stage 1.
  fe_v4f bf[4][4], af[2][4];
  // per-lane geometry that does not depend on the window: pixel offsets of the lane's patch / wb / centre pixels from the window origin
  int pk_dy[4], pk_dx[4], pk_rel[4], pk_q[4]; bool pk_in[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int q1 = (k < 3 ? wave + 4 * k : 12) * 32 + r;
    const int y1 = q1 / 20, x1 = q1 - y1 * 20;
    pk_q[k] = q1; pk_dy[k] = y1 - 2; pk_dx[k] = x1 - 2; pk_in[k] = q1 < 400;
    pk_rel[k] = ((y1 - 2) * p.W + (x1 - 2)) * p.ldx;
  }
  int wb_dy[3], wb_dx[3]; bool wb_in[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int q2 = wave * 96 + n * 32 + r;
    const int y2 = q2 / 20, x2 = q2 - y2 * 20;
    wb_dy[n] = y2 - 1; wb_dx[n] = x2 - 1; wb_in[n] = y2 < 18 && x2 < 18;
  }
  int c_rel[2];
#pragma unroll
  for (int mp = 0; mp < 2; ++mp) {
    const int pp = (wave * 2 + mp) * 32 + r;
    c_rel[mp] = ((pp >> 4) * p.W + (pp & 15)) * p.ldx;
  }
  auto window_origin = [&](const int v, int& y0, int& x0) -> const E* {      // scalar: the window's first pixel (always inside the image)
    const int per = p.tiles_x * p.tiles_y;
    const int b = v / per, rem = v - b * per, ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    y0 = ty * 16; x0 = tx * 16;
    return p.x + (((size_t)b * p.H + y0) * p.W + x0) * p.ldx + h * 8;
  };
  auto patch_valid = [&](const int k, const int y0, const int x0) -> bool {
    return pk_in[k] && (unsigned)(y0 + pk_dy[k]) < (unsigned)p.H && (unsigned)(x0 + pk_dx[k]) < (unsigned)p.W;
  };
  auto fetch_patch = [&](const int v) {
    if (v >= p.ntiles) return;
    int y0, x0;
    const E* const org = window_origin(v, y0, x0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const E* src = org + (patch_valid(k, y0, x0) ? pk_rel[k] : 0);      // a pixel outside the image reads the window origin; masked in stage 1
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[k][ks] = *reinterpret_cast<const fe_v4f*>(src + ks * 16);
    }
  };
  fetch_patch(blockIdx.x);
#ifdef G64_STAMPS
  long long tacc[5] = {0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
#define G64_STAMP(i) { const long long n_ = __builtin_readcyclecounter(); tacc[i] += n_ - tl; tl = n_; }
#else
#define G64_STAMP(i)
#endif

  for (int v = blockIdx.x; v < p.ntiles; v += gridDim.x) {
    int y0, x0;
    const E* const org = window_origin(v, y0, x0);

    // ---- stage 1: wa on the 20 x 20 patch (13 pixel tiles of 32: wave w takes w, w + 4, w + 8 and wave 0 the 13th) -----------------------
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q1 = pk_q[k];
      const float vm = patch_valid(k, y0, x0) ? 1.f : 0.f;      // zero padding of the 3x3 that reads wa (zero, not act(b0))
      char* const row = sWA + q1 * G64_WA_PITCH;
      auto finish = [&](const fe_f32x16& a, const int mt, const int g, const float4 bb) __attribute__((always_inline)) {
        const int c0 = mt * 32 + 8 * g + 4 * h;
        const g64_f2 m2 = {vm, vm};
        const g64_f2 lo = g64_act2<AW>(g64_f2{a[4 * g + 0], a[4 * g + 1]} + g64_f2{bb.x, bb.y}, p.wblk_act) * m2;
        const g64_f2 hi = g64_act2<AW>(g64_f2{a[4 * g + 2], a[4 * g + 3]} + g64_f2{bb.z, bb.w}, p.wblk_act) * m2;
        *reinterpret_cast<uint2*>(row + c0 * 2) = make_uint2(fe_pack2(tag, lo.x, lo.y), fe_pack2(tag, hi.x, hi.y));
      };
      if (k < 3) {
        fe_f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = fe_mfma16(tag, w0f[ks][mt], bf[k][ks], acc[mt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) finish(acc[mt], mt, g, rb0[mt][g]);
      } else {      // the shared tile: channel half wave & 1, quads 2 (wave >> 1) and + 1 of it
        fe_f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // (uniform branches, not indexed registers: a run-time index into w0f / rb0 would move the arrays to scratch memory)
        if (wave & 1) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = fe_mfma16(tag, w0f[ks][1], bf[3][ks], acc);
          if (wave >> 1) { finish(acc, 1, 2, rb0[1][2]); finish(acc, 1, 3, rb0[1][3]); } else { finish(acc, 1, 0, rb0[1][0]); finish(acc, 1, 1, rb0[1][1]); }
        } else {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = fe_mfma16(tag, w0f[ks][0], bf[3][ks], acc);
          if (wave >> 1) { finish(acc, 0, 2, rb0[0][2]); finish(acc, 0, 3, rb0[0][3]); } else { finish(acc, 0, 0, rb0[0][0]); finish(acc, 0, 1, rb0[0][1]); }
        }
      }
    }
    __syncthreads();
    G64_STAMP(0)

    // ---- stage 2: wb on 18 rows x 20 columns (columns 18, 19 discarded), three pixel tiles per wave; then P = W4 . wb ------------------
    {
      const int q2b = wave * 96 + r;
      const char* const bB = sWA + q2b * G64_WA_PITCH + h * 16;
      const char* const bA = sW2 + lane * 16;
      fe_f32x16 acc[3][2];
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[n][mt][e] = 0.f;
      // 36 K-steps (tap x 16 channels), fragments of step s + 1 requested before the six matrix instructions of step s
      fe_v4f fr[2][5];
      auto read_step = [&](const int st, fe_v4f (&f)[5]) __attribute__((always_inline)) {
        const int tp = st >> 2, ks = st & 3, dy = tp / 3, dx = tp - dy * 3;
        f[0] = *reinterpret_cast<const fe_v4f*>(bA + (st * 2 + 0) * 1024);
        f[1] = *reinterpret_cast<const fe_v4f*>(bA + (st * 2 + 1) * 1024);
#pragma unroll
        for (int n = 0; n < 3; ++n) f[2 + n] = *reinterpret_cast<const fe_v4f*>(bB + (n * 32 + dy * 20 + dx) * G64_WA_PITCH + ks * 32);
      };
      read_step(0, fr[0]);
#pragma unroll
      for (int st = 0; st < 36; ++st) {
        if (st + 1 < 36) read_step(st + 1, fr[(st + 1) & 1]);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          acc[n][0] = fe_mfma16(tag, fr[st & 1][0], fr[st & 1][2 + n], acc[n][0]);
          acc[n][1] = fe_mfma16(tag, fr[st & 1][1], fr[st & 1][2 + n], acc[n][1]);
        }
      }
      G64_STAMP(1)
      // requests that land behind the activation arithmetic below: this window's centre (stage 4), the next window's patch (stage 1)
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {
        const E* src = org + c_rel[mp];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[mp][ks] = *reinterpret_cast<const fe_v4f*>(src + ks * 16);
      }
      fetch_patch(v + (int)gridDim.x);
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int q2 = q2b + n * 32;
        const bool valid = wb_in[n] && (unsigned)(y0 + wb_dy[n]) < (unsigned)p.H && (unsigned)(x0 + wb_dx[n]) < (unsigned)p.W;
        fe_f32x16 pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) pacc[e] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            float vv[8];
            const g64_f2 m2 = {valid ? 1.f : 0.f, valid ? 1.f : 0.f};      // zero padding of the 64 -> 1 convolution
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              const int g = 2 * half + g2;
              const float4 bb = rb2[mt][g];
              const g64_f2 lo = g64_act2<AW>(g64_f2{acc[n][mt][4 * g + 0], acc[n][mt][4 * g + 1]} + g64_f2{bb.x, bb.y}, p.wblk_act) * m2;
              const g64_f2 hi = g64_act2<AW>(g64_f2{acc[n][mt][4 * g + 2], acc[n][mt][4 * g + 3]} + g64_f2{bb.z, bb.w}, p.wblk_act) * m2;
              vv[4 * g2 + 0] = lo.x; vv[4 * g2 + 1] = lo.y; vv[4 * g2 + 2] = hi.x; vv[4 * g2 + 3] = hi.y;
            }
            pacc = fe_mfma16(tag, w4f[mt * 2 + half], g64_pack8<E>(vv), pacc);
          }
        // rows of pacc = taps: h = 0 lanes hold taps 0..3 (e 0..3) and 8 (e 4), h = 1 lanes taps 4..7
        float* const pq = sP + q2;
        pq[(4 * h + 0) * G64_P_STRIDE] = pacc[0];
        pq[(4 * h + 1) * G64_P_STRIDE] = pacc[1];
        pq[(4 * h + 2) * G64_P_STRIDE] = pacc[2];
        pq[(4 * h + 3) * G64_P_STRIDE] = pacc[3];
        if (h == 0) pq[8 * G64_P_STRIDE] = pacc[4];
      }
    }
    __syncthreads();
    G64_STAMP(2)

    // ---- stage 3: wc of the 16 x 16 window, one pixel per thread -----------------------------------------------------------------------
    {
      const int cy = t >> 4, cx = t & 15;
      float s = b4;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) s += sP[(kh * 3 + kw) * G64_P_STRIDE + (cy + kh) * 20 + cx + kw];
      E wc;                                         // the gate is a tensor of the model's element type
      stf(&wc, fe_rcp_fast(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * s)));
      sWC[t] = ldf(&wc);
    }
    __syncthreads();

    G64_STAMP(3)
    // ---- stage 4: act(Wx f + bx) . wc summed over the window (rows = pixels: two tiles of 32 per wave; columns = channels) ----------------
    {
      float sum[2] = {0.f, 0.f};
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {
        fe_f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[nt] = fe_mfma16(tag, af[mp][ks], wxf[ks][nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const g64_f2 bx2 = {rbx[nt], rbx[nt]};
          g64_f2 s2 = {0.f, 0.f};
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 wc = *reinterpret_cast<const float4*>(sWC + (wave * 2 + mp) * 32 + 8 * g + 4 * h);
            s2 += g64_act2<AG>(g64_f2{acc[nt][4 * g + 0], acc[nt][4 * g + 1]} + bx2, p.gate_act) * g64_f2{wc.x, wc.y};
            s2 += g64_act2<AG>(g64_f2{acc[nt][4 * g + 2], acc[nt][4 * g + 3]} + bx2, p.gate_act) * g64_f2{wc.z, wc.w};
          }
          sum[nt] += s2.x + s2.y;
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        sum[nt] += __shfl_xor(sum[nt], 32);
        if (h == 0) sRed[wave * 64 + nt * 32 + r] = sum[nt];
      }
    }
    __syncthreads();
    if (t < 64) {
      const float m = ((sRed[t] + sRed[64 + t]) + (sRed[128 + t] + sRed[192 + t])) * (1.f / 256.f);
      stf(p.out + (size_t)v * p.ldo + t, m);
    }
    G64_STAMP(4)
  }
#ifdef G64_STAMPS
  if (p.stamps && blockIdx.x == 0 && t == 0)
    for (int i = 0; i < 5; ++i) p.stamps[i] = tacc[i];
#endif
}

// Host: the fragment blob of one level (see the kernel header for the orders). W0 [64][64] (composed), W2 [64][64][3][3], W4 [1][64][3][3],
// Wx [64][64] (x1 rows of splitconv), as fp32; prec = PREC_BF16 | PREC_F16.
void build_gate64_fragments(DeviceWeights& dw, GatedConvW& g, const float* W0, const float* B0, const float* W2, const float* B2, const float* W4, float b4,
                            const float* Wx, const float* Bx) {
  std::vector<uint16_t> f((size_t)G64_FRAG_ELEMS, 0);
  const int prec = dw.prec;
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 31, h = lane >> 5;
    for (int tp = 0; tp < 9; ++tp)
      for (int ks = 0; ks < 4; ++ks)
        for (int mt = 0; mt < 2; ++mt)
          for (int j = 0; j < 8; ++j) {
            const int co = mt * 32 + r, ci = ks * 16 + h * 8 + j;
            f[(size_t)((tp * 4 + ks) * 2 + mt) * 512 + lane * 8 + j] = f32_to_half_bits(W2[((size_t)co * 64 + ci) * 9 + tp], prec);
          }
    for (int ks = 0; ks < 4; ++ks)
      for (int mt = 0; mt < 2; ++mt)
        for (int j = 0; j < 8; ++j) {
          const int co = mt * 32 + r, ci = ks * 16 + h * 8 + j;
          f[(size_t)G64_OFF_W0 + (ks * 2 + mt) * 512 + lane * 8 + j] = f32_to_half_bits(W0[(size_t)co * 64 + ci], prec);
          f[(size_t)G64_OFF_WX + (ks * 2 + mt) * 512 + lane * 8 + j] = f32_to_half_bits(Wx[(size_t)co * 64 + ci], prec);
        }
    // W4 as the A operand of the 64 -> 9 product: row = tap, K slot (step s = (mt, half), lane half h, j) = channel mt 32 + 8 (2 half + j / 4) + 4 h + j % 4
    for (int s = 0; s < 4; ++s)
      for (int j = 0; j < 8; ++j) {
        const int mt = s >> 1, half = s & 1;
        const int ch = mt * 32 + 8 * (2 * half + j / 4) + 4 * h + (j & 3);
        f[(size_t)G64_OFF_W4 + s * 512 + lane * 8 + j] = r < 9 ? f32_to_half_bits(W4[(size_t)ch * 9 + r], prec) : (uint16_t)0;
      }
  }
  g.fused = dw.upload_raw(f.data(), f.size() * sizeof(uint16_t));
  std::vector<float> bias(256, 0.f);
  for (int i = 0; i < 64; ++i) { bias[i] = B0[i]; bias[64 + i] = B2[i]; bias[128 + i] = Bx[i]; }
  bias[192] = b4;
  g.fused_bias = (float*)dw.upload_raw(bias.data(), bias.size() * sizeof(float));
}

template <class E>
void launch_topiq_gate64(const TensorT<E>& x, const TensorT<E>& out, const void* frag, const float* bias, int wblk_act, int gate_act, hipStream_t s) {
  FE_CHECK(x.c == 64 && out.c == 64 && x.h % 16 == 0 && x.w % 16 == 0 && out.h == x.h / 16 && out.w == x.w / 16 && out.n == x.n && out.ld == 64 && x.ld % 8 == 0,
           "gate64: %dx%dx%d -> %dx%dx%d is not the fused form's shape", x.h, x.w, x.c, out.h, out.w, out.c);
  Gate64Params<E> p{};
  p.x = x.p; p.ldx = x.ld; p.B = x.n; p.H = x.h; p.W = x.w;
  p.out = out.p; p.ldo = out.ld;
  p.frag = (const E*)frag; p.bias = bias;
  p.wblk_act = wblk_act; p.gate_act = gate_act;
  p.tiles_x = x.w / 16; p.tiles_y = x.h / 16; p.ntiles = x.n * p.tiles_x * p.tiles_y;
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
#ifdef G64_STAMPS
  static long long* stamps_dev = nullptr;
  if (!stamps_dev) FE_HIP(hipMalloc(&stamps_dev, 5 * sizeof(long long)));
  FE_HIP(hipMemsetAsync(stamps_dev, 0, 5 * sizeof(long long), s));
  p.stamps = stamps_dev;
#endif
  if (wblk_act == ACT_GELU && gate_act == ACT_GELU) {      // pyiqa's defaults
    static std::atomic<uint64_t> done{0};
    ensure_dynamic_lds((const void*)topiq_gate64_kernel<E, ACT_GELU, ACT_GELU>, G64_LDS, done);
    hipLaunchKernelGGL((topiq_gate64_kernel<E, ACT_GELU, ACT_GELU>), dim3(grid), dim3(256), G64_LDS, s, p);
  } else {
    static std::atomic<uint64_t> done{0};
    ensure_dynamic_lds((const void*)topiq_gate64_kernel<E, -1, -1>, G64_LDS, done);
    hipLaunchKernelGGL((topiq_gate64_kernel<E, -1, -1>), dim3(grid), dim3(256), G64_LDS, s, p);
  }
  FE_HIP(hipGetLastError());
#ifdef G64_STAMPS
  long long hst[5];
  FE_HIP(hipMemcpyAsync(hst, stamps_dev, sizeof hst, hipMemcpyDeviceToHost, s));
  FE_HIP(hipStreamSynchronize(s));
  const double per = (double)((p.ntiles + grid - 1) / grid);
  fprintf(stderr, "[gate64 stamps] %d windows, %d per workgroup; ticks per window (100 MHz): stage1 %.1f | stage2 loop %.1f | stage2 epilogue %.1f | stage3 %.1f | stage4 %.1f\n",
          p.ntiles, (int)per, hst[0] / per, hst[1] / per, hst[2] / per, hst[3] / per, hst[4] / per);
#endif
}
template void launch_topiq_gate64<bf16>(const TensorT<bf16>&, const TensorT<bf16>&, const void*, const float*, int, int, hipStream_t);
template void launch_topiq_gate64<f16>(const TensorT<f16>&, const TensorT<f16>&, const void*, const float*, int, int, hipStream_t);

}  // namespace fe

// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   Y[m][co] = act( (sum_k A[m][k] * Wt[co][k]) * scale[co] + shift[co] (+ res[m][co]) )
//   m = (n, oh, ow) output pixel,  k = (kh, kw, ci) with ci fastest (NHWC im2col, never materialised)
//
// One kernel covers every dense contraction of the hot path: ResNet-50 / ResNet-18 / U2-Net-P
// convolutions, the ViT patch-embed + QKV/proj/MLP GEMMs (a GEMM is a 1x1 "conv" over an [M][K]
// matrix) and the TOPIQ head. Replaces cuDNN/ATen conv2d + batch_norm + relu reached from
// reference models/pyiqa_scorer.py:212 (pyiqa -> timm resnet50) and models/samp_net.py:49-54,772.
//
// Tiling: 256 threads = 4 waves (64 lanes). Wave tile = (TM*32) x (TN*32) built from 32x32x2 fp32
// MFMAs; block tile BM x BN = (WGM*TM*32) x (WGN*TN*32); K staged through LDS in BK-wide slabs,
// double buffered, register-prefetched (global->VGPR issued before the MFMA phase, VGPR->LDS after it).
// LDS rows are k-contiguous with a +4 float pad so the ds_read_b128 fragment reads are conflict free
// (row*(BK+4) mod 64 hits 16 distinct 16-B slots per 16-lane group).
// The k order inside an 8-wide step is permuted (lane half h, element j <-> k = 8s + 4h + j) so each
// lane fetches its four A (and B) operands of a step with ONE 16-byte LDS read.
#include "fe_common.h"
#include <cstdlib>

namespace fe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act) { return fe_apply_act(v, act); }

template <int WGM, int WGN, int TM, int TN, int BK, bool FAST>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p, const int ntiles) {
  if (p.batch > 1) {
    const int b = blockIdx.y, bo = b / p.nb1, bi = b - bo * p.nb1;
    p.x += bo * p.xs2 + bi * p.xs1;
    p.w += bo * p.ws2 + bi * p.ws1;
    p.y += bo * p.ys2 + bi * p.ys1;
    if (p.shift) p.shift += bi * p.hs1;
  }
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int S = BK + 4;        // LDS row stride (floats)
  constexpr int CH = BK / 4;       // 16-B chunks per row
  constexpr int RPP = 256 / CH;    // rows covered per staging pass
  constexpr int AP = (BM + RPP - 1) / RPP, BP = (BN + RPP - 1) / RPP;
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][S]
  float* Bs = smem + 2 * BM * S;    // [2][BN][S]

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (private L2); give each XCD a contiguous
  // run of tiles so the N-tiles of one M-tile (same A panel) and neighbouring M-tiles (halo rows) hit L2.
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int mt = swz / ntiles, nt = swz - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  // ---- staging coordinates -----------------------------------------------------------
  const int chunk = t % CH, srow = t / CH;
  const int HoWo = p.Ho * p.Wo;
  int ih0[AP], iw0[AP];
  size_t abase[AP];
  bool mval[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int row = srow + i * RPP;
    const int m = m0 + row;
    mval[i] = (row < BM) && (m < p.M);
    const int mm = mval[i] ? m : 0;
    const int nimg = mm / HoWo;
    const int rem = mm - nimg * HoWo;
    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    ih0[i] = oh * p.sh - p.ph;
    iw0[i] = ow * p.sw - p.pw;
    abase[i] = (size_t)nimg * p.H * p.W;
  }
  // running (kh, kw, ci) of this thread's chunk (FAST: ci0 is block-uniform, chunk added at use)
  int kh, kw, ci;
  if (FAST) {
    kh = 0; kw = 0; ci = 0;
  } else {
    const int kk = chunk * 4;
    const int tap = kk / p.Cin;
    ci = kk - tap * p.Cin;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
  }
  const float* wrow[BP];
  bool nval[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int row = srow + i * RPP;
    const int n = n0 + row;
    nval[i] = (row < BN) && (n < p.Cout);
    wrow[i] = p.w + (size_t)(nval[i] ? n : 0) * p.ldw + chunk * 4;
  }

  float4 ra[AP], rb[BP];
  const int nk = p.Kp / BK;

  auto load_tile = [&](int kt) {
    const int cofs = FAST ? ci + chunk * 4 : ci;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int ih = ih0[i] + kh * p.dh, iw = iw0[i] + kw * p.dw;
      const bool ok = mval[i] && (kh < p.KH) && (!FAST || ci < p.Cin) && ((unsigned)ih < (unsigned)p.H) &&
                      ((unsigned)iw < (unsigned)p.W);
      if (ok) {
        const float* src = p.x + (abase[i] + (size_t)ih * p.W + iw) * p.ldx + cofs;
        ra[i] = *reinterpret_cast<const float4*>(src);
      } else {
        ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      if (nval[i]) rb[i] = *reinterpret_cast<const float4*>(wrow[i] + (size_t)kt * BK);
      else rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // advance k state to the next slab
    if (FAST) {
      // packed K order for Cin % 16 == 0: 16-channel block outer, tap inner (see pack_conv / kernels_conv_dma.hip)
      if (++kw == p.KW) { kw = 0; ++kh; }
      if (kh == p.KH) { kh = 0; kw = 0; ci += 16; }
    } else {
      ci += BK;
      while (ci >= p.Cin) {
        ci -= p.Cin;
        if (++kw == p.KW) { kw = 0; ++kh; }
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int row = srow + i * RPP;
      if (row < BM) *reinterpret_cast<float4*>(&As[(buf * BM + row) * S + chunk * 4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int row = srow + i * RPP;
      if (row < BN) *reinterpret_cast<float4*>(&Bs[(buf * BN + row) * S + chunk * 4]) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int arow = wm * TM * 32 + r, brow = wn * TN * 32 + r;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* Ab = As + (buf * BM + arow) * S + h * 4;
    const float* Bb = Bs + (buf * BN + brow) * S + h * 4;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * S + s * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * S + s * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  // C/D map of a 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  // Vector path: each wave transposes its accumulators through a private LDS region (the staging buffers
  // are dead after the last barrier) so that global traffic is 16 B per lane over whole 128/256-B row
  // segments: residual loads are issued 8 deep before any use, stores are full float4.
  const bool vec_ok = p.vec_epi;
  if (vec_ok) {
    constexpr int WC = TN * 32;          // wave tile width (floats)
    constexpr int ES = WC + 4;           // LDS row stride
    constexpr int LPR = WC / 4;          // lanes per row (16-B each)
    constexpr int RPI = 64 / LPR;        // rows per wave instruction
    constexpr int NIT = 32 / RPI;        // instructions per 32-row slab
    static_assert(4 * 32 * ES <= 2 * (BM + BN) * S, "epilogue staging must fit in the k-slab buffers");
    float* E = smem + wave * 32 * ES;
    const int lr = lane / LPR, lc = (lane % LPR) * 4;
    const int colb = n0 + wn * WC + lc;
    const bool cok = colb < p.Cout;      // Cout % 4 == 0 on this path
    const int colc = cok ? colb : 0;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + colc);
    if (p.shift) sf = *reinterpret_cast<const float4*>(p.shift + colc);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * h) * ES + j * 32 + r] = acc[i][j][e];
      const int mrow0 = m0 + wm * TM * 32 + i * 32 + lr;
      float4 rv[NIT], gv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = mrow0 + it * RPI;
        const int mc = m < p.M ? m : p.M - 1;
        if (p.res) rv[it] = *reinterpret_cast<const float4*>(p.res + (size_t)mc * p.ldr + colc);
        if (p.gate) {
          if (p.gate_c1) { const float g = p.gate[(size_t)mc * p.ldg]; gv[it] = make_float4(g, g, g, g); }
          else gv[it] = *reinterpret_cast<const float4*>(p.gate + (size_t)mc * p.ldg + colc);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = mrow0 + it * RPI;
        float4 v = *reinterpret_cast<const float4*>(&E[(lr + it * RPI) * ES + lc]);
        v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
        if (p.res && !p.res_after_act) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
        if (p.act == ACT_PRELU) {
          const float4 sl = *reinterpret_cast<const float4*>(p.slope + colc);
          v.x = v.x > 0.f ? v.x : v.x * sl.x; v.y = v.y > 0.f ? v.y : v.y * sl.y;
          v.z = v.z > 0.f ? v.z : v.z * sl.z; v.w = v.w > 0.f ? v.w : v.w * sl.w;
        } else {
          v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
        }
        if (p.res && p.res_after_act) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
        if (p.gate) { v.x *= gv[it].x; v.y *= gv[it].y; v.z *= gv[it].z; v.w *= gv[it].w; }
        if (cok && m < p.M) *reinterpret_cast<float4*>(p.y + (size_t)m * p.ldy + colb) = v;
      }
    }
    return;
  }
  // Scalar path (Cout or a channel stride not a multiple of 4): rare, small layers only.
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * TN * 32 + j * 32 + r;
    const bool cok = col < p.Cout;
    const float sc = (cok && p.scale) ? p.scale[col] : 1.f;
    const float sf = (cok && p.shift) ? p.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int m = m0 + row;
        if (cok && m < p.M) {
          float v = acc[i][j][e] * sc + sf;
          if (p.res && !p.res_after_act) v += p.res[(size_t)m * p.ldr + col];
          v = p.act == ACT_PRELU ? (v > 0.f ? v : v * p.slope[col]) : apply_act(v, p.act);
          if (p.res && p.res_after_act) v += p.res[(size_t)m * p.ldr + col];
          if (p.gate) v *= p.gate[(size_t)m * p.ldg + (p.gate_c1 ? 0 : col)];
          p.y[(size_t)m * p.ldy + col] = v;
        }
      }
    }
  }
}

template <int WGM, int WGN, int TM, int TN, int BK, bool FAST>
static void launch_variant(const ConvParams& p, hipStream_t s) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
  auto kern = conv_igemm_kernel<WGM, WGN, TM, TN, BK, FAST>;
  static std::atomic<uint64_t> lds_set{0};
  ensure_dynamic_lds((const void*)kern, lds, lds_set);
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles, p.batch > 1 ? p.batch : 1), dim3(256), lds, s, p, ntiles);
  FE_HIP(hipGetLastError());
}

// Forced tile variants for A/B timing (FAST path only).
static void launch_forced(const ConvParams& q, hipStream_t s) {
  FE_CHECK(q.Cin % 16 == 0, "forced variants need Cin %% 16 == 0");
  switch (q.variant) {
    case 1: launch_variant<2, 2, 2, 2, 16, true>(q, s); break;   // 128x128
    case 2: launch_variant<4, 1, 2, 2, 16, true>(q, s); break;   // 256x64
    case 3: launch_variant<4, 1, 2, 1, 16, true>(q, s); break;   // 256x32
    case 4: launch_variant<2, 2, 1, 1, 16, true>(q, s); break;   // 64x64
    case 6: launch_variant<1, 4, 2, 2, 16, true>(q, s); break;   // 64x256
    case 7: launch_variant<2, 2, 2, 1, 16, true>(q, s); break;   // 128x64
    default: FE_CHECK(false, "unknown conv variant %d", q.variant);
  }
}

double conv_flops(const ConvParams& p) { return 2.0 * (double)p.M * p.K * p.Cout; }

void launch_conv(const ConvParams& p, hipStream_t s) {
  FE_CHECK(p.x && p.w && p.y && p.ldy >= p.Cout && p.ldx >= p.Cin, "conv: null operand or row stride below the channel count");
  FE_CHECK(p.Cin % 4 == 0 && p.ldx % 4 == 0, "conv: Cin=%d ldx=%d must be multiples of 4", p.Cin, p.ldx);
  FE_CHECK(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0, "conv: x/w must be 16-B aligned");
  FE_CHECK(p.Kp % CONV_KALIGN == 0 && p.Kp >= p.K, "conv: bad Kp=%d K=%d", p.Kp, p.K);
  FE_CHECK(p.M > 0 && p.Cout > 0, "conv: empty problem");
  FE_CHECK((long long)p.N * p.H * p.W < (1ll << 31), "conv: too many input pixels");
  ConvParams q = p;
  if (q.ldw == 0) q.ldw = q.Kp;
  if (q.batch < 1) q.batch = 1;
  if (q.nb1 < 1) q.nb1 = 1;
  FE_CHECK(q.ldw % 4 == 0, "conv: ldw=%d must be a multiple of 4", q.ldw);
  {      // tensors past 4 GiB: image groups that fit the LDS-DMA kernels' buffer addressing (instead of the register-staged fallback)
    const unsigned long long xsp = (unsigned long long)p.N * p.H * p.W * (unsigned long long)p.ldx * 4;
    if (xsp >= 0xFFFFFF00ull && p.variant == 0 && conv_split_by_images(q, [&](const ConvParams& sub) { launch_conv(sub, s); })) return;
  }
  FE_CHECK(q.batch == 1 || (!q.res && !q.gate && !q.scale), "conv: batched launches take no res/gate/scale");
  auto al16 = [](const void* ptr) { return ((uintptr_t)ptr & 15) == 0; };
  q.vec_epi = (p.Cout % 4 == 0) && (p.ldy % 4 == 0) && al16(p.y) && (!p.res || (p.ldr % 4 == 0 && al16(p.res))) &&
              (!p.gate || p.gate_c1 || (p.ldg % 4 == 0 && al16(p.gate))) && (!p.scale || al16(p.scale)) &&
              (!p.shift || (al16(p.shift) && p.hs1 % 4 == 0)) && (p.batch <= 1 || (p.ys1 % 4 == 0 && p.ys2 % 4 == 0));
  const bool fast = (p.Cin % 16 == 0);
  // Tile choice: 128x128 for wide outputs, 256x64 / 256x32 for narrow ones; problems too small to give every
  // CU a 128x128 tile drop to 64x64 tiles (4x the workgroups).
  // forced variants (tests / tools): 1-7 register-staged tiles, 11-18 LDS-DMA tiles, 21/22 lean LDS-DMA tiles
  if (q.variant > 10 && q.variant < 40) { launch_conv_dma(q, q.variant > 20 ? q.variant : q.variant - 10, s); return; }
  if (q.variant > 0) { launch_forced(q, s); return; }
  // the narrow kernel reads weights in plain (tap, ci) order; pack_conv blocks K by 16 channels once Cin % 16 == 0 and there
  // is more than one tap, so those go to the matrix-core path below (256x32 tile)
  if (p.Cout <= 4 && q.batch <= 1 && !p.gate && q.variant == 0 && (p.Cin % 16 != 0 || p.KH * p.KW == 1)) { launch_conv_narrow(q, s); return; }
  static const bool no_dma = getenv("FE_NO_DMA") != nullptr;
  const unsigned long long xspan = ((unsigned long long)p.N * p.H * p.W - 1) * (unsigned long long)p.ldx * 4 + (unsigned long long)p.Cin * 4;
  const unsigned long long wspan = ((unsigned long long)p.Cout - 1) * (unsigned long long)q.ldw * 4 + (unsigned long long)p.Kp * 4;
  if (!no_dma && p.Cin % 16 == 0 && p.KH * p.KW < 64 && xspan < 0xFFFFFF00ull && wspan < 0xFFFFFF00ull) {
    // Tile choice = wave-quantisation model: workgroups are dealt round-robin over the 256 CUs and the ones resident on
    // a CU share its matrix pipes, so a launch lasts ~ ceil(WGs / 256) tiles per CU x (tile area / tile efficiency).
    // Measured relative efficiencies on the ResNet/ViT shapes (tools/conv_bench.py; re-scanned in round 2): 128x128 1.00, 128x64 1.05, 64x64 0.93;
    // short-K (bandwidth-bound) layers run best on the 128x64 tile (4 waves/SIMD).
    struct Cand { int tile, bm, bn; double eff; };
    static const double eff7 = getenv("FE_F32_EFF7") ? atof(getenv("FE_F32_EFF7")) : 1.05;   // A/B hook; 0.97 -> 1.05 measured +1 % on TOPIQ with the lean-loop kernels
    static const int k_short = getenv("FE_F32_KSHORT") ? atoi(getenv("FE_F32_KSHORT")) : 256;
    const Cand wide[3] = {{1, 128, 128, 1.00}, {7, 128, 64, eff7}, {4, 64, 64, 0.93}};
    static const Cand narrow[2] = {{7, 128, 64, 1.00}, {4, 64, 64, 0.95}};
    int tile = 3;
    if (p.Cout > 32) {
      const Cand* cs = p.Cout > 64 ? wide : narrow;
      const int nc = p.Cout > 64 ? 3 : 2;
      double best = 1e300;
      for (int i = 0; i < nc; ++i) {
        if (p.Cout > 64 && p.K <= k_short && cs[i].tile == 1) continue;
        const long long wgs = (long long)((p.M + cs[i].bm - 1) / cs[i].bm) * ((p.Cout + cs[i].bn - 1) / cs[i].bn) * q.batch;
        const double cost = (double)((wgs + 255) / 256) * cs[i].bm * cs[i].bn / cs[i].eff;
        if (cost < best) { best = cost; tile = cs[i].tile; }
      }
    }
    launch_conv_dma(q, tile, s);
    return;
  }
  const long long wg128 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * q.batch;
  if (p.Cout > 64) {
    if (wg128 < 384 && fast) launch_variant<2, 2, 1, 1, 16, true>(q, s);
    else if (fast) launch_variant<2, 2, 2, 2, 16, true>(q, s);
    else launch_variant<2, 2, 2, 2, 16, false>(q, s);
  } else if (p.Cout > 32) {
    if (wg128 < 192 && fast) launch_variant<2, 2, 1, 1, 16, true>(q, s);
    else if (fast) launch_variant<4, 1, 2, 2, 16, true>(q, s);
    else launch_variant<4, 1, 2, 2, 16, false>(q, s);
  } else {
    if (fast) launch_variant<4, 1, 2, 1, 16, true>(q, s);
    else launch_variant<4, 1, 2, 1, 16, false>(q, s);
  }
}

}  // namespace fe

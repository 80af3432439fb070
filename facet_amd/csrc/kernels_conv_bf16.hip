// 2-byte (bf16 / fp16) implicit-GEMM convolution: tile policy, launcher and the narrow-tile instantiations (kernel: conv_bf16_kernel.h).
// FE_E = element type of this translation unit: bf16 here, f16 through kernels_conv_f16.hip, which includes this file.
#include "conv_bf16_kernel.h"
#include <algorithm>
#ifndef FE_E
#define FE_E bf16
#endif

namespace fe {

typedef ConvParamsT<FE_E> ConvParamsE;

// The fp32-stream forms (p.res32 / p.y32): their own instantiations of a reduced tile set (no 16-channel-block units, no slim tiles:
// the layers that carry an fp32 stream are the ResNet block outputs / downsample branches and the ViT projections).
static void launch_bf16_tile_s32(const ConvParamsE& p, int tile, bool one_tap, hipStream_t s) {
  // reduced tile set: no slim tiles, no 16-channel-block units. The 256-row wave tiles exist in the GEMM form with ONE epilogue form
  // per kernel (with several fp32-stream forms inlined side by side they spill ~150 bytes per lane and ran slower than 128x128): fp32
  // rows out (+ fp32 residual, no activation: the projections that add into the token stream) on 256x128 / 256x256, split-pair rows
  // out on 256x128 (both 2-byte images staged at once: no room for that beside eight waves).
  static const bool wide = getenv("FE_S32_NO_WIDE") == nullptr;      // A/B hook
  const bool f32_form = p.y32 && !p.y && p.act == ACT_NONE && !p.res_after_act && p.split_lo_off <= 0;
  const bool pair_form = p.split_lo_off > 0;
  if ((tile == 8 || tile == 9) && !(wide && one_tap && (f32_form || pair_form))) tile = 1;
  if (tile == 9 && pair_form) tile = 8;      // (the pair form on the eight-wave 256x256 tile measured the same: 741 / 779 vs 740 / 793 TFLOP/s)
  if (tile == 3) tile = 7;
  if (tile == 5) tile = 4;
  if (one_tap) {
    switch (tile) {
      case 1: launch_bf16_variant<FE_E, 2, 2, 2, 2, 1, 0, true, 1>(p, s); return;
      case 8: if (pair_form) launch_bf16_variant<FE_E, 2, 2, 4, 2, 1, 0, true, 3>(p, s); else launch_bf16_variant<FE_E, 2, 2, 4, 2, 1, 0, true, 2>(p, s); return;
      case 9: launch_bf16_variant<FE_E, 2, 4, 4, 2, 1, 0, true, 2>(p, s); return;
      case 7: launch_bf16_variant<FE_E, 2, 2, 2, 1, 1, 0, true, 1>(p, s); return;
      case 4: launch_bf16_variant<FE_E, 2, 2, 1, 1, 1, 0, true, 1>(p, s); return;
      default: break;
    }
  }
  switch (tile) {
    case 1: launch_bf16_variant<FE_E, 2, 2, 2, 2, 1, 0, false, 1>(p, s); break;
    case 7: launch_bf16_variant<FE_E, 2, 2, 2, 1, 1, 0, false, 1>(p, s); break;
    case 4: launch_bf16_variant<FE_E, 2, 2, 1, 1, 1, 0, false, 1>(p, s); break;
    default: FE_CHECK(false, "conv_bf16: unknown tile %d", tile);
  }
}

template <int UNITS>
static void launch_bf16_tile(const ConvParamsE& p, int tile, bool one_tap, hipStream_t s) {
  FE_CHECK(p.act != ACT_PRELU, "conv_bf16: PReLU epilogue is not instantiated (no bf16 model uses it)");
  if constexpr (UNITS == 1) {
    if (one_tap) {      // 1x1 kernels with Cin % 64 == 0: K offset as the scalar offset of the loads, no tap masks
      switch (tile) {
        case 1: launch_bf16_variant<FE_E, 2, 2, 2, 2, 1, 0, true>(p, s); return;   // 128x128
        case 8: launch_bf16_variant<FE_E, 2, 2, 4, 2, 1, 0, true>(p, s); return;   // 256x128, waves of 128x64
        case 9: launch_bf16_variant<FE_E, 2, 4, 4, 2, 1, 0, true>(p, s); return;   // 256x256, eight waves of 128x64, one workgroup per CU
        case 7: launch_bf16_variant<FE_E, 2, 2, 2, 1, 1, 0, true>(p, s); return;   // 128x64
        case 4: launch_bf16_variant<FE_E, 2, 2, 1, 1, 1, 0, true>(p, s); return;   // 64x64
        default: break;
      }
    }
  }
  switch (tile) {
    case 1: launch_bf16_variant<FE_E, 2, 2, 2, 2, UNITS>(p, s); break;   // 128x128
    case 8: launch_bf16_variant<FE_E, 2, 2, 4, 2, UNITS>(p, s); break;   // 256x128
    case 7: launch_bf16_variant<FE_E, 2, 2, 2, 1, UNITS>(p, s); break;   // 128x64
    case 4: launch_bf16_variant<FE_E, 2, 2, 1, 1, UNITS>(p, s); break;   // 64x64
    case 3: launch_bf16_variant<FE_E, 4, 1, 2, 1, UNITS>(p, s); break;   // 256x32
    case 5: launch_bf16_variant<FE_E, 4, 1, 1, 1, UNITS>(p, s); break;   // 128x32
    default: FE_CHECK(false, "conv_bf16: unknown tile %d", tile);
  }
}

void launch_conv_bf16(const ConvParamsE& p0, hipStream_t s) {
  ConvParamsE p = p0;
  const int acols = p.a_wrap > 0 ? p.a_wrap * 32 : p.Cin;      // columns an A row really holds (split-operand GEMMs wrap: fe_common.h a_wrap)
  FE_CHECK(p.x && p.w && (p.y || p.y32) && (!p.y || p.ldy >= p.Cout) && p.ldx >= acols, "conv_bf16: null operand or row stride below the channel count");
  FE_CHECK(p.a_wrap <= 0 || (p.KH * p.KW == 1 && p.Cin % 64 == 0 && p.Kp == p.Cin && p.a_wrap * 32 < p.Cin && p.Cin <= 2 * p.a_wrap * 32),
           "conv_bf16: a_wrap needs the GEMM form (1x1, K %% 64 == 0) and K <= twice the wrapped width");
  const bool f32s = p.y32 || p.res32 || p.split_lo_off > 0;      // fp32 residual / output streams (FE_PRECISION_RES32 models), split-pair outputs
  FE_CHECK(p.split_lo_off <= 0 || (p.y && !p.y32 && !p.res32 && !p.res && !p.gate && p.split_lo_off % 8 == 0 && p.split_lo_off >= p.Cout && p.ldy >= p.split_lo_off + p.Cout),
           "conv_bf16: split-pair output needs y rows of at least split_lo_off + Cout columns and no residual / gate / fp32 output");
  FE_CHECK(!f32s || ((!p.y32 || p.ldy32 >= p.Cout) && (!p.res32 || p.ldr32 >= p.Cout) && !p.res && !p.pad_store && p.batch <= 1 && p.act != ACT_PRELU),
           "conv_bf16: fp32 streams take no 2-byte residual / pad_store / batched launch / PReLU");
  if (!p.y) p.ldy = 0;
  if (p.ldw == 0) p.ldw = p.Kp;
  if (p.batch < 1) p.batch = 1;
  if (p.nb1 < 1) p.nb1 = 1;
  const int ntaps = p.KH * p.KW;
  FE_CHECK(p.M > 0 && p.Cout > 0 && ntaps >= 1 && ntaps < 64, "conv_bf16: empty problem or too many taps");
  FE_CHECK(p.Kp % CONV_KALIGN_H == 0 && p.Kp >= p.K, "conv_bf16: bad Kp=%d K=%d", p.Kp, p.K);
  FE_CHECK(p.Cin % 8 == 0 && p.ldx % 8 == 0 && p.ldw % 8 == 0, "conv_bf16: Cin=%d ldx=%d ldw=%d must be multiples of 8", p.Cin, p.ldx, p.ldw);
  FE_CHECK(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0, "conv_bf16: x/w must be 16-B aligned");
  FE_CHECK(p.xs1 % 8 == 0 && p.xs2 % 8 == 0 && p.ws1 % 8 == 0 && p.ws2 % 8 == 0, "conv_bf16: batch strides must keep 16-B alignment");
  if (p.cb == 0) p.cb = 32;
  FE_CHECK(p.cb == 32 || p.cb == 16, "conv_bf16: channel block %d", p.cb);
  FE_CHECK(ntaps == 1 || p.Cin % p.cb == 0, "conv_bf16: spatial kernels need Cin %% %d == 0 (Cin=%d)", p.cb, p.Cin);
  FE_CHECK(p.batch == 1 || (!p.res && !p.gate && !p.scale), "conv_bf16: batched launches take no res/gate/scale");
  FE_CHECK((long long)p.N * p.H * p.W < (1ll << 31), "conv_bf16: too many input pixels");
  const unsigned long long xs = ((unsigned long long)p.N * p.H * p.W - 1) * (unsigned long long)p.ldx * 2 + (unsigned long long)acols * 2;
  const unsigned long long ws = ((unsigned long long)p.Cout - 1) * (unsigned long long)p.ldw * 2 + (unsigned long long)p.Kp * 2;
  {      // tensors past 4 GiB (input, or the output / residual the wide tiles address through buffers): image groups that fit
    unsigned long long ysp = (unsigned long long)p.M * (unsigned long long)(p.ldy > p.ldr ? p.ldy : p.ldr) * 2;
    if (f32s) ysp = std::max(ysp, (unsigned long long)p.M * (unsigned long long)(p.ldy32 > p.ldr32 ? p.ldy32 : p.ldr32) * 4);
    if ((xs >= 0xFFFFFF00ull || ysp >= 0xFFFFFF00ull) && conv_split_by_images(p, [&](const ConvParamsE& sub) { launch_conv_bf16(sub, s); })) return;
  }
  FE_CHECK(xs < 0xFFFFFF00ull && ws < 0xFFFFFF00ull, "conv_bf16: operand spans exceed 32-bit buffer addressing");
  p.x_span = (unsigned)xs; p.w_span = (unsigned)ws;
  p.unit_stride = (p.KH == 1 && p.KW == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.Ho == p.H && p.Wo == p.W &&
                   (long long)p.N * p.H * p.W == (long long)p.M) ? 1 : 0;
  auto al16 = [](const void* ptr) { return ((uintptr_t)ptr & 15) == 0; };
  const bool cout_ok = (p.Cout % 8 == 0) || (p.pad_store && ((p.Cout + 7) & ~7) <= p.ldy);
  p.vec_epi = cout_ok && (p.ldy % 8 == 0) && al16(p.y) && (!p.res || (p.ldr % 8 == 0 && al16(p.res))) &&
              (!p.y32 || (p.ldy32 % 4 == 0 && al16(p.y32) && p.Cout % 8 == 0)) && (!p.res32 || (p.ldr32 % 4 == 0 && al16(p.res32) && p.Cout % 8 == 0)) &&
              (!p.gate || p.gate_c1 || (p.ldg % 8 == 0 && al16(p.gate))) && (!p.scale || al16(p.scale)) &&
              (!p.shift || (al16(p.shift) && p.hs1 % 4 == 0)) && (p.batch <= 1 || (p.ys1 % 8 == 0 && p.ys2 % 8 == 0));
  FE_CHECK(!p.pad_store || p.vec_epi, "conv_bf16: pad_store needs the vector epilogue");
  // tile choice: same wave-quantisation model as the fp32 kernel (launch_conv in kernels_conv.hip)
  int tile = p.variant;
  if (tile == 0) {
    // Measured (tools/perf_clip.py / perf_topiq.py with FE_BF16_TILE): the 128x128 tile (three workgroups per CU) wins once the K loop
    // dominates the launch (K >= 2048: 713 vs 545 TFLOP/s on K = 4096); below that the 128x64 tile's four workgroups per CU hide
    // prologue and epilogue better (K = 1024: 560 vs 459); K <= 256 is bound by the output / residual streams either way.
    struct Cand { int tile, bm, bn; double eff; };
    // relative efficiencies measured on the ViT / ResNet shapes (FE_BF16_TILE sweeps, profiles/r02_README.md): long K loops want the
    // 256x128 tile (847 vs 693 vs 535 TFLOP/s at K = 4096), K = 1024 still prefers the large tiles (655-670 / 645 / 535), short K is
    // bound by its output streams and wants the four workgroups per CU of the 128x64 tile
    // K = 256 (FE_BF16_TILE sweep of TOPIQ after the wide tiles got their coalesced epilogue): 128x128 wins the small-M layers by
    // 8-13 %, 256x128 the large-M ones by 20-30 %, 128x64 keeps K = 64 / 128
    const double e8 = p.K >= 2048 ? 1.00 : (p.K >= 512 ? 1.00 : 0.97);
    const double e1 = p.K >= 2048 ? 0.82 : (p.K >= 512 ? 0.97 : 1.00);
    const double e7 = p.K >= 2048 ? 0.63 : (p.K >= 512 ? 0.81 : 0.88);
    // 256x256 (eight waves, one workgroup per CU, 7.8 KB of L2 -> LDS fill per MFLOP against 11.7 of the 256x128 tile): GEMM form only
    const bool gemm_form = ntaps == 1 && p.Cin % 64 == 0 && p.Kp == p.Cin && p.cb == 32;
    static const double e9_env = getenv("FE_BF16_E9") ? atof(getenv("FE_BF16_E9")) : 0.0;
    // measured on the ViT-L/14 GEMMs against the 256x128 tile: K = 4096 1042 vs 862 TFLOP/s, K = 1024 824 vs 768, K = 1024 with the GELU
    // epilogue 714 vs 711 (one workgroup per CU: nothing runs beside its epilogue), batched V^T (N = 257) 502 vs 622
    const double e9 = (gemm_form && p.K >= 512 && p.batch <= 1) ? (e9_env > 0 ? e9_env : (p.K >= 2048 ? 1.2 : (p.act == ACT_GELU ? 0.95 : 1.05))) : 0.0;
    const Cand wide[5] = {{8, 256, 128, e8}, {1, 128, 128, e1}, {7, 128, 64, e7}, {4, 64, 64, 0.7 * e7}, {9, 256, 256, e9}};
    static const Cand narrow[2] = {{7, 128, 64, 1.00}, {4, 64, 64, 0.90}};
    static const Cand slim[2] = {{3, 256, 32, 1.00}, {5, 128, 32, 0.92}};
    const Cand* cs = p.Cout > 64 ? wide : (p.Cout > 32 ? narrow : slim);
    const int nc = p.Cout > 64 ? 5 : 2;
    double best = 1e300;
    for (int i = 0; i < nc; ++i) {
      if (cs[i].eff <= 0.0) continue;
      if (p.Cout > 64 && p.K < 256 && (cs[i].tile == 1 || cs[i].tile == 8)) continue;   // short K: 128x64 (four workgroups per CU)
      const long long wgs = (long long)((p.M + cs[i].bm - 1) / cs[i].bm) * ((p.Cout + cs[i].bn - 1) / cs[i].bn) * p.batch;
      const double cost = (double)((wgs + 255) / 256) * cs[i].bm * cs[i].bn / cs[i].eff;
      if (cost < best) { best = cost; tile = cs[i].tile; }
    }
  }
  // KB (slabs per barrier) = 1: two slabs per barrier doubles the LDS ring (96 KB for the 128x128 tile, one workgroup per CU) and
  // measured 733 vs 1028 images/s on TOPIQ - occupancy beats barrier count here
  static const int force_tile = getenv("FE_BF16_TILE") ? atoi(getenv("FE_BF16_TILE")) : 0;   // A/B hook
  if (force_tile && p.variant == 0 && p.Cout > 32) tile = force_tile;
  if (tile == 9 && !(ntaps == 1 && p.Cin % 64 == 0 && p.Kp == p.Cin && p.cb == 32)) tile = 8;
  // the wide tiles' epilogue (h_epilogue_wide) addresses y / res / gate through 32-bit buffer offsets in 8-byte quads
  const int climit = p.pad_store ? ((p.Cout + 7) & ~7) : p.Cout;
  const unsigned long long ysp = ((unsigned long long)(p.M - 1) * p.ldy + climit + (p.split_lo_off > 0 ? p.split_lo_off : 0)) * 2;
  const unsigned long long rsp = p.res ? ((unsigned long long)(p.M - 1) * p.ldr + p.Cout) * 2 : 0;
  const unsigned long long gsp = p.gate ? ((unsigned long long)(p.M - 1) * p.ldg + (p.gate_c1 ? 1 : p.Cout)) * 2 : 0;
  const unsigned long long y32sp = p.y32 ? ((unsigned long long)(p.M - 1) * p.ldy32 + p.Cout) * 4 : 0;
  const unsigned long long r32sp = p.res32 ? ((unsigned long long)(p.M - 1) * p.ldr32 + p.Cout) * 4 : 0;
  // (a fp32 residual without an fp32 output has no wide form: h_epilogue_wide32 always writes p.y32)
  const bool wide_ok = p.vec_epi && ysp < 0xFFFFFF00ull && rsp < 0xFFFFFF00ull && gsp < 0xFFFFFF00ull && (p.pad_store ? (!p.scale && !p.shift && !p.res && !p.gate) : true) &&
                       y32sp < 0xFFFFFF00ull && r32sp < 0xFFFFFF00ull && (!f32s || ((p.y32 || p.split_lo_off > 0) && !p.gate));
  p.y_span = p.y ? (unsigned)ysp : 0u; p.r_span = (unsigned)rsp; p.g_span = (unsigned)gsp;
  p.y32_span = (unsigned)y32sp; p.r32_span = (unsigned)r32sp;
  if ((tile == 1 || tile == 8 || tile == 9) && !wide_ok) tile = 7;
  const bool one_tap = ntaps == 1 && p.Cin % 64 == 0 && p.Kp == p.Cin;
  if (f32s) {
    FE_CHECK(p.cb == 32, "conv_bf16: fp32 streams are instantiated for 32-channel blocks only (Cin %% 32 == 0)");
    launch_bf16_tile_s32(p, tile, one_tap, s);
  } else if (p.cb == 16) launch_bf16_tile<2>(p, tile, false, s);
  else launch_bf16_tile<1>(p, tile, one_tap, s);
}

}  // namespace fe

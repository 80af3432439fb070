// Implicit-GEMM convolution / GEMM on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the contraction
// kernel of the reduced-precision path (BASELINE.json configs[3]: TOPIQ + SAMP-Net + CLIP ViT-L/14 in bf16; the reference itself runs
// CLIP in half precision on a GPU, processing/scorer.py:513-516). bf16 NHWC activations and bf16 packed weights in HBM, fp32
// accumulators, fp32 per-channel scale / shift (folded BatchNorm, bias), fp32 activation math, one rounding to bf16 at the store.
//
//   Y[m][co] = act( (sum_k A[m][k] * Wt[co][k]) * scale[co] + shift[co] (+ res[m][co]) ) (* gate[m])
//
// Structure = conv_dma_kernel's (kernels_conv_dma.hip), re-cut for 2-byte elements:
//   * a K "slab" is 32 bf16 = 64 B per row, moved HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB = 16 rows per wave
//     instruction); lane-linear LDS image with the XOR swizzle applied on the SOURCE side and mirrored in the ds_read_b128 address;
//   * one ds_read_b128 = 8 bf16 = a lane's whole A (or B) fragment of one 32x32x16 MFMA (lane (r, h): k = 8h .. 8h+7): 2 MFMAs per
//     32x32 tile and slab; the slab's DMA pieces for three slabs ahead are issued between those MFMAs;
//   * 3-slab LDS ring, counted vmcnt, one raw s_barrier per slab, register double-buffered fragments (inline-asm ds_read so hipcc
//     does not drain the DMA ring in front of every LDS read);
//   * K order of the packed weights / the implicit im2col: channel block (cb = 32, or 16 when Cin % 32 != 0) outer, tap inner,
//     channel-in-block innermost. With cb = 16 a slab holds TWO (block, tap) units: lanes pick theirs by the half of the slab
//     their 16-B chunk lies in. 1x1 kernels (GEMMs) take any Cin % 8 == 0: the chunk past Cin is fetched as zeros;
//   * padding taps / rows past M / chunks past Cin = out-of-range buffer offsets (the hardware returns zeros);
//   * epilogue: narrow wave tiles (TN = 1) through a wave-private LDS transpose to 16-byte (8 x bf16) row stores; wide wave tiles
//     (TN = 2) swap the MFMA operands so the accumulators come out transposed and finish in registers (h_epilogue_wide).
// This header holds the kernel and its launcher template; the instantiations are spread over kernels_conv_bf16*.hip so the library
// builds in parallel (the wide tiles carry eleven epilogue forms each).
#pragma once
#include "fe_common.h"
#include <cstdlib>

namespace fe {

typedef float h_f32x16 __attribute__((ext_vector_type(16)));
typedef float h_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* h_lptr_t;

union H8 {            // 16 bytes: one DMA chunk / one MFMA fragment / one epilogue store
  h_v4f f;
  uint4 u;
};

// E below = the 2-byte element type of a launch: bf16 or f16 (fe_common.h: fe_mfma16 / fe_unpack2 / fe_pack2 overloads). Everything
// but the matrix instruction and the conversions at the edges of the epilogue is the same code.
template <class E>
__device__ __forceinline__ void h_unpack8(const uint4 u, float v[8]) {
  const E* tag = nullptr;
  fe_unpack2(tag, u.x, v[0], v[1]); fe_unpack2(tag, u.y, v[2], v[3]);
  fe_unpack2(tag, u.z, v[4], v[5]); fe_unpack2(tag, u.w, v[6], v[7]);
}


// Epilogue of the wide wave tiles (TN >= 2). Their MFMAs take the operands SWAPPED (weights as the row operand): the accumulator of a
// 32x32 tile then holds, in lane (r, h), pixel row r and the channels 8g + 4h + {0..3} (g = 0..3) - four CONSECUTIVE channels per
// register quad. Scale / shift / residual / activation / the rounding to bf16 all run on the registers, straight-line: every global
// access is a buffer load whose offset is pushed out of range for rows past M and columns past Cout (reads return zeros), so the
// math is ONE basic block and the compiler hoists its loads. The packed quads (8 bytes) then go through a wave-private LDS image of
// the wave tile, row-major, and leave as 16-byte stores, eight lanes per 128-byte row: written straight from the registers each
// store instruction would touch 32 rows with 16 bytes each, and the L2 takes a request per row segment, not per byte (measured on
// the ViT-L/14 GEMMs: 8-byte register stores 642-790 TFLOP/s, the same kernels with the stores dropped 836-1000).
// The transposing fp32 epilogue these tiles had before ran NIT dependent LDS round trips and residual fetches per 32 rows.
// ACTK: 0 none, 1 ReLU, 2 GELU, 3 sigmoid, -1 the activation named by p.act; GATE: multiplicative gate (TOPIQ GatedConv); the residual is added
// before the activation unless p.res_after_act. launch_conv_bf16 picks a wide tile only where the 16-byte vector layout is legal
// (p.vec_epi) and y / res / gate span less than 4 GB.
typedef unsigned h_v2u __attribute__((ext_vector_type(2)));
typedef unsigned h_v4u __attribute__((ext_vector_type(4)));
typedef float h_v4 __attribute__((ext_vector_type(4)));
template <class E, int TM, int TN, int ACTK, bool RES, bool GATE>
__device__ __forceinline__ void h_epilogue_wide(h_f32x16 (&acc)[TM][TN], const ConvParamsT<E>& p, const int row0, const int col0,
                                                const int lane, char* const stage) {
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int PITCH = TN * 64 + 16;          // bytes per staged row: TN * 32 bf16 + 16 (conflict-free 8-byte writes, 16-byte aligned reads)
  const int r = lane & 31, h = lane >> 5;
  const int mb = row0 + r;                     // the lane's pixel row of tile row i: mb + 32 i
  const int cb = col0 + 4 * h;                 // its first channel of tile column j, quad g: cb + 32 j + 8 g
  const int climit = p.pad_store ? ((p.Cout + 7) & ~7) : p.Cout;
  const bool hs = p.scale != nullptr, hb = p.shift != nullptr;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(RES ? p.res : p.y), 0, RES ? (int)p.r_span : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(GATE ? p.gate : p.y), 0, GATE ? (int)p.g_span : 0, 0x00020000);
  // a null scale / shift becomes an empty buffer: every read returns 0
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hs ? p.scale : p.shift), 0, hs ? p.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? p.shift : p.scale), 0, hb ? p.Cout * 4 : 0, 0x00020000);
  bool rok[TM];
  unsigned rbo[TM], gbo[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mb + 32 * i;
    rok[i] = m < p.M;
    rbo[i] = RES ? (unsigned)m * (unsigned)(p.ldr * 2) : 0u;
    gbo[i] = GATE ? (unsigned)m * (unsigned)(p.ldg * 2) : 0u;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    h_v4 sc[4], sf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {      // Cout % 8 == 0 wherever a scale or shift exists (p.vec_epi): a quad is wholly inside or outside
      const int c0 = cb + 32 * j + 8 * g;
      const unsigned co = c0 < p.Cout ? (unsigned)c0 * 4u : OOB;
      sc[g] = __builtin_bit_cast(h_v4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)co, 0, 0));
      sf[g] = __builtin_bit_cast(h_v4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)co, 0, 0));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = cb + 32 * j + 8 * g;
      h_v2u ru[TM], gu[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (RES) ru[i] = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)((rok[i] && c0 < p.Cout) ? rbo[i] + (unsigned)c0 * 2u : OOB), 0, 0);
        if (GATE) {
          if (p.gate_c1) gu[i].x = gu[i].y = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rg, (int)(rok[i] ? gbo[i] : OOB), 0, 0) * 0x10001u;
          else gu[i] = __builtin_amdgcn_raw_buffer_load_b64(rg, (int)((rok[i] && c0 < p.Cout) ? gbo[i] + (unsigned)c0 * 2u : OOB), 0, 0);
        }
      }
      float s4[4], b4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {        // no scale: 1 inside Cout, 0 past it (the zero columns of a pad_store row)
        s4[e] = hs ? sc[g][e] : (c0 + e < p.Cout ? 1.f : 0.f);
        b4[e] = sf[g][e];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[4], rf4[4] = {0.f, 0.f, 0.f, 0.f}, gf4[4] = {1.f, 1.f, 1.f, 1.f};
        if (RES) { fe_unpack2((const E*)nullptr, ru[i].x, rf4[0], rf4[1]); fe_unpack2((const E*)nullptr, ru[i].y, rf4[2], rf4[3]); }
        if (GATE) { fe_unpack2((const E*)nullptr, gu[i].x, gf4[0], gf4[1]); fe_unpack2((const E*)nullptr, gu[i].y, gf4[2], gf4[3]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][j][4 * g + e] * s4[e] + b4[e];
          const float rf = rf4[e];
          if (RES && !p.res_after_act) x += rf;
          if (ACTK == 1) x = x > 0.f ? x : 0.f;
          else if (ACTK == 2) x = fe_gelu_fast(x);
          else if (ACTK == 3) x = fe_rcp_fast(1.f + __expf(-x));
          else if (ACTK < 0) x = fe_apply_act_fast(x, p.act);
          if (RES && p.res_after_act) x += rf;
          if (GATE) x *= gf4[e];
          if (ACTK < 0 || ACTK == 3) x = (c0 + e < p.Cout) ? x : 0.f;      // sigmoid(0) != 0: keep the padded columns zero
          v[e] = x;
        }
        h_v2u o;
        o.x = fe_pack2((const E*)nullptr, v[0], v[1]); o.y = fe_pack2((const E*)nullptr, v[2], v[3]);
        *reinterpret_cast<h_v2u*>(stage + (32 * i + r) * PITCH + (32 * j + 8 * g + 4 * h) * 2) = o;
      }
    }
  }
  // the wave's tile, row-major in LDS -> 16 bytes per lane, LPR lanes per row
  constexpr int LPR = TN * 4, RPI = 64 / LPR;
  const int lr = lane / LPR, lc = lane % LPR;
  const int c = col0 + lc * 8;
#pragma unroll
  for (int it = 0; it < TM * 32 / RPI; ++it) {
    const int row = it * RPI + lr, m = row0 + row;
    const h_v4 d = *reinterpret_cast<const h_v4*>(stage + row * PITCH + lc * 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(h_v4u, d), ry, (int)((m < p.M && c < climit) ? (unsigned)m * (unsigned)(p.ldy * 2) + (unsigned)c * 2u : OOB), 0, 0);
  }
}

// The wide epilogue of the fp32-stream forms (FE_PRECISION_RES32: the residual / skip stream of the network stays fp32 around 2-byte
// GEMM operands). The residual is read as fp32 quads (p.res32; a null residual is an empty buffer - zeros - so one code path serves
// both), the result leaves as fp32 rows (p.y32) and, with OUT == 2, also as 2-byte rows (p.y: the operand of the next layer's
// GEMM; the ResNet skip stream needs both, the ViT token stream only fp32 - its 2-byte copy is written by the LayerNorm that
// follows). Same register layout and LDS staging as h_epilogue_wide: the 2-byte image of the whole wave tile first, then the fp32
// values one 32-column block at a time through the same wave-private region (144 bytes per row either way), 16-byte stores, eight
// lanes per 128-byte row segment. ACTK: 0 none, 1 ReLU, -1 named by p.act. No gate, no pad_store.
// OUT == 3: the split-pair output of the split-operand GEMMs (ClipModel::split3) - no fp32 rows; the result leaves as two 2-byte
// values per element, hi = round(v) at column c and lo = round(v - hi) at column p.split_lo_off + c of the same row of p.y.
// OUT == 4: the pair output with BOTH 2-byte images staged at once (stage2 = a second wave-private region): the accumulators die after
// the first pass - the form of the 256-row wave tiles, whose 128 accumulator registers leave no room to carry the low parts.
template <class E, int TM, int TN, int ACTK, int OUT>
__device__ __forceinline__ void h_epilogue_wide32(h_f32x16 (&acc)[TM][TN], const ConvParamsT<E>& p, const int row0, const int col0,
                                                  const int lane, char* const stage, char* const stage2 = nullptr) {
  static_assert(TN == 2, "wide tiles");
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int PITCH = TN * 64 + 16;
  const int r = lane & 31, h = lane >> 5;
  const int mb = row0 + r, cb = col0 + 4 * h;
  const bool hs = p.scale != nullptr, hb = p.shift != nullptr;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y : const_cast<E*>(p.x), 0, p.y ? (int)p.y_span : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry32 = __builtin_amdgcn_make_buffer_rsrc(p.y32, 0, (int)p.y32_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res32 ? p.res32 : p.y32), 0, p.res32 ? (int)p.r32_span : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hs ? p.scale : p.y32), 0, hs ? p.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? p.shift : p.y32), 0, hb ? p.Cout * 4 : 0, 0x00020000);
  bool rok[TM];
  unsigned rbo[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mb + 32 * i;
    rok[i] = m < p.M;
    rbo[i] = (unsigned)m * (unsigned)(p.ldr32 * 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = cb + 32 * j + 8 * g;
      const unsigned co = c0 < p.Cout ? (unsigned)c0 * 4u : OOB;
      const h_v4 sc = __builtin_bit_cast(h_v4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)co, 0, 0));
      const h_v4 sf = __builtin_bit_cast(h_v4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)co, 0, 0));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[4];
        const h_v4 rvi = __builtin_bit_cast(h_v4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)((rok[i] && c0 < p.Cout) ? rbo[i] + (unsigned)c0 * 4u : OOB), 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][j][4 * g + e] * (hs ? sc[e] : 1.f) + sf[e];
          if (!p.res_after_act) x += rvi[e];
          if (ACTK == 1) x = x > 0.f ? x : 0.f;
          else if (ACTK == -2) x = fe_apply_act_precise(x, p.act);      // erf GELU (A&S polynomial, 1.5e-7): the split-pair outputs are not rounded to one 2-byte value
          else if (ACTK < 0) x = p.exact_act ? fe_apply_act_precise(x, p.act) : fe_apply_act_fast(x, p.act);
          if (p.res_after_act) x += rvi[e];
          v[e] = x;
          if (OUT != 4) acc[i][j][4 * g + e] = OUT == 3 ? x - (float)(E)x : x;          // kept for the second pass below (OUT == 3: the low part)
        }
        if (OUT >= 2) {
          h_v2u o;
          o.x = fe_pack2((const E*)nullptr, v[0], v[1]); o.y = fe_pack2((const E*)nullptr, v[2], v[3]);
          *reinterpret_cast<h_v2u*>(stage + (32 * i + r) * PITCH + (32 * j + 8 * g + 4 * h) * 2) = o;
        }
        if (OUT == 4) {
          h_v2u o;
          o.x = fe_pack2((const E*)nullptr, v[0] - (float)(E)v[0], v[1] - (float)(E)v[1]); o.y = fe_pack2((const E*)nullptr, v[2] - (float)(E)v[2], v[3] - (float)(E)v[3]);
          *reinterpret_cast<h_v2u*>(stage2 + (32 * i + r) * PITCH + (32 * j + 8 * g + 4 * h) * 2) = o;
        }
      }
      // one (column block, quad) group at a time: hoisting the residual quads of all eight groups (4 VGPRs x TM each) beside the
      // accumulators spilled the 256-row tiles
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  auto store_e_image = [&](const unsigned col_off_bytes, const char* const img) __attribute__((always_inline)) {      // a staged 2-byte image of the wave tile -> 16-byte row stores
    constexpr int LPR = TN * 4, RPI = 64 / LPR;
    const int lr = lane / LPR, lc = lane % LPR;
    const int c = col0 + lc * 8;
#pragma unroll
    for (int it = 0; it < TM * 32 / RPI; ++it) {
      const int row = it * RPI + lr, m = row0 + row;
      const h_v4 d = *reinterpret_cast<const h_v4*>(img + row * PITCH + lc * 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(h_v4u, d), ry, (int)((m < p.M && c < p.Cout) ? (unsigned)m * (unsigned)(p.ldy * 2) + (unsigned)c * 2u + col_off_bytes : OOB), 0, 0);
      if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (OUT >= 2) store_e_image(0u, stage);
  if (OUT == 4) { store_e_image((unsigned)p.split_lo_off * 2u, stage2); return; }
  if (OUT == 3) {      // second image: the low parts, at column split_lo_off
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h_v2u o;
          o.x = fe_pack2((const E*)nullptr, acc[i][j][4 * g], acc[i][j][4 * g + 1]); o.y = fe_pack2((const E*)nullptr, acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          *reinterpret_cast<h_v2u*>(stage + (32 * i + r) * PITCH + (32 * j + 8 * g + 4 * h) * 2) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_e_image((unsigned)p.split_lo_off * 2u, stage);
    return;
  }
  // fp32 rows: one 32-column block of the wave tile at a time through the same staging region
  const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads of the previous pass have retired before its image is overwritten
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h_v4 q;
        q[0] = acc[i][j][4 * g]; q[1] = acc[i][j][4 * g + 1]; q[2] = acc[i][j][4 * g + 2]; q[3] = acc[i][j][4 * g + 3];
        *reinterpret_cast<h_v4*>(stage + (32 * i + r) * PITCH + (8 * g + 4 * h) * 4) = q;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int c = col0 + 32 * j + lc * 4;
#pragma unroll
    for (int it = 0; it < TM * 4; ++it) {
      const int row = it * 8 + lr, m = row0 + row;
      const h_v4 d = *reinterpret_cast<const h_v4*>(stage + row * PITCH + lc * 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(h_v4u, d), ry32, (int)((m < p.M && c < p.Cout) ? (unsigned)m * (unsigned)(p.ldy32 * 4) + (unsigned)c * 4u : OOB), 0, 0);
      if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four rows in flight at a time (the accumulators of the other column block are still live)
    }
  }
}

// UNITS = 32 / cb: (channel block, tap) units per 32-element slab. MODE 0 plain, 2 PReLU epilogue.
// ONE_TAP: 1x1 kernels (GEMMs) - no tap masks, the K offset of a slab goes into the scalar offset of the buffer load.
// S32: the fp32-stream form of the layer (p.res32 / p.y32, FE_PRECISION_RES32 models) - its own instantiations, so the plain kernels
// keep their register budgets (with the fp32 residual quads inlined beside the eleven plain forms the 256-row tiles spilled).
// S32 == 2 / 3: ONE form per kernel (2: fp32 rows out, optional fp32 residual, no activation; 3: split-pair rows out, activation named
// by p.act) - the 256-row wave tiles, which spill when several of these forms are inlined side by side.
template <class E, int WGM, int WGN, int TM, int TN, int UNITS, int MODE = 0, bool ONE_TAP = false, int S32 = 0>
__global__ __launch_bounds__(WGM * WGN * 64, (WGM * WGN == 8 ? 1 : (TM * TN >= 8 ? 2 : (TM * TN >= 4 ? (WGM * TM * 32 % 64 == 0 && WGN * TN * 32 % 64 == 0 ? 3 : 2) : 3)))) void conv_bf16_kernel(ConvParamsT<E> p, const int ntiles, const int ntotal) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int NW = WGM * WGN;                              // waves per workgroup: 4, or 8 for the 256x256 tile
  constexpr int AI = (BM + 16 * NW - 1) / (16 * NW), BI = (BN + 16 * NW - 1) / (16 * NW);    // DMA pieces per wave per slab (16 rows x 64 B each)
  constexpr int SLAB = (BM + BN) * 64;                       // bytes per slab
  static_assert(NW == 4 || (NW == 8 && TN > 1), "4 waves; 8 for wide wave tiles");
  extern __shared__ __attribute__((aligned(16))) char smem_h[];

  if (p.batch > 1) {
    const int b = blockIdx.y, bo = b / p.nb1, bi = b - bo * p.nb1;
    p.x += bo * p.xs2 + bi * p.xs1;
    p.w += bo * p.ws2 + bi * p.ws1;
    p.y += bo * p.ys2 + bi * p.ys1;
    if (p.shift) p.shift += bi * p.hs1;
  }
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA source coordinates (per lane, fixed for the K loop of one tile) -------------------------------
  const int rsub = lane >> 2, slot = lane & 3;
  const int chunk = slot ^ ((rsub >> 2) & 3);            // source chunk (8 elements) of the slab after the swizzle
  const int upar = UNITS == 2 ? (chunk >> 1) : 0;        // which unit of the slab this lane's chunk belongs to
  const int cofs = UNITS == 2 ? (chunk & 1) * 8 : chunk * 8;   // element offset of the chunk inside its unit
  const int HoWo = p.Ho * p.Wo;
  const int ntaps = p.KH * p.KW;
  int m0 = 0, n0 = 0;
  unsigned aoffs[AI];          // byte offset of the row's (kh=0,kw=0,ci=cofs) element from p.x (buffer addressing)
  unsigned long long amask[AI];
  unsigned boffs[BI];
  // tile v of the launch's ntotal = mtiles * ntiles: XCD-aware bijection (workgroup ids go round-robin over the 8 XCDs, each XCD gets a
  // contiguous run of tiles, n fastest, so the tiles sharing an A row block meet in one L2). The persistent wide tiles walk
  // v = blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0 whenever a workgroup takes more than one tile: it stays on its XCD's run).
  auto setup_tile = [&](const int v) __attribute__((always_inline)) {
    const int q8 = ntotal >> 3, rr = ntotal & 7, xcd = v & 7;
    const int swz = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (v >> 3);
    const int mt = swz / ntiles, nt = swz - mt * ntiles;
    m0 = mt * BM; n0 = nt * BN;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int row = 16 * (NW * j + wave) + rsub;
      const int m = m0 + row;
      const bool valid = (row < BM) && (m < p.M);
      const int mm = valid ? m : 0;
      if (ONE_TAP && p.unit_stride) {      // 1x1, stride 1, no padding: output pixel m IS input pixel m (no divisions in the prologue)
        aoffs[j] = (unsigned)(((long long)mm * p.ldx + cofs) * 2);
        amask[j] = valid ? 1ull : 0ull;
        continue;
      }
      const int nimg = mm / HoWo;
      const int rem = mm - nimg * HoWo;
      const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
      const int ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
      const long long pix = ((long long)nimg * p.H + ih0) * p.W + iw0;
      aoffs[j] = (unsigned)((pix * p.ldx + cofs) * 2);
      unsigned long long mk = 0;
      if (valid) {
        for (int tp = 0; tp < ntaps; ++tp) {
          const int kh = tp / p.KW, kw = tp - kh * p.KW;
          const int ih = ih0 + kh * p.dh, iw = iw0 + kw * p.dw;
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) mk |= 1ull << tp;
        }
      }
      amask[j] = mk;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int row = 16 * (NW * j + wave) + rsub;
      int n = n0 + row;
      if (n > p.Cout - 1) n = p.Cout - 1;   // columns past Cout are computed on a valid row and discarded
      boffs[j] = (unsigned)(((size_t)n * p.ldw + chunk * 8) * 2);
    }
  };
  setup_tile(blockIdx.x);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(p.x), 0, (int)p.x_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(p.w), 0, (int)p.w_span, 0x00020000);

  // block-uniform running state of the next unit(s) to fetch: unit u -> channel block u / ntaps, tap u % ntaps
  const int CB = 32 / UNITS;
  int tap0 = 0, kh0 = 0, kw0 = 0, ci0 = 0;                       // unit 0 of the next slab
  int tap1 = 0, kh1 = 0, kw1 = 0, ci1 = 0;                       // unit 1 (UNITS == 2)
  auto reset_units = [&]() __attribute__((always_inline)) {                                     // K position 0 (start of a tile)
    tap0 = kh0 = kw0 = ci0 = 0;
    tap1 = kh1 = kw1 = ci1 = 0;
    if (UNITS == 2) {
      tap1 = 1; kw1 = 1;
      if (kw1 == p.KW) { kw1 = 0; kh1 = 1; }
      if (tap1 == ntaps) { tap1 = 0; kh1 = 0; kw1 = 0; ci1 = CB; }
    }
  };
  reset_units();
  auto advance = [&](int& tap, int& kh, int& kw, int& ci) {      // + UNITS units
#pragma unroll
    for (int s = 0; s < UNITS; ++s) {
      ++tap;
      if (++kw == p.KW) { kw = 0; ++kh; }
      if (tap == ntaps) { tap = 0; kh = 0; kw = 0; ci += CB; }
    }
  };
  const int nsub = p.Kp / 32;            // slabs; Kp % 64 == 0

  h_f32x16 acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();

  // fragment read offsets (bytes): row*64 + ((2s+h) ^ sw)*16, sw = (row>>2)&3 = (r>>2)&3; s = which 16-deep MFMA of the slab
  const int sw = (r >> 2) & 3;
  const int fo0 = ((h ^ sw) << 4), fo1 = fo0 ^ 32;
  const int aoff = (wm * TM * 32 + r) * 64, boff = BM * 64 + (wn * TN * 32 + r) * 64;

  const unsigned lds_base = (unsigned)(size_t)(h_lptr_t)smem_h;
  h_v4f fa[2][2 * TM], fb[2][2 * TN];   // [register set][fragment]; indices are compile-time everywhere below
  // The main loop. Its first form spent ~125 instructions per slab and wave (scalar branches around every piece, a switch for the
  // counted wait, modulo-3 ring arithmetic, one VALU address per LDS read) on 4-8 MFMAs of 32 cycles - with four waves per SIMD the
  // instruction issue, not the matrix pipe, set the pace (PMC: SQ_ACTIVE_INST_ANY 0.35 of the wave cycles). This form is unrolled by 6 = ring slot (mod 3) x register set (mod 2),
  // so every LDS read is `base + immediate` and every DMA destination `base + immediate`; the counted wait is an immediate; the last
  // three slabs still "issue" their pieces, with out-of-range offsets (zero fill, no memory traffic), so every iteration is the
  // same straight-line code; 1x1 kernels pass the slab's K offset as the scalar offset of the load (no VALU at all per piece).
  // wide wave tiles: the register-resident epilogue on the transposed accumulators, specialised on the common activation / residual forms
  auto wide_epilogue = [&](const int row0, const int col0) __attribute__((always_inline)) {
    if constexpr (TN > 1) {
      char* const stage = smem_h + wave * (TM * 32) * (TN * 64 + 16);       // wave-private image of its TM*32 x TN*32 tile
      // straight-line forms of the combinations the models use; the rest (softplus gates, residual after the activation) take the
      // form that reads activation, gate and residual order from the parameters
      const bool plain = !p.gate && !(p.res && p.res_after_act);
      if constexpr (S32 == 2) {
        h_epilogue_wide32<E, TM, TN, 0, 1>(acc, p, row0, col0, lane, stage);
        return;
      } else if constexpr (S32 == 3) {
#ifdef FE_PAIR_DUAL_STAGE
        h_epilogue_wide32<E, TM, TN, -2, 4>(acc, p, row0, col0, lane, stage, stage + NW * (TM * 32) * (TN * 64 + 16));
#else
        // one staged image at a time (the low parts wait in the accumulator registers): half the LDS of the dual-stage form, so two
        // workgroups still fit a CU - measured against it on the split-operand ViT GEMMs (profiles/r03_README.md)
        h_epilogue_wide32<E, TM, TN, -2, 3>(acc, p, row0, col0, lane, stage);
#endif
        return;
      } else if constexpr (S32 != 0) {      // fp32-stream forms (launch_conv_bf16 admits them only without gate / E-typed residual / pad_store)
        if (p.split_lo_off > 0) h_epilogue_wide32<E, TM, TN, -1, 3>(acc, p, row0, col0, lane, stage);                                  // split-pair output
        else if (!p.y && p.act == ACT_NONE && !p.res_after_act) h_epilogue_wide32<E, TM, TN, 0, 1>(acc, p, row0, col0, lane, stage);        // ViT projections
        else if (p.y && p.act == ACT_RELU && !p.res_after_act) h_epilogue_wide32<E, TM, TN, 1, 2>(acc, p, row0, col0, lane, stage);    // ResNet block outputs
        else h_epilogue_wide32<E, TM, TN, -1, 2>(acc, p, row0, col0, lane, stage);      // anything else (a null p.y is an empty buffer: stores dropped)
        return;
      }
      if (plain && p.act == ACT_NONE) {
        if (p.res) h_epilogue_wide<E, TM, TN, 0, true, false>(acc, p, row0, col0, lane, stage); else h_epilogue_wide<E, TM, TN, 0, false, false>(acc, p, row0, col0, lane, stage);
      } else if (plain && p.act == ACT_RELU) {
        if (p.res) h_epilogue_wide<E, TM, TN, 1, true, false>(acc, p, row0, col0, lane, stage); else h_epilogue_wide<E, TM, TN, 1, false, false>(acc, p, row0, col0, lane, stage);
      } else if (plain && p.act == ACT_GELU) {
        if (p.res) h_epilogue_wide<E, TM, TN, 2, true, false>(acc, p, row0, col0, lane, stage); else h_epilogue_wide<E, TM, TN, 2, false, false>(acc, p, row0, col0, lane, stage);
      } else if (plain && p.act == ACT_SIGMOID && !p.res) {
        h_epilogue_wide<E, TM, TN, 3, false, false>(acc, p, row0, col0, lane, stage);
      } else if (p.gate && !p.res && p.act == ACT_GELU) {
        h_epilogue_wide<E, TM, TN, 2, false, true>(acc, p, row0, col0, lane, stage);       // TOPIQ GatedConv, default gate activation
      } else if (p.gate) {
        if (p.res) h_epilogue_wide<E, TM, TN, -1, true, true>(acc, p, row0, col0, lane, stage); else h_epilogue_wide<E, TM, TN, -1, false, true>(acc, p, row0, col0, lane, stage);
      } else {
        if (p.res) h_epilogue_wide<E, TM, TN, -1, true, false>(acc, p, row0, col0, lane, stage); else h_epilogue_wide<E, TM, TN, -1, false, false>(acc, p, row0, col0, lane, stage);
      }
    }
  };
  // (32-column tiles: only waves 0 and 1 own a B piece; the other two issue an out-of-range dummy piece into a scratch KiB behind the
  // ring, so every wave still counts the same number of pieces per slab)
  constexpr bool LEAN = (BM % (16 * NW) == 0) && (BN % (16 * NW) == 0 || (BN == 32 && NW == 4));
  static_assert(TN == 1 || (LEAN && MODE == 0), "wide tiles: lean loop, no PReLU epilogue");
  // 64x64 wave tiles keep ONE fragment set (read after the MFMAs of the slab before, exposed LDS latency covered by the other waves):
  // 32 VGPRs fewer = three workgroups per CU instead of two, i.e. 144 KB instead of 96 KB of the 160 KB LDS holding DMA data in flight
  constexpr bool DBUF = (TM * TN < 4);
  static_assert(LEAN, "every instantiated tile has a uniform piece count per wave");
  {      // (scope of the main-loop macros' locals)
    constexpr int NPW = AI + BI;                          // pieces per wave and slab, the same for every wave
    const unsigned bA0 = lds_base + (unsigned)(aoff + fo0), bA1 = lds_base + (unsigned)(aoff + fo1);
    const unsigned bB0 = lds_base + (unsigned)(boff + fo0), bB1 = lds_base + (unsigned)(boff + fo1);
    const unsigned bA0h = bA0 + 2 * SLAB, bA1h = bA1 + 2 * SLAB, bB0h = bB0 + 2 * SLAB, bB1h = bB1 + 2 * SLAB;
    (void)bA0h; (void)bA1h; (void)bB0h; (void)bB1h;
    char* const dA = smem_h + 1024 * wave;                // + SLOT * SLAB + 1024 * NW * j
    char* const dB = smem_h + BM * 64 + 1024 * wave;
    unsigned aoffs_l[AI];
#pragma unroll
    for (int j = 0; j < AI; ++j) aoffs_l[j] = (ONE_TAP && !(amask[j] & 1ull)) ? 0xFFFFFFF0u : aoffs[j];   // 1x1: row validity folded in
    int g3 = 0;                                           // slab being issued
    int l_tb = 0, l_tap = 0; bool l_cok = true;
    auto lean_begin = [&]() {                             // spatial kernels: tap state of slab g3
      if constexpr (!ONE_TAP) {
        const int tb0 = ((kh0 * p.dh * p.W + kw0 * p.dw) * p.ldx + ci0) * 2;
        int cil = ci0;
        l_tb = tb0; l_tap = tap0;
        if (UNITS == 2) {
          const int tb1 = ((kh1 * p.dh * p.W + kw1 * p.dw) * p.ldx + ci1) * 2;
          l_tb = upar ? tb1 : tb0; l_tap = upar ? tap1 : tap0; cil = upar ? ci1 : ci0;
        }
        l_cok = (cil + cofs) < p.Cin;
      }
    };
    // a_wrap (GEMM form only): the A operand restarts from column 0 once, at slab a_wrap - the split-operand products
    // [xh | xl | xh] . [Wh | Wh | Wl]^T read their activation row as [xh | xl] and wrap (ClipModel split3, model_clip.hip)
    const int a_wrap = (ONE_TAP && p.a_wrap > 0) ? p.a_wrap : 0x3fffffff;
    int ga3 = 0;                                          // slab of A being issued (== g3 until the wrap)
    auto lean_end = [&]() {
      if constexpr (!ONE_TAP) {
        advance(tap0, kh0, kw0, ci0);
        if (UNITS == 2) advance(tap1, kh1, kw1, ci1);
      }
      ++g3;
      ga3 = g3 >= a_wrap ? g3 - a_wrap : g3;
    };
#define FL_PIECE(SLOT, Q)                                                                                               \
    {                                                                                                                   \
      const bool live_ = g3 < nsub;                                                                                     \
      if constexpr ((Q) < AI) {                                                                                         \
        if constexpr (ONE_TAP) {                                                                                        \
          const unsigned off_ = live_ ? aoffs_l[(Q) < AI ? (Q) : 0] : 0xFFFFFFF0u;                      \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (h_lptr_t)(dA + (SLOT) * SLAB + 1024 * NW * (Q)), 16, (int)off_, ga3 * 64, 0, 0);  \
        } else {                                                                                                        \
          const bool ok_ = ((amask[(Q) < AI ? (Q) : 0] >> l_tap) & 1ull) && l_cok && live_;                              \
          const unsigned off_ = ok_ ? aoffs[(Q) < AI ? (Q) : 0] + (unsigned)l_tb : 0xFFFFFFF0u;                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (h_lptr_t)(dA + (SLOT) * SLAB + 1024 * NW * (Q)), 16, (int)off_, 0, 0, 0);         \
        }                                                                                                               \
      } else {                                                                                                          \
        const bool mine_ = BN >= 16 * NW || 16 * wave < BN;                                                             \
        const unsigned off_ = (live_ && mine_) ? boffs[(Q) >= AI ? (Q) - AI : 0] : 0xFFFFFFF0u;                         \
        char* const dst_ = mine_ ? dB + (SLOT) * SLAB + 1024 * NW * ((Q) - AI) : smem_h + 3 * SLAB + 1024 * wave;       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (h_lptr_t)dst_, 16, (int)off_, g3 * 64, 0, 0);                    \
      }                                                                                                                 \
    }
    // ds_read offsets are 16-bit: ring slot 2 of the 256x256 tile (2 x 32 KB in) is read through a second base register
#define FL_READ1(DST, BASE, IMM)                                                                                        \
      if constexpr ((IMM) <= 65535) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE), "i"(IMM)); }  \
      else { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE##h), "i"((IMM) - 2 * SLAB)); }
#define FL_READ_FRAGS(SET, SLOT)                                                                                        \
    {                                                                                                                   \
      FL_READ1(fa[SET][0], bA0, (SLOT) * SLAB) FL_READ1(fa[SET][1], bA1, (SLOT) * SLAB)                                 \
      if constexpr (TM > 1) { FL_READ1(fa[SET][2], bA0, (SLOT) * SLAB + 2048) FL_READ1(fa[SET][3], bA1, (SLOT) * SLAB + 2048) } \
      if constexpr (TM > 2) { FL_READ1(fa[SET][4], bA0, (SLOT) * SLAB + 4096) FL_READ1(fa[SET][5], bA1, (SLOT) * SLAB + 4096)   \
                              FL_READ1(fa[SET][6], bA0, (SLOT) * SLAB + 6144) FL_READ1(fa[SET][7], bA1, (SLOT) * SLAB + 6144) } \
      FL_READ1(fb[SET][0], bB0, (SLOT) * SLAB) FL_READ1(fb[SET][1], bB1, (SLOT) * SLAB)                                 \
      if constexpr (TN > 1) { FL_READ1(fb[SET][2], bB0, (SLOT) * SLAB + 2048) FL_READ1(fb[SET][3], bB1, (SLOT) * SLAB + 2048) } \
    }
#define FL_MFMA(SET, Q)                                                                                                 \
    {                                                                                                                   \
      constexpr int hh_ = (Q) / (TM * TN), i_ = ((Q) / TN) % TM, j_ = (Q) % TN;                                         \
      H8 a_, b_;                                                                                                        \
      a_.f = fa[SET][2 * i_ + hh_]; b_.f = fb[SET][2 * j_ + hh_];                                                       \
      if constexpr (TN > 1) acc[i_][j_] = fe_mfma16((const E*)nullptr, b_.f, a_.f, acc[i_][j_]);  /* transposed tile */      \
      else acc[i_][j_] = fe_mfma16((const E*)nullptr, a_.f, b_.f, acc[i_][j_]);                                           \
    }
#define FL_PIECE_AT(SLOT, N, Q)                                                                                          \
    if constexpr ((N) < NPW && (((N) + 1) * (2 * TM * TN) / (NPW + 1) - 1 < 0 ? 0 : ((N) + 1) * (2 * TM * TN) / (NPW + 1) - 1) == (Q)) { \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      FL_PIECE(SLOT, N)                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
#define FL_STEP_Q(CUR, SLOTI, Q)                                                                                         \
    if constexpr ((Q) < 2 * TM * TN) {                                                                                   \
      FL_MFMA(CUR, Q)                                                                                                    \
      FL_PIECE_AT(SLOTI, 0, Q) FL_PIECE_AT(SLOTI, 1, Q) FL_PIECE_AT(SLOTI, 2, Q) FL_PIECE_AT(SLOTI, 3, Q) FL_PIECE_AT(SLOTI, 4, Q) FL_PIECE_AT(SLOTI, 5, Q) \
    }
    // slab g (fragments in set CUR): wait for this wave's pieces of slab g+1, publish it, read its fragments into the other set,
    // MFMAs of slab g with the pieces of slab g+3 (ring slot SLOTI = g % 3) between them
#define FL_SLAB(CUR_, NXT_, SLOTR, SLOTI)                                                                                \
    {                                                                                                                    \
      constexpr int CUR = DBUF ? (CUR_) : 0, NXT = DBUF ? (NXT_) : 0;                                                    \
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPW) : "memory");                                                         \
      __builtin_amdgcn_s_barrier();                                                                                      \
      lean_begin();                                                                                                      \
      if constexpr (DBUF) { FL_READ_FRAGS(NXT, SLOTR) }                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      FL_STEP_Q(CUR, SLOTI, 0) FL_STEP_Q(CUR, SLOTI, 1) FL_STEP_Q(CUR, SLOTI, 2) FL_STEP_Q(CUR, SLOTI, 3)                \
      FL_STEP_Q(CUR, SLOTI, 4) FL_STEP_Q(CUR, SLOTI, 5) FL_STEP_Q(CUR, SLOTI, 6) FL_STEP_Q(CUR, SLOTI, 7)                \
      FL_STEP_Q(CUR, SLOTI, 8) FL_STEP_Q(CUR, SLOTI, 9) FL_STEP_Q(CUR, SLOTI, 10) FL_STEP_Q(CUR, SLOTI, 11)              \
      FL_STEP_Q(CUR, SLOTI, 12) FL_STEP_Q(CUR, SLOTI, 13) FL_STEP_Q(CUR, SLOTI, 14) FL_STEP_Q(CUR, SLOTI, 15)            \
      lean_end();                                                                                                        \
      if constexpr (!DBUF) { __builtin_amdgcn_sched_barrier(0); FL_READ_FRAGS(NXT, SLOTR) }                              \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
    static_assert(NPW <= 6 && 2 * TM * TN <= 16 && TM <= 4 && TN <= 2, "lean loop: piece / MFMA slots");
    // prologue: slabs 0, 1, 2 (slab 2 is a dummy when nsub == 2)
#define FL_ISSUE_ALL(SLOT) { lean_begin(); FL_PIECE_ALL(SLOT) lean_end(); }
#define FL_PIECE_N(SLOT, N) if constexpr ((N) < NPW) FL_PIECE(SLOT, N)
#define FL_PIECE_ALL(SLOT) FL_PIECE_N(SLOT, 0) FL_PIECE_N(SLOT, 1) FL_PIECE_N(SLOT, 2) FL_PIECE_N(SLOT, 3) FL_PIECE_N(SLOT, 4) FL_PIECE_N(SLOT, 5)
    FL_ISSUE_ALL(0) FL_ISSUE_ALL(1) FL_ISSUE_ALL(2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NPW) : "memory");
    __builtin_amdgcn_s_barrier();
    FL_READ_FRAGS(0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int g = 0; g < nsub; g += 6) {      // nsub is even
      FL_SLAB(0, 1, 1, 0)
      FL_SLAB(1, 0, 2, 1)
      if (g + 2 < nsub) {
        FL_SLAB(0, 1, 0, 2)
        FL_SLAB(1, 0, 1, 0)
      }
      if (g + 4 < nsub) {
        FL_SLAB(0, 1, 2, 1)
        FL_SLAB(1, 0, 0, 2)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy pieces of the tail have landed before the epilogue reuses the ring
#undef FL_SLAB
#undef FL_STEP_Q
#undef FL_PIECE_AT
#undef FL_MFMA
#undef FL_READ_FRAGS
#undef FL_READ1
#undef FL_PIECE
#undef FL_ISSUE_ALL
#undef FL_PIECE_N
#undef FL_PIECE_ALL
  }
  __syncthreads();   // all fragment reads retired before the epilogue reuses the ring as staging
  if constexpr (TN > 1) {
    wide_epilogue(m0 + wm * TM * 32, n0 + wn * TN * 32);
    return;
  }

  // Two epilogue forms (as in kernels_conv_dma.hip). Narrow wave tiles (TN = 1: the tiles of the HBM-bound short-K layers) keep the
  // fully unrolled row code with every residual row requested up front (K = 64 -> 256 expand: 4.2 vs 3.8 TB/s against the rolled
  // form: hipcc drains the outstanding loads at the scalar branches the rolled form has per row). Wide wave tiles take the compact
  // rolled form.
  if constexpr (TN == 1) {
  // ---- epilogue: transpose through a wave-private LDS region, 8 bf16 (16 B) per lane and store ------------------
    float* smem = reinterpret_cast<float*>(smem_h);
    if (p.vec_epi) {
      constexpr int WC = TN * 32, ES = WC + 4, LPR = WC / 8, RPI = 64 / LPR, NIT = 32 / RPI;
      float* Et = smem + wave * 32 * ES;
      const int lr = lane / LPR, lc = (lane % LPR) * 8;
      const int colb = n0 + wn * WC + lc;
      const int climit = p.pad_store ? ((p.Cout + 7) & ~7) : p.Cout;
      const bool cok = colb < climit;
      const int colc = (colb + 8 <= p.Cout) ? colb : 0;     // per-channel vectors are only read for fully valid groups
      const bool cfull = colb + 8 <= p.Cout;
      float sc[8], sf[8];
  #pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sf[e] = 0.f; }
      if (p.scale && cfull) {
        const float4 a = *reinterpret_cast<const float4*>(p.scale + colc), b = *reinterpret_cast<const float4*>(p.scale + colc + 4);
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
      }
      if (p.shift && cfull) {
        const float4 a = *reinterpret_cast<const float4*>(p.shift + colc), b = *reinterpret_cast<const float4*>(p.shift + colc + 4);
        sf[0] = a.x; sf[1] = a.y; sf[2] = a.z; sf[3] = a.w; sf[4] = b.x; sf[5] = b.y; sf[6] = b.z; sf[7] = b.w;
      }
      if (!cfull && cok) {   // ragged last group (pad_store): scalar reads of what exists
  #pragma unroll
        for (int e = 0; e < 8; ++e)
          if (colb + e < p.Cout) { if (p.scale) sc[e] = p.scale[colb + e]; if (p.shift) sf[e] = p.shift[colb + e]; }
      }
      float sl[8];
      if constexpr (MODE == 2) {
  #pragma unroll
        for (int e = 0; e < 8; ++e) sl[e] = (colb + e < p.Cout) ? p.slope[colb + e] : 0.f;
      }
      // residual rows of ALL the wave's 32-row slabs are requested up front (TM * NIT <= 8 x 16 B per lane): the short-K layers that
      // carry a residual are bound by their HBM streams, and this doubles the bytes in flight during the LDS transposes
      constexpr bool RES_AHEAD = (TM * NIT <= 8);
      uint4 rall[RES_AHEAD ? TM * NIT : 1];
      if constexpr (RES_AHEAD) {
        if (p.res) {
  #pragma unroll
          for (int i = 0; i < TM; ++i)
  #pragma unroll
            for (int it = 0; it < NIT; ++it) {
              const int m = m0 + wm * TM * 32 + i * 32 + lr + it * RPI;
              const int mc = m < p.M ? m : p.M - 1;
              rall[i * NIT + it] = *reinterpret_cast<const uint4*>(p.res + (size_t)mc * p.ldr + colc);
            }
        }
      }
  #pragma unroll
      for (int i = 0; i < TM; ++i) {
  #pragma unroll
        for (int j = 0; j < TN; ++j)
  #pragma unroll
          for (int e = 0; e < 16; ++e) Et[((e & 3) + 8 * (e >> 2) + 4 * h) * ES + j * 32 + r] = acc[i][j][e];
        const int mrow0 = m0 + wm * TM * 32 + i * 32 + lr;
        uint4 rv[NIT];
        float4 r32a[S32 ? NIT : 1], r32b[S32 ? NIT : 1];      // fp32 residual rows (FE_PRECISION_RES32 streams)
        float gs[NIT];
        uint4 gv[NIT];
  #pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int m = mrow0 + it * RPI;
          const int mc = m < p.M ? m : p.M - 1;
          if constexpr (RES_AHEAD) { if (p.res) rv[it] = rall[i * NIT + it]; }
          else if (p.res) rv[it] = *reinterpret_cast<const uint4*>(p.res + (size_t)mc * p.ldr + colc);
          if constexpr (S32) {
            if (p.res32) {
              r32a[it] = *reinterpret_cast<const float4*>(p.res32 + (size_t)mc * p.ldr32 + colc);
              r32b[it] = *reinterpret_cast<const float4*>(p.res32 + (size_t)mc * p.ldr32 + colc + 4);
            }
          }
          if (p.gate) {
            if (p.gate_c1) gs[it] = (float)p.gate[(size_t)mc * p.ldg];
            else gv[it] = *reinterpret_cast<const uint4*>(p.gate + (size_t)mc * p.ldg + colc);
          }
        }
  #pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int m = mrow0 + it * RPI;
          const float4 v0 = *reinterpret_cast<const float4*>(&Et[(lr + it * RPI) * ES + lc]);
          const float4 v1 = *reinterpret_cast<const float4*>(&Et[(lr + it * RPI) * ES + lc + 4]);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          float rf[8], gf[8];
          bool hres = p.res != nullptr;
          if (p.res) h_unpack8<E>(rv[it], rf);
          if constexpr (S32) {
            if (p.res32) {
              hres = true;
              rf[0] = r32a[it].x; rf[1] = r32a[it].y; rf[2] = r32a[it].z; rf[3] = r32a[it].w;
              rf[4] = r32b[it].x; rf[5] = r32b[it].y; rf[6] = r32b[it].z; rf[7] = r32b[it].w;
            }
          }
          if (p.gate && !p.gate_c1) h_unpack8<E>(gv[it], gf);
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e] * sc[e] + sf[e];
            if (hres && !p.res_after_act) x += rf[e];
            if constexpr (MODE == 2) x = x > 0.f ? x : x * sl[e];
            else x = (S32 && p.exact_act) ? fe_apply_act_precise(x, p.act) : fe_apply_act_fast(x, p.act);
            if (hres && p.res_after_act) x += rf[e];
            if (p.gate) x *= p.gate_c1 ? gs[it] : gf[e];
            if (!cfull && colb + e >= p.Cout) x = 0.f;
            v[e] = x;
          }
          H8 o;
          o.u = make_uint4(fe_pack2((const E*)nullptr, v[0], v[1]), fe_pack2((const E*)nullptr, v[2], v[3]),
                           fe_pack2((const E*)nullptr, v[4], v[5]), fe_pack2((const E*)nullptr, v[6], v[7]));
          if (cok && m < p.M) {
            if constexpr (S32) {
              if (p.y) *reinterpret_cast<uint4*>(p.y + (size_t)m * p.ldy + colb) = o.u;
              if (p.split_lo_off > 0) {
                float w8[8];
                h_unpack8<E>(o.u, w8);
                H8 lo8;
                lo8.u = make_uint4(fe_pack2((const E*)nullptr, v[0] - w8[0], v[1] - w8[1]), fe_pack2((const E*)nullptr, v[2] - w8[2], v[3] - w8[3]),
                                   fe_pack2((const E*)nullptr, v[4] - w8[4], v[5] - w8[5]), fe_pack2((const E*)nullptr, v[6] - w8[6], v[7] - w8[7]));
                *reinterpret_cast<uint4*>(p.y + (size_t)m * p.ldy + p.split_lo_off + colb) = lo8.u;
              }
              if (p.y32) {      // (vec_epi with an fp32 output: Cout % 8 == 0, no pad_store)
                *reinterpret_cast<float4*>(p.y32 + (size_t)m * p.ldy32 + colb) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.y32 + (size_t)m * p.ldy32 + colb + 4) = make_float4(v[4], v[5], v[6], v[7]);
              }
            } else {
              *reinterpret_cast<uint4*>(p.y + (size_t)m * p.ldy + colb) = o.u;
            }
          }
        }
      }
      return;
    }
    // scalar epilogue (Cout or a stride not a multiple of 8): rare, small layers only
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
            float rs_ = p.res ? (float)p.res[(size_t)m * p.ldr + col] : 0.f;
            if constexpr (S32) { if (p.res32) rs_ = p.res32[(size_t)m * p.ldr32 + col]; }
            if (!p.res_after_act) v += rs_;
            if constexpr (MODE == 2) v = v > 0.f ? v : v * p.slope[col];
            else v = fe_apply_act(v, p.act);
            if (p.res_after_act) v += rs_;
            if (p.gate) v *= (float)p.gate[(size_t)m * p.ldg + (p.gate_c1 ? 0 : col)];
            if constexpr (S32) {
              if (p.y) stf(&p.y[(size_t)m * p.ldy + col], v);
              if (p.y32) p.y32[(size_t)m * p.ldy32 + col] = v;
            } else {
              stf(&p.y[(size_t)m * p.ldy + col], v);
            }
          }
        }
      }
    }
  
    return;
  }
}

template <class E, int WGM, int WGN, int TM, int TN, int UNITS, int MODE = 0, bool ONE_TAP = false, int S32 = 0>
void launch_bf16_variant(const ConvParamsT<E>& p, hipStream_t s) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
  constexpr size_t main_lds = (size_t)3 * (BM + BN) * 64 + (BN == 32 ? 4096 : 0);      // + the dummy pieces' scratch (32-column tiles)
#ifdef FE_PAIR_DUAL_STAGE
  constexpr int PAIR_IMAGES = 2;
#else
  constexpr int PAIR_IMAGES = 1;
#endif
  constexpr size_t epi_lds = TN > 1 ? (size_t)(S32 == 3 ? PAIR_IMAGES : 1) * (WGM * WGN) * (TM * 32) * (TN * 64 + 16)      // wide tiles: bf16 image of every wave's tile (two for the pair form)
                                    : (size_t)4 * 32 * (TN * 32 + 4) * sizeof(float);
  constexpr size_t lds = main_lds > epi_lds ? main_lds : epi_lds;
  auto kern = conv_bf16_kernel<E, WGM, WGN, TM, TN, UNITS, MODE, ONE_TAP, S32>;
  static std::atomic<uint64_t> lds_set{0};
  ensure_dynamic_lds((const void*)kern, lds, lds_set);
  const int ntotal = mtiles * ntiles;
  int gx = ntotal;
  hipLaunchKernelGGL(kern, dim3(gx, p.batch > 1 ? p.batch : 1), dim3(WGM * WGN * 64), lds, s, p, ntiles, ntotal);
  FE_HIP(hipGetLastError());
}


// wide tiles (TN = 2): instantiated in kernels_conv_bf16_wide{1,2,3,4}.hip (bf16) and kernels_conv_f16_wide{1,2,3,4}.hip (f16)
#define FE_WIDE_TILES(X, E)                                                                            \
  X template void launch_bf16_variant<E, 2, 2, 2, 2, 1, 0, true>(const ConvParamsT<E>&, hipStream_t);  \
  X template void launch_bf16_variant<E, 2, 2, 4, 2, 1, 0, true>(const ConvParamsT<E>&, hipStream_t);  \
  X template void launch_bf16_variant<E, 2, 2, 2, 2, 1, 0, false>(const ConvParamsT<E>&, hipStream_t); \
  X template void launch_bf16_variant<E, 2, 2, 4, 2, 1, 0, false>(const ConvParamsT<E>&, hipStream_t); \
  X template void launch_bf16_variant<E, 2, 2, 2, 2, 2, 0, false>(const ConvParamsT<E>&, hipStream_t); \
  X template void launch_bf16_variant<E, 2, 2, 4, 2, 2, 0, false>(const ConvParamsT<E>&, hipStream_t); \
  X template void launch_bf16_variant<E, 2, 4, 4, 2, 1, 0, true>(const ConvParamsT<E>&, hipStream_t);      /* 256x256, eight waves */
FE_WIDE_TILES(extern, bf16)
FE_WIDE_TILES(extern, f16)
// the fp32-stream forms of the wide tiles: kernels_conv_{bf16,f16}_s32.hip
#define FE_WIDE_TILES_S32(X, E)                                                                                  \
  X template void launch_bf16_variant<E, 2, 2, 2, 2, 1, 0, true, 1>(const ConvParamsT<E>&, hipStream_t);     \
  X template void launch_bf16_variant<E, 2, 2, 2, 2, 1, 0, false, 1>(const ConvParamsT<E>&, hipStream_t);    \
  X template void launch_bf16_variant<E, 2, 2, 4, 2, 1, 0, true, 2>(const ConvParamsT<E>&, hipStream_t);     /* 256x128: fp32 rows */ \
  X template void launch_bf16_variant<E, 2, 2, 4, 2, 1, 0, true, 3>(const ConvParamsT<E>&, hipStream_t);     /* 256x128: pair rows */ \
  X template void launch_bf16_variant<E, 2, 4, 4, 2, 1, 0, true, 2>(const ConvParamsT<E>&, hipStream_t);     /* 256x256: fp32 rows */
FE_WIDE_TILES_S32(extern, bf16)
FE_WIDE_TILES_S32(extern, f16)

}  // namespace fe

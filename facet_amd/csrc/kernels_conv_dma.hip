// Implicit-GEMM convolution, LDS-DMA main loop (the fast path of the engine: Cin % 16 == 0).
//
// Same math, tiling and epilogue as conv_igemm_kernel (kernels_conv.hip), but the K-slabs go HBM/L2 -> LDS with
// `global_load_lds_dwordx4` (no VGPR staging, no ds_write, ~6 VALU of address work per 1 KiB piece):
//   * LDS image per slab is lane-linear [row][16 floats] (64-B rows, what the DMA writes: wave base + lane*16);
//     bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE side
//     (lane at row r, slot q fetches global chunk q ^ ((r>>2)&3)) and mirrored in the read address.
//   * im2col zero padding: a lane whose tap falls outside the image (or whose row is past M) fetches from a 16-byte
//     zero page instead; validity of every tap of a row is a 64-bit mask computed once per workgroup.
//   * per K-step address work is one 64-bit add of a block-uniform tap offset to a per-row base pointer.
// Three-slab LDS ring: two slabs are in flight while one is consumed (counted vmcnt + one raw s_barrier per K-step).
#include "fe_common.h"
#include <cstdlib>

namespace fe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float apply_act_d(float v, int act) { return fe_apply_act(v, act); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE 0: 3-slab ring + double-buffered fragments (3 waves/SIMD for 64x64 wave tiles).
// MODE 1: lean - 2 slabs, one fragment set, register budget of 128 so FOUR waves/SIMD are resident.
// MODE 2: MODE 0 with a PReLU epilogue (per-channel slopes); separate instantiation so the hot tiles keep their registers.
// ONE_TAP: 1x1 kernels (GEMMs) with Cin % 32 == 0 - no tap masks, the K offset of a slab is the scalar offset of the buffer load.
template <int WGM, int WGN, int TM, int TN, int MODE = 0, bool ONE_TAP = false>
__global__ __launch_bounds__(256, (MODE == 1 ? 4 : (TM * TN >= 8 ? 2 : (TM * TN >= 4 ? 3 : 4)))) void conv_dma_kernel(ConvParams p, const int ntiles) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = 16;
  constexpr int AI = (BM + 63) / 64, BI = (BN + 63) / 64;   // DMA pieces per wave per slab (16 rows x 64 B each)
  constexpr int SLAB = (BM + BN) * BK;                      // floats per buffer
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  if (p.batch > 1) {
    const int b = blockIdx.y, bo = b / p.nb1, bi = b - bo * p.nb1;
    p.x += bo * p.xs2 + bi * p.xs1;
    p.w += bo * p.ws2 + bi * p.ws1;
    p.y += bo * p.ys2 + bi * p.ys1;
    if (p.shift) p.shift += bi * p.hs1;
  }
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int q8 = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int mt = swz / ntiles, nt = swz - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA source coordinates (per lane, fixed for the whole K loop) -------------------------------------
  const int rsub = lane >> 2, slot = lane & 3;
  const int gofs = (slot ^ ((rsub >> 2) & 3)) * 4;   // source chunk (floats) after the swizzle
  unsigned aoffs[AI];          // byte offset of the row's (kh=0,kw=0,ci=0) element from p.x (buffer addressing)
  unsigned long long amask[AI];
  const int HoWo = p.Ho * p.Wo;
  const int ntaps = p.KH * p.KW;
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int row = 16 * (4 * j + wave) + rsub;
    const int m = m0 + row;
    const bool valid = (row < BM) && (m < p.M);
    const int mm = valid ? m : 0;
    if (ONE_TAP && p.unit_stride) {      // 1x1, stride 1, no padding: output pixel m IS input pixel m (no divisions in the prologue)
      aoffs[j] = (unsigned)(((long long)mm * p.ldx + gofs) * 4);
      amask[j] = valid ? 1ull : 0ull;
      continue;
    }
    const int nimg = mm / HoWo;
    const int rem = mm - nimg * HoWo;
    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    const int ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
    const long long pix = ((long long)nimg * p.H + ih0) * p.W + iw0;
    aoffs[j] = (unsigned)((pix * p.ldx + gofs) * 4);
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
  unsigned boffs[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int row = 16 * (4 * j + wave) + rsub;
    int n = n0 + row;
    if (n > p.Cout - 1) n = p.Cout - 1;   // columns past Cout are computed on a valid row and discarded
    boffs[j] = (unsigned)(((size_t)n * p.ldw + gofs) * 4);
  }
  // Buffer addressing (p.buf_ok: both operands span < 4 GiB): out-of-range offsets read as zero in hardware, so a
  // padding tap is one v_cndmask to an out-of-range offset instead of a 64-bit pointer select to a zero page.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_span, 0x00020000);

  // block-uniform running tap state
  int tap = 0, kh = 0, kw = 0, ci = 0;
  const int nk = p.Kp / BK;

  // One K-slab = AI A-pieces + BI B-pieces per wave. issue_begin fixes the slab's addresses, issue_piece(q) launches one 1-KiB
  // LDS-DMA, issue_end advances the tap state: the main loop spreads the pieces BETWEEN the MFMAs of the current slab, where their
  // issue cost (address VALU + the VMEM issue slot, 60-185 cycles each) disappears in the shadow of the 64-cycle MFMAs.
  float* iAb = nullptr; float* iBb = nullptr;
  int i_tb = 0, i_tbb2 = 0;
  bool i_cin_ok = true;
  auto issue_begin = [&](int kt, int buf) {
    iAb = smem + buf * SLAB;
    iBb = iAb + BM * BK;
    i_tb = ((kh * p.dh * p.W + kw * p.dw) * p.ldx + ci) * 4;   // byte offset of this tap/channel slab
    i_tbb2 = kt * (BK * 4);
    i_cin_ok = ci < p.Cin;                                       // ci == Cin only in the zero-padded K tail
  };
  auto issue_piece = [&](int q) {      // q < AI: A piece q; else B piece q - AI  (q is a compile-time constant at every call site)
    if (q < AI) {
      const int j = q;
      if (16 * (4 * j + wave) < BM) {
        const bool ok = ((amask[j] >> tap) & 1ull) && i_cin_ok;
        const unsigned off = ok ? aoffs[j] + (unsigned)i_tb : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(iAb + 256 * (4 * j + wave)), 16, (int)off, 0, 0, 0);
      }
    } else {
      const int j = q - AI;
      if (16 * (4 * j + wave) < BN)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(iBb + 256 * (4 * j + wave)), 16, (int)boffs[j], i_tbb2, 0, 0);
    }
  };
  auto issue_end = [&]() {
    // K order: 16-channel block outer, tap inner - the 9 taps of a 3x3 re-read the same input rows back to back,
    // so the re-reads hit L1/L2 instead of coming back from the Infinity Cache a third of a K-loop later.
    ++tap;
    if (++kw == p.KW) { kw = 0; ++kh; }
    if (tap == ntaps) { tap = 0; kh = 0; kw = 0; ci += BK; }
  };
  auto issue = [&](int kt, int buf) {
    issue_begin(kt, buf);
#pragma unroll
    for (int q = 0; q < AI + BI; ++q) issue_piece(q);
    issue_end();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets (floats): row*16 + ((2s+h) ^ sw)*4, sw = (row>>2)&3 = (r>>2)&3
  const int sw = (r >> 2) & 3;
  const int fo0 = ((h ^ sw) << 2), fo1 = fo0 ^ 8;
  const int aoff = (wm * TM * 32 + r) * BK, boff = BM * BK + (wn * TN * 32 + r) * BK;

  // 3-slab LDS ring + register double-buffered fragments:
  //   iteration t:  wait own pieces of slab t+1 (counted vmcnt keeps slab t+2 in flight) -> raw barrier (publishes
  //   slab t+1, retires every wave's reads of slab t) -> DMA slab t+3 into slab t's buffer -> ds_read fragments of
  //   slab t+1 into the other register set -> 32 MFMAs on the fragments of slab t (the LDS reads land in their shadow).
  // Fragment reads are inline asm on purpose: hipcc orders every C++ LDS read behind ALL outstanding LDS-DMA
  // (s_waitcnt vmcnt(0)), which would drain the ring each step.
  constexpr int NPW_A = BM / 64;
  static_assert(BM % 64 == 0, "A pieces must be uniform over waves");
  const int npw = NPW_A + ((16 * wave < BN) ? 1 : 0) + ((BN > 64 && 16 * (4 + wave) < BN) ? 1 : 0) +
                  ((BN > 128 && 16 * (8 + wave) < BN) ? 1 : 0) + ((BN > 192 && 16 * (12 + wave) < BN) ? 1 : 0);
  auto wait_vm = [&](int n) {   // n is wave-uniform
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
#ifndef FE_DMA_SPREAD
#define FE_DMA_SPREAD 1     // build-time A/B hook: 0 = all pieces of a slab issued in front of the MFMA burst (round-1 form)
#endif
  constexpr bool spread_issue = FE_DMA_SPREAD != 0;
  v4f fa[2][2 * TM], fb[2][2 * TN];   // [register set][fragment]; indices are compile-time everywhere below
#define FE_READ_FRAGS(SET, SLOT)                                                                                   \
  {                                                                                                                \
    const unsigned sb_ = lds_base + (unsigned)((SLOT) * SLAB * 4);                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                               \
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[SET][2 * i]) : "v"(sb_ + (unsigned)((aoff + i * 32 * BK + fo0) * 4)));     \
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[SET][2 * i + 1]) : "v"(sb_ + (unsigned)((aoff + i * 32 * BK + fo1) * 4))); \
    }                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                               \
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[SET][2 * j]) : "v"(sb_ + (unsigned)((boff + j * 32 * BK + fo0) * 4)));     \
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[SET][2 * j + 1]) : "v"(sb_ + (unsigned)((boff + j * 32 * BK + fo1) * 4))); \
    }                                                                                                              \
  }
#define FE_MFMA_BURST(SET)                                                                                         \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                 \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].x, fb[SET][2 * j + hh].x, acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].y, fb[SET][2 * j + hh].y, acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].z, fb[SET][2 * j + hh].z, acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].w, fb[SET][2 * j + hh].w, acc[i][j], 0, 0, 0); \
      }
  // The same burst with this wave's DMA pieces of slab kt+3 spread between its MFMA quads (DO: block-uniform, false in the tail).
  // Quad q = the four k-slots of one (hh, i, j); piece n goes behind quad max(0, (n + 1) * Q / (NP + 1) - 1) (always < Q). sched_barrier pins the order:
  // hipcc otherwise gathers the loads in front of the MFMAs again.
#define FE_MFMA_BURST_ISSUE(SET, DO)                                                                               \
  {                                                                                                                \
    constexpr int Q_ = 2 * TM * TN, NP_ = AI + BI;                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < Q_; ++q_) {                                                            \
      const int hh = q_ / (TM * TN), i = (q_ / TN) % TM, j = q_ % TN;                                              \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].x, fb[SET][2 * j + hh].x, acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].y, fb[SET][2 * j + hh].y, acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].z, fb[SET][2 * j + hh].z, acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i + hh].w, fb[SET][2 * j + hh].w, acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int n_ = 0; n_ < NP_; ++n_)                                                           \
        if (((n_ + 1) * Q_ / (NP_ + 1) - 1 < 0 ? 0 : (n_ + 1) * Q_ / (NP_ + 1) - 1) == q_) {                       \
          __builtin_amdgcn_sched_barrier(0);                                                                       \
          if (DO) issue_piece(n_);                                                                                 \
          __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                          \
    }                                                                                                              \
  }
  // one K-step: fragments of slab kt are in register set CUR
#define FE_STEP(CUR, NXT, KT)                                                                                      \
  {                                                                                                                \
    const int kt_ = (KT);                                                                                          \
    bool do_issue_ = false;                                                                                        \
    if (kt_ + 1 < nk) {                                                                                            \
      wait_vm(kt_ + 2 < nk ? npw : 0);                                                                             \
      __builtin_amdgcn_s_barrier();                                                                                \
      do_issue_ = kt_ + 3 < nk;                                                                                    \
      if (do_issue_) {                                                                                             \
        if constexpr (spread_issue) issue_begin(kt_ + 3, kt_ % 3);                                                 \
        else issue(kt_ + 3, kt_ % 3);                                                                              \
      }                                                                                                            \
      FE_READ_FRAGS(NXT, (kt_ + 1) % 3)                                                                            \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if constexpr (spread_issue) {                                                                                  \
      FE_MFMA_BURST_ISSUE(CUR, do_issue_)                                                                          \
      if (do_issue_) issue_end();                                                                                  \
    } else {                                                                                                       \
      FE_MFMA_BURST(CUR)                                                                                           \
    }                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }

  // LEAN loop (tiles whose waves all issue the same number of pieces; see kernels_conv_bf16.hip, where the same form was measured
  // first): unrolled by 6 = ring slot (mod 3) x register set (mod 2), so every LDS read is `base + immediate`, every DMA destination
  // `base + immediate`, the counted wait an immediate; the last three slabs still issue their pieces, with out-of-range offsets (zero
  // fill, no memory traffic), so every iteration is the same straight-line code (~45 instead of ~130 instructions around the 16-32
  // MFMAs of a slab); 1x1 kernels pass the slab's K offset as the scalar offset of the load.
  // (32-column tiles: only waves 0 and 1 own a B piece; the other two issue an out-of-range dummy piece into a scratch KiB behind the
  // ring, so every wave still counts the same number of pieces per slab)
  constexpr bool LEAN = (MODE != 1) && (BM % 64 == 0) && (BN % 64 == 0 || BN == 32) && (TM <= 2) && (TN <= 2);
  if constexpr (LEAN) {
    constexpr int NPW = AI + BI;
    constexpr int SLABB = SLAB * 4;                         // bytes per ring slot
    const unsigned bA0 = lds_base + (unsigned)((aoff + fo0) * 4), bA1 = lds_base + (unsigned)((aoff + fo1) * 4);
    const unsigned bB0 = lds_base + (unsigned)((boff + fo0) * 4), bB1 = lds_base + (unsigned)((boff + fo1) * 4);
    char* const dA = reinterpret_cast<char*>(smem) + 1024 * wave;            // + SLOT * SLABB + 4096 * j
    char* const dB = reinterpret_cast<char*>(smem) + BM * 64 + 1024 * wave;
    unsigned aoffs_l[AI];
#pragma unroll
    for (int j = 0; j < AI; ++j) aoffs_l[j] = (ONE_TAP && !(amask[j] & 1ull)) ? 0xFFFFFFF0u : aoffs[j];
    int g3 = 0;                                             // slab being issued
    int l_tb = 0; bool l_cok = true;
    auto lean_begin = [&]() {
      if constexpr (!ONE_TAP) {
        l_tb = ((kh * p.dh * p.W + kw * p.dw) * p.ldx + ci) * 4;
        l_cok = ci < p.Cin;
      }
    };
    auto lean_end = [&]() {
      if constexpr (!ONE_TAP) {
        ++tap;
        if (++kw == p.KW) { kw = 0; ++kh; }
        if (tap == ntaps) { tap = 0; kh = 0; kw = 0; ci += BK; }
      }
      ++g3;
    };
#define FL_PIECE(SLOT, Q)                                                                                               \
    {                                                                                                                   \
      const bool live_ = g3 < nk;                                                                                       \
      if constexpr ((Q) < AI) {                                                                                         \
        if constexpr (ONE_TAP) {                                                                                        \
          const unsigned off_ = live_ ? aoffs_l[(Q) < AI ? (Q) : 0] : 0xFFFFFFF0u;                                      \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dA + (SLOT) * SLABB + 4096 * (Q)), 16, (int)off_, g3 * 64, 0, 0);   \
        } else {                                                                                                        \
          const bool ok_ = ((amask[(Q) < AI ? (Q) : 0] >> tap) & 1ull) && l_cok && live_;                                \
          const unsigned off_ = ok_ ? aoffs[(Q) < AI ? (Q) : 0] + (unsigned)l_tb : 0xFFFFFFF0u;                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dA + (SLOT) * SLABB + 4096 * (Q)), 16, (int)off_, 0, 0, 0);         \
        }                                                                                                               \
      } else {                                                                                                          \
        const bool mine_ = BN >= 64 || 16 * wave < BN;                                                                  \
        const unsigned off_ = (live_ && mine_) ? boffs[(Q) >= AI ? (Q) - AI : 0] : 0xFFFFFFF0u;                         \
        char* const dst_ = mine_ ? dB + (SLOT) * SLABB + 4096 * ((Q) - AI) : reinterpret_cast<char*>(smem) + 3 * SLABB + 1024 * wave; \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)dst_, 16, (int)off_, g3 * 64, 0, 0);                      \
      }                                                                                                                 \
    }
#define FL_READ1(DST, BASE, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE), "i"(IMM));
#define FL_READ_FRAGS(SET, SLOT)                                                                                        \
    {                                                                                                                   \
      FL_READ1(fa[SET][0], bA0, (SLOT) * SLABB) FL_READ1(fa[SET][1], bA1, (SLOT) * SLABB)                               \
      if constexpr (TM > 1) { FL_READ1(fa[SET][2], bA0, (SLOT) * SLABB + 2048) FL_READ1(fa[SET][3], bA1, (SLOT) * SLABB + 2048) } \
      FL_READ1(fb[SET][0], bB0, (SLOT) * SLABB) FL_READ1(fb[SET][1], bB1, (SLOT) * SLABB)                               \
      if constexpr (TN > 1) { FL_READ1(fb[SET][2], bB0, (SLOT) * SLABB + 2048) FL_READ1(fb[SET][3], bB1, (SLOT) * SLABB + 2048) } \
    }
#define FL_MFMA4(SET, Q)                                                                                                \
    {                                                                                                                   \
      constexpr int hh_ = (Q) / (TM * TN), i_ = ((Q) / TN) % TM, j_ = (Q) % TN;                                         \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i_ + hh_].x, fb[SET][2 * j_ + hh_].x, acc[i_][j_], 0, 0, 0); \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i_ + hh_].y, fb[SET][2 * j_ + hh_].y, acc[i_][j_], 0, 0, 0); \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i_ + hh_].z, fb[SET][2 * j_ + hh_].z, acc[i_][j_], 0, 0, 0); \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][2 * i_ + hh_].w, fb[SET][2 * j_ + hh_].w, acc[i_][j_], 0, 0, 0); \
    }
#define FL_PIECE_AT(SLOT, N, Q)                                                                                          \
    if constexpr ((N) < NPW && (((N) + 1) * (2 * TM * TN) / (NPW + 1) - 1 < 0 ? 0 : ((N) + 1) * (2 * TM * TN) / (NPW + 1) - 1) == (Q)) { \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      FL_PIECE(SLOT, N)                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
#define FL_STEP_Q(CUR, SLOTI, Q)                                                                                         \
    if constexpr ((Q) < 2 * TM * TN) {                                                                                   \
      FL_MFMA4(CUR, Q)                                                                                                   \
      FL_PIECE_AT(SLOTI, 0, Q) FL_PIECE_AT(SLOTI, 1, Q) FL_PIECE_AT(SLOTI, 2, Q) FL_PIECE_AT(SLOTI, 3, Q) FL_PIECE_AT(SLOTI, 4, Q) FL_PIECE_AT(SLOTI, 5, Q) \
    }
#define FL_SLAB(CUR, NXT, SLOTR, SLOTI)                                                                                  \
    {                                                                                                                    \
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPW) : "memory");                                                         \
      __builtin_amdgcn_s_barrier();                                                                                      \
      lean_begin();                                                                                                      \
      FL_READ_FRAGS(NXT, SLOTR)                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      FL_STEP_Q(CUR, SLOTI, 0) FL_STEP_Q(CUR, SLOTI, 1) FL_STEP_Q(CUR, SLOTI, 2) FL_STEP_Q(CUR, SLOTI, 3)                \
      FL_STEP_Q(CUR, SLOTI, 4) FL_STEP_Q(CUR, SLOTI, 5) FL_STEP_Q(CUR, SLOTI, 6) FL_STEP_Q(CUR, SLOTI, 7)                \
      lean_end();                                                                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
    static_assert(NPW <= 6 && 2 * TM * TN <= 8, "lean loop: piece / MFMA slots");
#define FL_PIECE_N(SLOT, N) if constexpr ((N) < NPW) FL_PIECE(SLOT, N)
#define FL_PIECE_ALL(SLOT) FL_PIECE_N(SLOT, 0) FL_PIECE_N(SLOT, 1) FL_PIECE_N(SLOT, 2) FL_PIECE_N(SLOT, 3) FL_PIECE_N(SLOT, 4) FL_PIECE_N(SLOT, 5)
#define FL_ISSUE_ALL(SLOT) { lean_begin(); FL_PIECE_ALL(SLOT) lean_end(); }
    FL_ISSUE_ALL(0) FL_ISSUE_ALL(1) FL_ISSUE_ALL(2)         // slab 2 is a dummy when nk == 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NPW) : "memory");
    __builtin_amdgcn_s_barrier();
    FL_READ_FRAGS(0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int g = 0; g < nk; g += 6) {                       // nk is even (Kp % 32 == 0)
      FL_SLAB(0, 1, 1, 0)
      FL_SLAB(1, 0, 2, 1)
      if (g + 2 < nk) {
        FL_SLAB(0, 1, 0, 2)
        FL_SLAB(1, 0, 1, 0)
      }
      if (g + 4 < nk) {
        FL_SLAB(0, 1, 2, 1)
        FL_SLAB(1, 0, 0, 2)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dummy pieces of the tail have landed before the epilogue reuses the ring
#undef FL_SLAB
#undef FL_STEP_Q
#undef FL_PIECE_AT
#undef FL_MFMA4
#undef FL_READ_FRAGS
#undef FL_READ1
#undef FL_PIECE
#undef FL_ISSUE_ALL
#undef FL_PIECE_N
#undef FL_PIECE_ALL
  } else if constexpr (MODE == 1) {
    issue(0, 0);
    wait_vm(0);
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      FE_READ_FRAGS(0, kt & 1)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      FE_MFMA_BURST(0)
      wait_vm(0);
      __builtin_amdgcn_s_barrier();
    }
  } else {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 2) issue(2, 2);
    wait_vm(nk > 2 ? 2 * npw : (nk > 1 ? npw : 0));
    __builtin_amdgcn_s_barrier();
    FE_READ_FRAGS(0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; kt += 2) {
      FE_STEP(0, 1, kt)
      if (kt + 1 < nk) FE_STEP(1, 0, kt + 1)
    }
  }
#undef FE_STEP
#undef FE_MFMA_BURST_ISSUE
#undef FE_MFMA_BURST
#undef FE_READ_FRAGS
  __syncthreads();   // all fragment reads retired before the epilogue reuses the slabs as staging

  // Two epilogue forms. Narrow wave tiles (TN = 1: the tiles of the HBM-bound short-K layers) keep the fully unrolled, branch-free
  // row code with every residual row requested up front - measured 64.6 vs 55.1 TFLOP/s on the K = 64 -> 256 expand against the
  // rolled form (hipcc drains the outstanding loads at the scalar branches the rolled form has per row). Wide wave tiles take the
  // compact rolled form: unrolled, their epilogue spilled 70 registers and ran to tens of thousands of instructions.
  if constexpr (TN == 1) {
  // ---- epilogue (same as conv_igemm_kernel): transpose through a wave-private LDS region ------------------
    if (p.vec_epi) {
      constexpr int WC = TN * 32, ES = WC + 4, LPR = WC / 4, RPI = 64 / LPR, NIT = 32 / RPI;
      float* E = smem + wave * 32 * ES;
      const int lr = lane / LPR, lc = (lane % LPR) * 4;
      const int colb = n0 + wn * WC + lc;
      const bool cok = colb < p.Cout;
      const int colc = cok ? colb : 0;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + colc);
      if (p.shift) sf = *reinterpret_cast<const float4*>(p.shift + colc);
      // Narrow wave tiles (TN = 1: the 128x64 / 256x32 / 64x64 blocks that carry the HBM-bound short-K layers) fetch the residual
      // rows of ALL their 32-row slabs up front: TM * NIT <= 8 float4 per lane, twice the bytes in flight while the accumulators
      // go through the LDS transpose. Wider tiles load per slab (their registers are taken by the accumulators).
      constexpr bool RES_AHEAD = (TM * NIT <= 8);
      float4 rall[RES_AHEAD ? TM * NIT : 1];
      if constexpr (RES_AHEAD) {
        if (p.res) {
  #pragma unroll
          for (int i = 0; i < TM; ++i)
  #pragma unroll
            for (int it = 0; it < NIT; ++it) {
              const int m = m0 + wm * TM * 32 + i * 32 + lr + it * RPI;
              const int mc = m < p.M ? m : p.M - 1;
              rall[i * NIT + it] = *reinterpret_cast<const float4*>(p.res + (size_t)mc * p.ldr + colc);
            }
        }
      }
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
          if constexpr (RES_AHEAD) { if (p.res) rv[it] = rall[i * NIT + it]; }
          else if (p.res) rv[it] = *reinterpret_cast<const float4*>(p.res + (size_t)mc * p.ldr + colc);
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
          if constexpr (MODE == 2) {
            const float4 sl = *reinterpret_cast<const float4*>(p.slope + colc);
            v.x = v.x > 0.f ? v.x : v.x * sl.x; v.y = v.y > 0.f ? v.y : v.y * sl.y;
            v.z = v.z > 0.f ? v.z : v.z * sl.z; v.w = v.w > 0.f ? v.w : v.w * sl.w;
          } else {
            v.x = apply_act_d(v.x, p.act); v.y = apply_act_d(v.y, p.act); v.z = apply_act_d(v.z, p.act); v.w = apply_act_d(v.w, p.act);
          }
          if (p.res && p.res_after_act) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
          if (p.gate) { v.x *= gv[it].x; v.y *= gv[it].y; v.z *= gv[it].z; v.w *= gv[it].w; }
          if (cok && m < p.M) *reinterpret_cast<float4*>(p.y + (size_t)m * p.ldy + colb) = v;
        }
      }
      return;
    }
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
            if constexpr (MODE == 2) v = v > 0.f ? v : v * p.slope[col];
            else v = apply_act_d(v, p.act);
            if (p.res && p.res_after_act) v += p.res[(size_t)m * p.ldr + col];
            if (p.gate) v *= p.gate[(size_t)m * p.ldg + (p.gate_c1 ? 0 : col)];
            p.y[(size_t)m * p.ldy + col] = v;
          }
        }
      }
    }
  
    return;
  } else {
  // ---- epilogue: transpose through a wave-private LDS region; each lane then owns 4 consecutive channels of a row -----------
    // Only the accumulator -> LDS writes are unrolled (register indices). The read-back / scale / residual / activation / gate / store
    // part is a loop over the lane's rows with ONE activation branch per row: fully unrolled with the activation chain (erf, exp,
    // log1p) inlined per element the epilogue alone was tens of thousands of instructions per kernel - more than the instruction
    // cache holds - and the 128x128 tile spilled 70 registers in it. 16-byte accesses when legal (p.vec_epi), per-element otherwise.
    constexpr int WC = TN * 32, ES = WC + 4, LPR = WC / 4, RPI = 64 / LPR, NIT = 32 / RPI;
    float* E = smem + wave * 32 * ES;
    const int lr = lane / LPR, lc = (lane % LPR) * 4;
    const int colb = n0 + wn * WC + lc;
    const bool vec = p.vec_epi != 0;
    const bool cok = colb < p.Cout;
    const bool cfull = colb + 4 <= p.Cout;
    float sc[4], sf[4], sl[4];
  #pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool cv = colb + e < p.Cout;
      sc[e] = (cv && p.scale) ? p.scale[colb + e] : 1.f;
      sf[e] = (cv && p.shift) ? p.shift[colb + e] : 0.f;
      sl[e] = (MODE == 2 && cv) ? p.slope[colb + e] : 0.f;
    }
    const int act = p.act;
    auto load4 = [&](const float* base, size_t row_off) -> float4 {
      if (vec && cfull) return *reinterpret_cast<const float4*>(base + row_off + colb);
      float w[4];
  #pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (colb + e < p.Cout) ? base[row_off + colb + e] : 0.f;
      return make_float4(w[0], w[1], w[2], w[3]);
    };
    auto row_of = [&](int i, int it) { return m0 + wm * TM * 32 + i * 32 + lr + it * RPI; };
    auto fetch_res = [&](int m) -> float4 {
      const int mc = m < p.M ? m : p.M - 1;
      return p.res ? load4(p.res, (size_t)mc * p.ldr) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto fetch_gate = [&](int m) -> float4 {
      const int mc = m < p.M ? m : p.M - 1;
      if (!p.gate) return make_float4(1.f, 1.f, 1.f, 1.f);
      if (p.gate_c1) { const float g = p.gate[(size_t)mc * p.ldg]; return make_float4(g, g, g, g); }
      return load4(p.gate, (size_t)mc * p.ldg);
    };
    auto read_row = [&](int erow) -> float4 { return *reinterpret_cast<const float4*>(&E[erow * ES + lc]); };
    auto finish_row = [&](int m, const float4 v0, const float4 ru, const float4 gu) {
      float v[4] = {v0.x, v0.y, v0.z, v0.w};
      const float rf[4] = {ru.x, ru.y, ru.z, ru.w}, gf[4] = {gu.x, gu.y, gu.z, gu.w};
  #pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e] * sc[e] + sf[e];
        if (p.res && !p.res_after_act) x += rf[e];
        v[e] = x;
      }
      if constexpr (MODE == 2) {
  #pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
      } else if (act == ACT_RELU) {
  #pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      } else if (act == ACT_GELU) {
  #pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
      } else if (act == ACT_SIGMOID) {
  #pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      } else if (act == ACT_SOFTPLUS) {
  #pragma unroll 1
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 20.f ? v[e] : log1pf(expf(v[e]));
      }
  #pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (p.res && p.res_after_act) x += rf[e];
        if (p.gate) x *= gf[e];
        v[e] = x;
      }
      if (cok && m < p.M) {
        float* yp = p.y + (size_t)m * p.ldy + colb;
        if (vec) {
          *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            if (colb + e < p.Cout) yp[e] = v[e];
        }
      }
    };
    // Narrow wave tiles (TN = 1: the 128x64 / 256x32 / 64x64 blocks that carry the HBM-bound short-K layers) request the residual and
    // gate rows of ALL their 32-row slabs before the first transpose (TM * NIT <= 8 float4 each): twice the bytes in flight.
    constexpr bool AHEAD = (TM * NIT <= 8);
    float4 rall[AHEAD ? TM * NIT : 1];      // the gate (one layer type of the TOPIQ head) is fetched per row, just in time
    if constexpr (AHEAD) {
  #pragma unroll
      for (int i = 0; i < TM; ++i)
  #pragma unroll
        for (int it = 0; it < NIT; ++it) rall[i * NIT + it] = fetch_res(row_of(i, it));
    }
  #pragma unroll
    for (int i = 0; i < TM; ++i) {
  #pragma unroll
      for (int j = 0; j < TN; ++j)
  #pragma unroll
        for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * h) * ES + j * 32 + r] = acc[i][j][e];
      if constexpr (AHEAD) {
        float4 ev[NIT];      // all LDS reads of the slab first: their latency overlaps instead of adding up row by row
  #pragma unroll
        for (int it = 0; it < NIT; ++it) ev[it] = read_row(lr + it * RPI);
  #pragma unroll
        for (int it = 0; it < NIT; ++it) finish_row(row_of(i, it), ev[it], rall[i * NIT + it], fetch_gate(row_of(i, it)));
      } else {
        float4 rn = fetch_res(row_of(i, 0)), gn = fetch_gate(row_of(i, 0));
  #pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
          const float4 ru = rn, gu = gn;
          if (it + 1 < NIT) { rn = fetch_res(row_of(i, it + 1)); gn = fetch_gate(row_of(i, it + 1)); }
          finish_row(row_of(i, it), read_row(lr + it * RPI), ru, gu);
        }
      }
    }
  
  }
}

template <int WGM, int WGN, int TM, int TN, int MODE = 0, bool ONE_TAP = false>
static void launch_dma_variant(const ConvParams& p, hipStream_t s) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
  constexpr size_t main_lds = (size_t)(MODE == 1 ? 2 : 3) * (BM + BN) * 16 * sizeof(float) + (BN == 32 ? 4096 : 0);   // + dummy-piece scratch
  constexpr size_t epi_lds = (size_t)4 * 32 * (TN * 32 + 4) * sizeof(float);
  constexpr size_t lds = main_lds > epi_lds ? main_lds : epi_lds;
  auto kern = conv_dma_kernel<WGM, WGN, TM, TN, MODE, ONE_TAP>;
  static std::atomic<uint64_t> lds_set{0};
  ensure_dynamic_lds((const void*)kern, lds, lds_set);
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles, p.batch > 1 ? p.batch : 1), dim3(256), lds, s, p, ntiles);
  FE_HIP(hipGetLastError());
}

// tile: 1 = 128x128, 2 = 256x64, 3 = 256x32, 4 = 64x64, 7 = 128x64
void launch_conv_dma(const ConvParams& p0, int tile, hipStream_t s) {
  FE_CHECK(p0.Cin % 16 == 0 && p0.KH * p0.KW < 64, "conv_dma: needs Cin %% 16 == 0 and < 64 taps");
  ConvParams p = p0;
  const unsigned long long xs = ((unsigned long long)p.N * p.H * p.W - 1) * (unsigned long long)p.ldx * 4 + (unsigned long long)p.Cin * 4;
  const unsigned long long ws = ((unsigned long long)p.Cout - 1) * (unsigned long long)p.ldw * 4 + (unsigned long long)p.Kp * 4;
  static const bool no_buf = getenv("FE_NO_BUF") != nullptr;
  p.buf_ok = !no_buf && xs < 0xFFFFFF00ull && ws < 0xFFFFFF00ull;
  FE_CHECK(p.buf_ok, "conv_dma: operand spans exceed 32-bit buffer addressing (caller must use the register-staged kernel)");
  p.x_span = (unsigned)xs; p.w_span = (unsigned)ws;
  p.unit_stride = (p.KH == 1 && p.KW == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.Ho == p.H && p.Wo == p.W &&
                   (long long)p.N * p.H * p.W == (long long)p.M) ? 1 : 0;
  if (p.act == ACT_PRELU) {
    FE_CHECK(p.slope, "conv_dma: PReLU without slopes");
    switch (tile) {
      case 3: launch_dma_variant<4, 1, 2, 1, 2>(p, s); break;
      case 4: launch_dma_variant<2, 2, 1, 1, 2>(p, s); break;
      case 2: case 22: case 7: launch_dma_variant<2, 2, 2, 1, 2>(p, s); break;
      default: launch_dma_variant<2, 2, 2, 2, 2>(p, s); break;
    }
    return;
  }
  if (p.KH * p.KW == 1 && p.Cin % 32 == 0 && p.Kp == p.Cin) {   // 1x1 kernels: the lean loop's scalar-offset form
    switch (tile) {
      case 1: launch_dma_variant<2, 2, 2, 2, 0, true>(p, s); return;
      case 7: launch_dma_variant<2, 2, 2, 1, 0, true>(p, s); return;
      case 4: launch_dma_variant<2, 2, 1, 1, 0, true>(p, s); return;
      default: break;
    }
  }
  switch (tile) {
    case 1: launch_dma_variant<2, 2, 2, 2>(p, s); break;
    case 21: launch_dma_variant<2, 2, 2, 2, 1>(p, s); break;   // lean 128x128, 4 waves/SIMD
    case 22: launch_dma_variant<4, 1, 2, 2, 1>(p, s); break;   // lean 256x64
    case 2: launch_dma_variant<4, 1, 2, 2>(p, s); break;
    case 3: launch_dma_variant<4, 1, 2, 1>(p, s); break;
    case 4: launch_dma_variant<2, 2, 1, 1>(p, s); break;
    case 7: launch_dma_variant<2, 2, 2, 1>(p, s); break;
    case 8: launch_dma_variant<2, 2, 4, 2>(p, s); break;   // 256x128
    case 9: launch_dma_variant<2, 2, 2, 4>(p, s); break;   // 128x256
    default: FE_CHECK(false, "conv_dma: unknown tile %d", tile);
  }
}

}  // namespace fe

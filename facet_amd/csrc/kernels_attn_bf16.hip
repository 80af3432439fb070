// Fused multi-head attention for head_dim 64 on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): the bf16 twin of kernels_attn.hip
// for the reduced-precision path. bf16 Q / K / V^T in, fp32 scores, online softmax in fp32 registers, P rounded to bf16 only as the
// operand of P.V, fp32 accumulation of O, bf16 out. Stands behind torch.nn.MultiheadAttention inside open_clip's resblocks
// (reference: model.encode_image, processing/scorer.py:662, run in half precision on a GPU at :513-516) and pyiqa's CFANet layers.
//
// One wave owns 32 queries; a workgroup (NW waves) shares 32-key K / V^T tiles through LDS.
//   S^T = K Q^T     A = K tile (rows = keys; lane (r, h) reads 8 consecutive d: one ds_read_b128), B = the wave's Q fragment kept
//                   in registers (4 x 8 bf16 per lane), 4 MFMAs per tile
//   online softmax  each lane owns ONE query (accumulator column) and 16 of the tile's 32 keys, the other 16 sit in lane ^ 32
//   O^T += V^T P^T  the exponentiated S^T accumulator, rounded pairwise to bf16, IS the B operand: registers 8s .. 8s+7 form the
//                   fragment of k-step s, whose element j is key 16s + 8(j>>2) + 4h + (j&3) - so the A operand (V^T rows = d)
//                   takes the same keys with two 8-byte LDS reads per step (the k order inside a step is free as long as both
//                   operands agree).
// Q is pre-scaled by 1/sqrt(64) in the projection epilogue; V^T comes straight from the role-swapped projection GEMM.
#include "fe_common.h"

namespace fe {

typedef float ah_f32x16 __attribute__((ext_vector_type(16)));
// E = bf16 | f16 (fe_common.h: fe_mfma16 / fe_pack2 overloads): one source for both 2-byte types

constexpr int ATH_KS = 144;   // K tile row stride in BYTES: 128 + 16 -> conflict-free ds_read_b128 over rows distinct mod 16
constexpr int ATH_VS = 72;    // V^T tile row stride in BYTES: 64 + 8 -> conflict-free ds_read_b64 over 32 rows

template <class E>
struct AttnParamsH {
  const E* q; int ldq;         // [B*Lq][ldq], head h at column h*64
  const E* k; int ldk;         // [B*Lk][ldk]
  const E* vt; int lp;         // [B][d_model][lp]  (V transposed, zero padded to lp >= roundup32(Lk))
  const float* bv;             // [d_model] V bias, added to the output (softmax rows sum to 1)
  E* o; int ldo;               // [B*Lq][ldo]
  int B, H, Lq, Lk, dmodel;
  int causal;
};

union AH8 { uint4 u; fe_v4f f; };

template <class E, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_bf16_kernel(const AttnParamsH<E> p) {
  const E* const tag = nullptr;
  __shared__ __attribute__((aligned(16))) char Ks[2][32 * ATH_KS];
  __shared__ __attribute__((aligned(16))) char Vs[2][64 * ATH_VS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.H, head = bh - b * p.H;
  const E* Qp = p.q + (size_t)b * p.Lq * p.ldq + head * 64;
  const E* Kp = p.k + (size_t)b * p.Lk * p.ldk + head * 64;
  const E* Vp = p.vt + ((size_t)b * p.dmodel + head * 64) * p.lp;

  const int q = (blockIdx.x * NW + wave) * 32 + r;
  const bool qok = q < p.Lq;
  const int qc = qok ? q : p.Lq - 1;
  AH8 qf[4];   // B operand of S^T = K Q^T: Q[query][16s + 8h .. +8]
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s].u = *reinterpret_cast<const uint4*>(Qp + (size_t)qc * p.ldq + 16 * s + 8 * h);

  // staging: K tile = 32 rows x 8 chunks of 16 B (256 chunks); V^T tile = 64 rows x 8 pieces of 8 B (512 pieces)
  constexpr int NT = NW * 64;
  constexpr int KP = 256 / NT, VP = 512 / NT;      // NT = 256: 1, 2;  NT = 128: 2, 4
  uint4 kr0, kr1 = {};
  uint2 vr0, vr1, vr2 = {}, vr3 = {};
#define ATH_LD_K(REG, I)                                                                                   \
  {                                                                                                        \
    const int c = t + (I) * NT;                                                                            \
    int row = k0_ + (c >> 3);                                                                              \
    if (row > p.Lk - 1) row = p.Lk - 1; /* masked after the QK^T product */                               \
    REG = *reinterpret_cast<const uint4*>(Kp + (size_t)row * p.ldk + (c & 7) * 8);                         \
  }
#define ATH_LD_V(REG, I)                                                                                   \
  {                                                                                                        \
    const int c = t + (I) * NT;                                                                            \
    REG = *reinterpret_cast<const uint2*>(Vp + (size_t)(c >> 3) * p.lp + k0_ + (c & 7) * 4); /* pad = 0 */ \
  }
#define ATH_LOAD_TILE(KT)                                     \
  {                                                           \
    const int k0_ = (KT) * 32;                                \
    ATH_LD_K(kr0, 0)                                          \
    if (KP > 1) ATH_LD_K(kr1, 1)                              \
    ATH_LD_V(vr0, 0) ATH_LD_V(vr1, 1)                         \
    if (VP > 2) { ATH_LD_V(vr2, 2) ATH_LD_V(vr3, 3) }         \
  }
#define ATH_ST_K(REG, I, BUF) { const int c = t + (I) * NT; *reinterpret_cast<uint4*>(&Ks[BUF][(c >> 3) * ATH_KS + (c & 7) * 16]) = REG; }
#define ATH_ST_V(REG, I, BUF) { const int c = t + (I) * NT; *reinterpret_cast<uint2*>(&Vs[BUF][(c >> 3) * ATH_VS + (c & 7) * 8]) = REG; }
#define ATH_STORE_TILE(BUF)                                           \
  {                                                                   \
    ATH_ST_K(kr0, 0, BUF)                                             \
    if (KP > 1) ATH_ST_K(kr1, 1, BUF)                                 \
    ATH_ST_V(vr0, 0, BUF) ATH_ST_V(vr1, 1, BUF)                       \
    if (VP > 2) { ATH_ST_V(vr2, 2, BUF) ATH_ST_V(vr3, 3, BUF) }       \
  }

  ah_f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m = -INFINITY, l = 0.f;

  const int nt = (p.Lk + 31) / 32;
  ATH_LOAD_TILE(0)
  ATH_STORE_TILE(0)
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nt) ATH_LOAD_TILE(kt + 1)
    // ---- S^T = K Q^T -------------------------------------------------------------------------------------
    ah_f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    const char* kb = &Ks[buf][r * ATH_KS + 16 * h];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      AH8 kf;
      kf.u = *reinterpret_cast<const uint4*>(kb + 32 * s);
      st = fe_mfma16(tag, kf.f, qf[s].f, st);
    }
    // ---- online softmax over this lane's query -----------------------------------------------------------
    const int kbase = kt * 32 + 4 * h;
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kbase + (e & 3) + 8 * (e >> 2);
      if (key >= p.Lk || (p.causal && key > q)) st[e] = -INFINITY;
      tmax = fmaxf(tmax, st[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);          // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = __expf(st[e] - mn); psum += st[e]; }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    // ---- O^T += V^T P^T: k-step s uses accumulator registers 8s .. 8s+7 = keys 16s + 8(j>>2) + 4h + (j&3) ------------
    const char* vb = &Vs[buf][r * ATH_VS + 8 * h];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      AH8 pf;
      pf.u = make_uint4(fe_pack2(tag, st[8 * s], st[8 * s + 1]), fe_pack2(tag, st[8 * s + 2], st[8 * s + 3]),
                        fe_pack2(tag, st[8 * s + 4], st[8 * s + 5]), fe_pack2(tag, st[8 * s + 6], st[8 * s + 7]));
      AH8 v0, v1;   // V^T[d = r (+32)][keys 16s + 4h .. +4 | 16s + 8 + 4h .. +4]
      const uint2 a0 = *reinterpret_cast<const uint2*>(vb + 32 * s), a1 = *reinterpret_cast<const uint2*>(vb + 32 * s + 16);
      const uint2 c0 = *reinterpret_cast<const uint2*>(vb + 32 * ATH_VS + 32 * s), c1 = *reinterpret_cast<const uint2*>(vb + 32 * ATH_VS + 32 * s + 16);
      v0.u = make_uint4(a0.x, a0.y, a1.x, a1.y);
      v1.u = make_uint4(c0.x, c0.y, c1.x, c1.y);
      o0 = fe_mfma16(tag, v0.f, pf.f, o0);
      o1 = fe_mfma16(tag, v1.f, pf.f, o1);
    }
    if (kt + 1 < nt) ATH_STORE_TILE(buf ^ 1)
    __syncthreads();
  }
  // ---- epilogue: O[q][head*64 + d] = O^T[d][q] / l + bv[d]; register e of tile dt is d = 32*dt + (e&3) + 8(e>>2) + 4h
  if (qok) {
    const float inv = 1.f / l;
    E* op = p.o + ((size_t)b * p.Lq + q) * p.ldo + head * 64;
    const float* bp = p.bv + head * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * h;
      const float4 b0 = *reinterpret_cast<const float4*>(bp + d0);
      const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 + d0);
      st4(op + d0, make_float4(o0[4 * g] * inv + b0.x, o0[4 * g + 1] * inv + b0.y, o0[4 * g + 2] * inv + b0.z, o0[4 * g + 3] * inv + b0.w));
      st4(op + 32 + d0, make_float4(o1[4 * g] * inv + b1.x, o1[4 * g + 1] * inv + b1.y, o1[4 * g + 2] * inv + b1.z, o1[4 * g + 3] * inv + b1.w));
    }
  }
}

template <class E>
static void launch_attention_half(const E* q, int ldq, const E* k, int ldk, const E* vt, int lp, const float* bv, E* o,
                                  int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s) {
  FE_CHECK(dmodel == H * 64, "attention kernel is built for head_dim 64 (d_model %d, %d heads)", dmodel, H);
  FE_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && lp % 4 == 0 && ldo % 4 == 0 && lp >= (Lk + 31) / 32 * 32, "attention(2-byte): strides");
  FE_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)bv) & 15) == 0 && (((uintptr_t)vt | (uintptr_t)o) & 7) == 0, "attention(2-byte): alignment");
  AttnParamsH<E> p{q, ldq, k, ldk, vt, lp, bv, o, ldo, B, H, Lq, Lk, dmodel, causal};
  const int w4 = (Lq + 127) / 128 * 128, w2 = (Lq + 63) / 64 * 64;
  if (w4 * 100 <= w2 * 108) {
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<E, 4>), dim3((Lq + 127) / 128, B * H), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<E, 2>), dim3((Lq + 63) / 64, B * H), dim3(128), 0, s, p);
  }
  FE_HIP(hipGetLastError());
}
void launch_attention(const bf16* q, int ldq, const bf16* k, int ldk, const bf16* vt, int lp, const float* bv, bf16* o,
                      int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s) {
  launch_attention_half(q, ldq, k, ldk, vt, lp, bv, o, ldo, B, H, Lq, Lk, dmodel, causal, s);
}
void launch_attention(const f16* q, int ldq, const f16* k, int ldk, const f16* vt, int lp, const float* bv, f16* o,
                      int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s) {
  launch_attention_half(q, ldq, k, ldk, vt, lp, bv, o, ldo, B, H, Lq, Lk, dmodel, causal, s);
}

}  // namespace fe

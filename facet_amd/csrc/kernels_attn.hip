// Fused multi-head attention for head_dim 64 on the fp32 matrix cores: O = softmax(Q K^T) V + bv, never materialising
// the score matrix (LDS-tiled MFMA QK^T / PV for the ViT blocks and the CFANet attention blocks).
// Stands behind torch.nn.MultiheadAttention inside open_clip's resblocks (reference call: model.encode_image,
// processing/scorer.py:662) and pyiqa's CFANet layers (models/pyiqa_scorer.py:212).
//
// One wave owns 32 queries; a workgroup (NW waves) shares 32-key K / V^T tiles through LDS.
//   S^T = K Q^T     (A = K tile rows = keys, B = Q fragment kept in registers, columns = queries): 32 MFMAs / tile
//   online softmax  each lane owns ONE query (accumulator column) and 16 of the tile's 32 keys; the other 16 sit in
//                   lane^32, so a row max / sum is 15 VALU ops + one cross-half shuffle
//   O^T += V^T P^T  the exponentiated S^T accumulator is used AS IS as the B operand (its register e holds key
//                   (e&3) + 8(e>>2) + 4*half, exactly the k-slot pairing of v_mfma_f32_32x32x2_f32); A = V^T tile rows = d
// Q is pre-scaled by 1/sqrt(64) in the projection epilogue; V^T comes straight from the role-swapped projection GEMM.
#include "fe_common.h"

namespace fe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_KS = 68;   // K tile row stride (floats): 64 + 4 -> conflict-free ds_read_b128 over 16 rows
constexpr int ATT_VS = 36;   // V^T tile row stride (floats): 32 + 4

struct AttnParams {
  const float* q; int ldq;     // [B*Lq][ldq], head h at column h*64
  const float* k; int ldk;     // [B*Lk][ldk]
  const float* vt; int lp;     // [B][d_model][lp]  (V transposed, zero padded to lp >= roundup32(Lk))
  const float* bv;             // [d_model] V bias, added to the output (softmax rows sum to 1)
  float* o; int ldo;           // [B*Lq][ldo]
  int B, H, Lq, Lk, dmodel;
  int causal;                  // 1: key j is visible to query i only if j <= i (CLIP text tower)
};

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) float Ks[2][32 * ATT_KS];
  __shared__ __attribute__((aligned(16))) float Vs[2][64 * ATT_VS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.H, head = bh - b * p.H;
  const float* Qp = p.q + (size_t)b * p.Lq * p.ldq + head * 64;
  const float* Kp = p.k + (size_t)b * p.Lk * p.ldk + head * 64;
  const float* Vp = p.vt + ((size_t)b * p.dmodel + head * 64) * p.lp;

  const int q = (blockIdx.x * NW + wave) * 32 + r;
  const bool qok = q < p.Lq;
  const int qc = qok ? q : p.Lq - 1;
  float4 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const float4*>(Qp + (size_t)qc * p.ldq + 8 * s + 4 * h);

  // staging coordinates: K tile = 32 rows x 16 chunks (512 chunks), V^T tile = 64 rows x 8 chunks (512 chunks)
  constexpr int NT = NW * 64;
  constexpr int KP = 512 / NT, VP = 512 / NT;
  // Prefetch registers are named scalars on purpose: as arrays (even with fully unrolled static indices) hipcc kept
  // them in scratch memory, which showed up as 88 MB/image of WRITE_SIZE in the profile.
  float4 kr0, kr1, kr2 = {}, kr3 = {}, vr0, vr1, vr2 = {}, vr3 = {};
#define ATT_LD_K(REG, I)                                                                                  \
  {                                                                                                       \
    const int c = t + (I) * NT;                                                                           \
    int row = k0_ + (c >> 4);                                                                             \
    if (row > p.Lk - 1) row = p.Lk - 1; /* masked after the QK^T product */                              \
    REG = *reinterpret_cast<const float4*>(Kp + (size_t)row * p.ldk + (c & 15) * 4);                      \
  }
#define ATT_LD_V(REG, I)                                                                                  \
  {                                                                                                       \
    const int c = t + (I) * NT;                                                                           \
    REG = *reinterpret_cast<const float4*>(Vp + (size_t)(c >> 3) * p.lp + k0_ + (c & 7) * 4); /* pad = 0 */ \
  }
#define ATT_LOAD_TILE(KT)                                    \
  {                                                          \
    const int k0_ = (KT) * 32;                               \
    ATT_LD_K(kr0, 0) ATT_LD_K(kr1, 1)                        \
    if (KP > 2) { ATT_LD_K(kr2, 2) ATT_LD_K(kr3, 3) }        \
    ATT_LD_V(vr0, 0) ATT_LD_V(vr1, 1)                        \
    if (VP > 2) { ATT_LD_V(vr2, 2) ATT_LD_V(vr3, 3) }        \
  }
#define ATT_ST_K(REG, I, BUF) { const int c = t + (I) * NT; *reinterpret_cast<float4*>(&Ks[BUF][(c >> 4) * ATT_KS + (c & 15) * 4]) = REG; }
#define ATT_ST_V(REG, I, BUF) { const int c = t + (I) * NT; *reinterpret_cast<float4*>(&Vs[BUF][(c >> 3) * ATT_VS + (c & 7) * 4]) = REG; }
#define ATT_STORE_TILE(BUF)                                          \
  {                                                                  \
    ATT_ST_K(kr0, 0, BUF) ATT_ST_K(kr1, 1, BUF)                      \
    if (KP > 2) { ATT_ST_K(kr2, 2, BUF) ATT_ST_K(kr3, 3, BUF) }      \
    ATT_ST_V(vr0, 0, BUF) ATT_ST_V(vr1, 1, BUF)                      \
    if (VP > 2) { ATT_ST_V(vr2, 2, BUF) ATT_ST_V(vr3, 3, BUF) }      \
  }

  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m = -INFINITY, l = 0.f;

  const int nt = (p.Lk + 31) / 32;
  ATT_LOAD_TILE(0)
  ATT_STORE_TILE(0)
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nt) ATT_LOAD_TILE(kt + 1)
    // ---- S^T = K Q^T -------------------------------------------------------------------------------------
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    const float* kb = &Ks[buf][r * ATT_KS + 4 * h];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 kf = *reinterpret_cast<const float4*>(kb + 8 * s);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[s].x, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[s].y, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[s].z, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[s].w, st, 0, 0, 0);
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
    // ---- O^T += V^T P^T ----------------------------------------------------------------------------------
    const float* vb = &Vs[buf][r * ATT_VS + 4 * h];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v0 = *reinterpret_cast<const float4*>(vb + 8 * g);
      const float4 v1 = *reinterpret_cast<const float4*>(vb + 32 * ATT_VS + 8 * g);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, st[4 * g + 0], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, st[4 * g + 0], o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, st[4 * g + 1], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, st[4 * g + 1], o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.z, st[4 * g + 2], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.z, st[4 * g + 2], o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.w, st[4 * g + 3], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.w, st[4 * g + 3], o1, 0, 0, 0);
    }
    if (kt + 1 < nt) ATT_STORE_TILE(buf ^ 1)
    __syncthreads();
  }
  // ---- epilogue: O[q][head*64 + d] = O^T[d][q] / l + bv[d]; register e of tile dt is d = 32*dt + (e&3) + 8(e>>2) + 4h
  if (qok) {
    const float inv = 1.f / l;
    float* op = p.o + ((size_t)b * p.Lq + q) * p.ldo + head * 64;
    const float* bp = p.bv + head * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * h;
      const float4 b0 = *reinterpret_cast<const float4*>(bp + d0);
      const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 + d0);
      *reinterpret_cast<float4*>(op + d0) = make_float4(o0[4 * g] * inv + b0.x, o0[4 * g + 1] * inv + b0.y,
                                                        o0[4 * g + 2] * inv + b0.z, o0[4 * g + 3] * inv + b0.w);
      *reinterpret_cast<float4*>(op + 32 + d0) = make_float4(o1[4 * g] * inv + b1.x, o1[4 * g + 1] * inv + b1.y,
                                                             o1[4 * g + 2] * inv + b1.z, o1[4 * g + 3] * inv + b1.w);
    }
  }
}

void launch_attention(const float* q, int ldq, const float* k, int ldk, const float* vt, int lp, const float* bv, float* o,
                      int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s) {
  FE_CHECK(dmodel == H * 64, "attention kernel is built for head_dim 64 (d_model %d, %d heads)", dmodel, H);
  FE_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && lp % 4 == 0 && ldo % 4 == 0 && lp >= (Lk + 31) / 32 * 32, "attention: strides");
  FE_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)bv | (uintptr_t)o) & 15) == 0, "attention: alignment");
  AttnParams p{q, ldq, k, ldk, vt, lp, bv, o, ldo, B, H, Lq, Lk, dmodel, causal};
  // 4 waves (128 queries) per workgroup when that wastes little; 2 waves for short / ragged sequences (CLIP: 257)
  const int w4 = (Lq + 127) / 128 * 128, w2 = (Lq + 63) / 64 * 64;
  if (w4 * 100 <= w2 * 108) {
    hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3((Lq + 127) / 128, B * H), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((Lq + 63) / 64, B * H), dim3(128), 0, s, p);
  }
  FE_HIP(hipGetLastError());
}

}  // namespace fe

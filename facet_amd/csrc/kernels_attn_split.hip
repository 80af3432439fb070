// Fused multi-head attention (head_dim 64) on split fp16 operands: Q, K, V and the probabilities each travel as an fp16 pair hi + lo
// (~22 significant bits), every product as its three leading terms (hh + lh + hl) on v_mfma_f32_32x32x16_f16 with fp32 accumulation,
// and the output leaves as a pair again. The attention of the split-operand CLIP tower (ClipModel::split3, model_clip.hip; reference:
// open_clip's resblocks behind model.encode_image, processing/scorer.py:662), whose error budget against the fp32 path is 1e-3 on
// the FINAL scores: with plain fp16 q / k / v / o the tower holds 4e-4 on the features, which an ill-conditioned aesthetic head
// turns into 1.7e-3. Same structure as kernels_attn_bf16.hip (one wave = 32 queries, 32-key tiles double-buffered in LDS, online
// softmax with one query per lane, the exponentiated S^T accumulator re-used as the B operand of O^T += V^T P^T): 12 matrix
// instructions per tile for S^T instead of 4, 12 for O^T instead of 4 - attention is 4 % of the tower's multiply-adds.
#include "fe_common.h"

namespace fe {

constexpr int ATS_KS = 144;   // K tile row stride in bytes (128 + 16)
constexpr int ATS_VS = 72;    // V^T tile row stride in bytes (64 + 8)

struct AttnSplitParams {
  const f16* q; const f16* k; int ld; int lo_off;      // rows [B*L][ld]: hi at column c, lo at column lo_off + c (q pre-scaled)
  const f16* vt_hi; const f16* vt_lo; int lp;          // [B][d_model][lp], zero padded
  f16* o; int ldo; int o_lo_off;                       // [B*Lq][ldo]: hi | lo
  int B, H, Lq, Lk, dmodel;
};
union AS8 { uint4 u; fe_v4f f; };

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_split_kernel(const AttnSplitParams p) {
  __shared__ __attribute__((aligned(16))) char Ks[2][2][32 * ATS_KS];      // [buffer][hi / lo]
  __shared__ __attribute__((aligned(16))) char Vs[2][2][64 * ATS_VS];
  const f16* const tag = nullptr;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.H, head = bh - b * p.H;
  const f16* Qp = p.q + (size_t)b * p.Lq * p.ld + head * 64;
  const f16* Kp = p.k + (size_t)b * p.Lk * p.ld + head * 64;
  const f16* Vh = p.vt_hi + ((size_t)b * p.dmodel + head * 64) * p.lp;
  const f16* Vl = p.vt_lo + ((size_t)b * p.dmodel + head * 64) * p.lp;
  const int q = (blockIdx.x * NW + wave) * 32 + r;
  const bool qok = q < p.Lq;
  const int qc = qok ? q : p.Lq - 1;
  AS8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s].u = *reinterpret_cast<const uint4*>(Qp + (size_t)qc * p.ld + 16 * s + 8 * h);
    ql[s].u = *reinterpret_cast<const uint4*>(Qp + (size_t)qc * p.ld + p.lo_off + 16 * s + 8 * h);
  }
  constexpr int NT = NW * 64;
  constexpr int KP = 256 / NT, VP = 512 / NT;
  uint4 kr[2][2];
  uint2 vr[2][4];
  auto load_tile = [&](int kt) {
    const int k0 = kt * 32;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int c = t + i * NT;
      int row = k0 + (c >> 3);
      if (row > p.Lk - 1) row = p.Lk - 1;
      kr[0][i] = *reinterpret_cast<const uint4*>(Kp + (size_t)row * p.ld + (c & 7) * 8);
      kr[1][i] = *reinterpret_cast<const uint4*>(Kp + (size_t)row * p.ld + p.lo_off + (c & 7) * 8);
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int c = t + i * NT;
      vr[0][i] = *reinterpret_cast<const uint2*>(Vh + (size_t)(c >> 3) * p.lp + k0 + (c & 7) * 4);
      vr[1][i] = *reinterpret_cast<const uint2*>(Vl + (size_t)(c >> 3) * p.lp + k0 + (c & 7) * 4);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
      for (int i = 0; i < KP; ++i) { const int c = t + i * NT; *reinterpret_cast<uint4*>(&Ks[buf][part][(c >> 3) * ATS_KS + (c & 7) * 16]) = kr[part][i]; }
#pragma unroll
      for (int i = 0; i < VP; ++i) { const int c = t + i * NT; *reinterpret_cast<uint2*>(&Vs[buf][part][(c >> 3) * ATS_VS + (c & 7) * 8]) = vr[part][i]; }
    }
  };
  fe_f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int nt = (p.Lk + 31) / 32;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nt) load_tile(kt + 1);
    fe_f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    const char* kbh = &Ks[buf][0][r * ATS_KS + 16 * h];
    const char* kbl = &Ks[buf][1][r * ATS_KS + 16 * h];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      AS8 kh, kl;
      kh.u = *reinterpret_cast<const uint4*>(kbh + 32 * s);
      kl.u = *reinterpret_cast<const uint4*>(kbl + 32 * s);
      st = fe_mfma16(tag, kl.f, qh[s].f, st);      // small terms first
      st = fe_mfma16(tag, kh.f, ql[s].f, st);
      st = fe_mfma16(tag, kh.f, qh[s].f, st);
    }
    const int kbase = kt * 32 + 4 * h;
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kbase + (e & 3) + 8 * (e >> 2);
      if (key >= p.Lk) st[e] = -INFINITY;
      tmax = fmaxf(tmax, st[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(m, tmax);
    const float alpha = expf(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = expf(st[e] - mn); psum += st[e]; }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    const char* vbh = &Vs[buf][0][r * ATS_VS + 8 * h];
    const char* vbl = &Vs[buf][1][r * ATS_VS + 8 * h];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      AS8 ph, pl;
      float lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) lo[j] = st[8 * s + j] - (float)fe_to_f16(st[8 * s + j]);
      ph.u = make_uint4(fe_pack2(tag, st[8 * s], st[8 * s + 1]), fe_pack2(tag, st[8 * s + 2], st[8 * s + 3]),
                        fe_pack2(tag, st[8 * s + 4], st[8 * s + 5]), fe_pack2(tag, st[8 * s + 6], st[8 * s + 7]));
      pl.u = make_uint4(fe_pack2(tag, lo[0], lo[1]), fe_pack2(tag, lo[2], lo[3]), fe_pack2(tag, lo[4], lo[5]), fe_pack2(tag, lo[6], lo[7]));
      auto frag = [&](const char* base) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(base + 32 * s), a1 = *reinterpret_cast<const uint2*>(base + 32 * s + 16);
        AS8 v; v.u = make_uint4(a0.x, a0.y, a1.x, a1.y);
        return v;
      };
      const AS8 v0h = frag(vbh), v1h = frag(vbh + 32 * ATS_VS), v0l = frag(vbl), v1l = frag(vbl + 32 * ATS_VS);
      o0 = fe_mfma16(tag, v0l.f, ph.f, o0); o0 = fe_mfma16(tag, v0h.f, pl.f, o0); o0 = fe_mfma16(tag, v0h.f, ph.f, o0);
      o1 = fe_mfma16(tag, v1l.f, ph.f, o1); o1 = fe_mfma16(tag, v1h.f, pl.f, o1); o1 = fe_mfma16(tag, v1h.f, ph.f, o1);
    }
    if (kt + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (qok) {
    const float inv = 1.f / l;
    f16* op = p.o + ((size_t)b * p.Lq + q) * p.ldo + head * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * h;
      const float4 a = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      const float4 c = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
      const float4 ah = make_float4((float)fe_to_f16(a.x), (float)fe_to_f16(a.y), (float)fe_to_f16(a.z), (float)fe_to_f16(a.w));
      const float4 ch = make_float4((float)fe_to_f16(c.x), (float)fe_to_f16(c.y), (float)fe_to_f16(c.z), (float)fe_to_f16(c.w));
      st4(op + d0, ah); st4(op + 32 + d0, ch);
      st4(op + p.o_lo_off + d0, make_float4(a.x - ah.x, a.y - ah.y, a.z - ah.z, a.w - ah.w));
      st4(op + p.o_lo_off + 32 + d0, make_float4(c.x - ch.x, c.y - ch.y, c.z - ch.z, c.w - ch.w));
    }
  }
}

void launch_attention_split(const f16* q, const f16* k, int ld, int lo_off, const f16* vt_hi, const f16* vt_lo, int lp, f16* o, int ldo, int o_lo_off,
                            int B, int H, int Lq, int Lk, int dmodel, hipStream_t s) {
  FE_CHECK(dmodel == H * 64 && ld % 8 == 0 && lo_off % 8 == 0 && lp % 4 == 0 && ldo % 4 == 0 && o_lo_off % 4 == 0 && lp >= (Lk + 31) / 32 * 32, "attention(split): geometry");
  FE_CHECK((((uintptr_t)q | (uintptr_t)k) & 15) == 0 && (((uintptr_t)vt_hi | (uintptr_t)vt_lo | (uintptr_t)o) & 7) == 0, "attention(split): alignment");
  AttnSplitParams p{q, k, ld, lo_off, vt_hi, vt_lo, lp, o, ldo, o_lo_off, B, H, Lq, Lk, dmodel};
  hipLaunchKernelGGL(attn_fwd_split_kernel<4>, dim3((Lq + 127) / 128, B * H), dim3(256), 0, s, p);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

// VLM tagger, slice 1: the text decoder of Qwen2.5-VL (SURVEY 8(f)-4 / BASELINE configs[4]) - prefill + single-token decode with a
// contiguous KV cache, greedy next-token selection.
//
// Stands behind reference models/vlm_tagger.py: `Qwen2_5_VLForConditionalGeneration.from_pretrained(..., dtype=torch.bfloat16)`
// (:163-184) and `self.model.generate(**inputs, max_new_tokens=..., do_sample=False)` (:250-259, :355-360). The arithmetic is
// transformers' (modeling_qwen2_5_vl.py: Qwen2_5_VLDecoderLayer = RMSNorm -> q/k/v projections with bias -> multimodal rotary
// embedding (sections 16/24/24 over the 64 frequency pairs of head_dim 128) -> grouped-query causal attention -> o_proj -> residual ->
// RMSNorm -> SwiGLU MLP -> residual; final RMSNorm; untied lm_head), restated here in the precision the reference loads: bf16 storage
// of every activation with the SAME rounding points as the bf16 torch modules (each Linear output, RMSNorm's normalised value before the
// weight multiply, both products of the rotary embedding, SiLU before the gate multiply, each residual sum, the logits), fp32
// accumulation inside every contraction. Parity: tests/test_vlm_gpu.py against vectors of the reference's own model class
// (tests/golden/make_vlm_golden.py) - pinned.
// Not in this slice: the vision tower (window attention, patch merger) and fp8 attention; image tokens enter as rows of `embeds`.
#include "engine.h"
#include <type_traits>
#include <cmath>

namespace fe {

__device__ __forceinline__ void h_unpack8_bf16(const uint4 u, float v[8]) {
  const bf16* tag = nullptr;
  fe_unpack2(tag, u.x, v[0], v[1]); fe_unpack2(tag, u.y, v[2], v[3]); fe_unpack2(tag, u.z, v[4], v[5]); fe_unpack2(tag, u.w, v[6], v[7]);
}

// ---- small row kernels ------------------------------------------------------------------------------------------------------------
// x[row][:] = E[tok[row]][:]   (bf16 table)
__global__ void vlm_embed_kernel(const int* __restrict__ tok, const bf16* __restrict__ E, bf16* __restrict__ x, int rows, int d, int vocab) {
  const size_t total = (size_t)rows * (d / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / (d / 8)), c = (int)(i % (d / 8)) * 8;
    int t = tok[row];
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    *reinterpret_cast<uint4*>(x + (size_t)row * d + c) = *reinterpret_cast<const uint4*>(E + (size_t)t * d + c);
  }
}

// transformers' RMSNorm in bf16: x32 = float(x); y = w * bf16(x32 * rsqrt(mean(x32^2) + eps)), the product rounded to bf16.
// One wave per row; d % 8 == 0.
__global__ void vlm_rmsnorm_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ w, bf16* __restrict__ y, int ldy, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    const bf16* xr = x + (size_t)row * ldx;
    float ss = 0.f;
    for (int i = lane * 4; i < d; i += 256) { const float4 v = ld4(xr + i); ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rs = rsqrtf(ss / (float)d + eps);
    bf16* yr = y + (size_t)row * ldy;
    for (int i = lane * 4; i < d; i += 256) {
      const float4 v = ld4(xr + i), g = ld4(w + i);
      st4(yr + i, make_float4(g.x * (float)(bf16)(v.x * rs), g.y * (float)(bf16)(v.y * rs), g.z * (float)(bf16)(v.z * rs), g.w * (float)(bf16)(v.w * rs)));
    }
  }
}

// Multimodal rotary embedding on the q and k heads of a fused QKV row block + the KV-cache append.
//   qkv: [rows][(nh + 2 nkv) * 128]; pos: [3][rows] (temporal, height, width position of every token; equal for text tokens)
//   q_out: [rows][nh * 128]; kc / vc: cache [B][nkv][max_seq][128], row `row` = (b, t) lands at position start + t.
// Dimension i of a head takes frequency i % 64 and the position component of its section (sections doubled over the two halves:
// [16, 24, 24, 16, 24, 24]); cos / sin are rounded to bf16 and x*cos, rotate_half(x)*sin and their sum are each rounded to bf16, as
// apply_multimodal_rotary_pos_emb does on bf16 tensors. One thread per (row, head, pair d < 64).
__global__ void vlm_rope_cache_kernel(const bf16* __restrict__ qkv, const int* __restrict__ pos, const float* __restrict__ inv_freq, bf16* __restrict__ q_out,
                                      bf16* __restrict__ kc, bf16* __restrict__ vc, int rows, int L, int nh, int nkv, int s0, int s1, int start, int max_seq,
                                      const int* __restrict__ start_dev) {
  if (start_dev) start = *start_dev;      // decode steps replayed from a captured graph: the cache length lives in device memory
  const int heads = nh + 2 * nkv;
  const size_t total = (size_t)rows * heads * 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i & 63), hd = (int)((i >> 6) % heads), row = (int)(i / ((size_t)heads * 64));
    const bf16* src = qkv + ((size_t)row * heads + hd) * 128;
    const int b = row / L, t = row - b * L;
    if (hd >= nh + nkv) {      // V: plain copy into the cache
      bf16* dst = vc + (((size_t)b * nkv + (hd - nh - nkv)) * max_seq + start + t) * 128;
      dst[d] = src[d]; dst[d + 64] = src[d + 64];
      continue;
    }
    const int comp = d < s0 ? 0 : (d < s0 + s1 ? 1 : 2);
    const float ang = (float)pos[(size_t)comp * rows + row] * inv_freq[d];
    const float c = (float)(bf16)cosf(ang), s = (float)(bf16)sinf(ang);
    const float x1 = (float)src[d], x2 = (float)src[d + 64];
    const bf16 o1 = (bf16)((float)(bf16)(x1 * c) + (float)(bf16)(-x2 * s));
    const bf16 o2 = (bf16)((float)(bf16)(x2 * c) + (float)(bf16)(x1 * s));
    bf16* dst = hd < nh ? q_out + ((size_t)row * nh + hd) * 128 : kc + (((size_t)b * nkv + (hd - nh)) * max_seq + start + t) * 128;
    dst[d] = o1; dst[d + 64] = o2;
  }
}

// h = bf16(bf16(silu(g)) * u)   (Qwen2MLP: act_fn(gate_proj(x)) * up_proj(x) on bf16 tensors)
__global__ void vlm_silu_mul_kernel(const bf16* g, const bf16* __restrict__ u, bf16* h, size_t n4) {      // h may be g
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = ld4(g + 4 * i), b = ld4(u + 4 * i);
    auto f = [](float x, float y) { return (float)(bf16)(x / (1.f + expf(-x))) * y; };
    st4(h + 4 * i, make_float4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w)));
  }
}
// x = bf16(x + y)
__global__ void vlm_add_kernel(bf16* __restrict__ x, const bf16* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = ld4(x + 4 * i), b = ld4(y + 4 * i);
    st4(x + 4 * i, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
  }
}
// last[b][:] = x[b * L + L - 1][:]
__global__ void vlm_last_rows_kernel(const bf16* __restrict__ x, bf16* __restrict__ last, int B, int L, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * d) return;
  const int b = i / d, c = i - b * d;
  last[i] = x[((size_t)b * L + L - 1) * d + c];
}

// ---- prefill attention: causal, grouped-query, head_dim 128, on the bf16 matrix cores --------------------------------------------------
// One wave owns 32 queries; a workgroup (4 waves = 128 queries) shares 32-key K / V tiles through LDS. Same scheme as
// kernels_attn_bf16.hip (S^T = K Q^T with one query per accumulator column, online softmax per lane, the exponentiated accumulator
// rounded to bf16 IS the B operand of O^T += V^T P^T), with 8 k-steps per S tile and four 32-row d-tiles of O. K / V come straight from
// the cache rows ([pos][128]); V is transposed on its way into LDS (two-byte scatter), K rows are copied as they are.
constexpr int VA_KS = 272;      // K tile row stride in bytes (256 + 16: conflict-free 16-byte reads over rows distinct mod 16... )
constexpr int VA_VS = 72;       // V^T tile row stride in bytes (64 + 8)
struct VlmAttnParams {
  const bf16* q; int ldq;       // [B*Lq][nh*128]
  const bf16* kc; const bf16* vc;   // caches [B][nkv][max_seq][128]
  bf16* o; int ldo;             // [B*Lq][nh*128]
  int B, nh, nkv, Lq, Lk, max_seq, qpos0;      // query i sits at sequence position qpos0 + i and sees keys 0 .. qpos0 + i
  float scale;
};
union VA8 { uint4 u; fe_v4f f; };

__global__ __launch_bounds__(256, 2) void vlm_attn_prefill_kernel(const VlmAttnParams p) {
  __shared__ __attribute__((aligned(16))) char Ks[2][32 * VA_KS];
  __shared__ __attribute__((aligned(16))) char Vs[2][128 * VA_VS];
  const bf16* const tag = nullptr;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.nh, head = bh - b * p.nh, kvh = head / (p.nh / p.nkv);
  const bf16* Qp = p.q + (size_t)b * p.Lq * p.ldq + head * 128;
  const bf16* Kp = p.kc + ((size_t)b * p.nkv + kvh) * p.max_seq * 128;
  const bf16* Vp = p.vc + ((size_t)b * p.nkv + kvh) * p.max_seq * 128;
  const int q = (blockIdx.x * 4 + wave) * 32 + r;
  const bool qok = q < p.Lq;
  const int qc = qok ? q : p.Lq - 1;
  const int qpos = p.qpos0 + q;
  VA8 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s].u = *reinterpret_cast<const uint4*>(Qp + (size_t)qc * p.ldq + 16 * s + 8 * h);
  // this workgroup's queries end at position qpos0 + (blockIdx.x + 1) * 128 - 1: later keys are masked for all of them
  const int kend = min(p.Lk, p.qpos0 + (int)(blockIdx.x + 1) * 128);
  const int nt = (kend + 31) / 32;
  uint4 kr[2], vr[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = t + i * 256;                 // 512 chunks of 16 B: key = c >> 4, d = (c & 15) * 8
      int key = kt * 32 + (c >> 4);
      if (key > p.Lk - 1) key = p.Lk - 1;        // masked after QK^T
      kr[i] = *reinterpret_cast<const uint4*>(Kp + (size_t)key * 128 + (c & 15) * 8);
      vr[i] = *reinterpret_cast<const uint4*>(Vp + (size_t)key * 128 + (c & 15) * 8);
    }
  };
  auto store_tile = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = t + i * 256, key = c >> 4, d0 = (c & 15) * 8;
      *reinterpret_cast<uint4*>(&Ks[buf][key * VA_KS + d0 * 2]) = kr[i];
      const bool live = kt * 32 + key < p.Lk;   // keys past Lk contribute zero rows of V (their probabilities are zero anyway)
      const unsigned w[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned short v = live ? (unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu)) : (unsigned short)0;
        *reinterpret_cast<unsigned short*>(&Vs[buf][(d0 + e) * VA_VS + key * 2]) = v;
      }
    }
  };
  fe_f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nt) load_tile(kt + 1);
    fe_f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    const char* kb = &Ks[buf][r * VA_KS + 16 * h];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      VA8 kf;
      kf.u = *reinterpret_cast<const uint4*>(kb + 32 * s);
      st = fe_mfma16(tag, kf.f, qf[s].f, st);
    }
    const int kbase = kt * 32 + 4 * h;
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kbase + (e & 3) + 8 * (e >> 2);
      st[e] = (key >= p.Lk || key > qpos) ? -INFINITY : st[e] * p.scale;
      tmax = fmaxf(tmax, st[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(m, tmax);
    const float msafe = mn == -INFINITY ? 0.f : mn;      // a query row whose keys so far are all masked (rows past Lq only)
    const float alpha = __expf(m - msafe);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = __expf(st[e] - msafe); psum += st[e]; }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    const char* vb = &Vs[buf][r * VA_VS + 8 * h];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      VA8 pf;
      pf.u = make_uint4(fe_pack2(tag, st[8 * s], st[8 * s + 1]), fe_pack2(tag, st[8 * s + 2], st[8 * s + 3]),
                        fe_pack2(tag, st[8 * s + 4], st[8 * s + 5]), fe_pack2(tag, st[8 * s + 6], st[8 * s + 7]));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(vb + dt * 32 * VA_VS + 32 * s), a1 = *reinterpret_cast<const uint2*>(vb + dt * 32 * VA_VS + 32 * s + 16);
        VA8 v;
        v.u = make_uint4(a0.x, a0.y, a1.x, a1.y);
        o[dt] = fe_mfma16(tag, v.f, pf.f, o[dt]);
      }
    }
    if (kt + 1 < nt) store_tile(buf ^ 1, kt + 1);
    __syncthreads();
  }
  if (qok) {
    const float inv = 1.f / l;
    bf16* op = p.o + ((size_t)b * p.Lq + q) * p.ldo + head * 128;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(op + dt * 32 + 8 * g + 4 * h, make_float4(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv));
  }
}

// x[index[i]][:] = rows[i][:]   (image embeddings into the rows of their <|image_pad|> tokens)
__global__ void vlm_put_rows_kernel(bf16* __restrict__ x, const bf16* __restrict__ rows, const int* __restrict__ index, int n, int d) {
  const size_t total = (size_t)n * (d / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (d / 8)), c = (int)(i % (d / 8)) * 8;
    *reinterpret_cast<uint4*>(x + (size_t)index[r] * d + c) = *reinterpret_cast<const uint4*>(rows + (size_t)r * d + c);
  }
}

// ---- decode GEMV: y[m][n] = sum_k x[m][k] w[n][k] (+ bias) for M <= 4 sequences - the shape of every projection of a decode step at the
// reference's batch sizes (vlm_batch_size 2, models/vlm_tagger.py:76). Pure weight streaming: one wave per TWO output columns, 16-byte
// loads (8 weights per lane and row), the M activation rows come from L1 / L2; fp32 accumulation, one rounding to bf16 (or fp32 out
// for the logits). HBM-bound: N * K * 2 bytes per launch.
template <int MR, class TO>
__global__ __launch_bounds__(256) void vlm_gemv_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                       TO* __restrict__ y, int ldy, int M, int N, int K) {
  // one workgroup = two output columns; its four waves take interleaved 512-column steps of K (so even the 3584-column projections put
  // 7 waves on every SIMD) and meet in LDS
  __shared__ float red[4][MR][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 2;
  const bool two = n0 + 1 < N;
  const bf16* w0 = w + (size_t)n0 * ldw;
  const bf16* w1 = w + (size_t)(two ? n0 + 1 : n0) * ldw;
  float a0[MR], a1[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { a0[m] = 0.f; a1[m] = 0.f; }
  // products on v_dot2c_f32_bf16 (two bf16 pairs per instruction, fp32 accumulate): no unpacking of either operand - with it the
  // 2- and 4-sequence steps were bound by the vector ALU, not by the weight stream
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  auto dot8 = [](const uint4 p, const uint4 q, float acc) __attribute__((always_inline)) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p.x), __builtin_bit_cast(bf2, q.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p.y), __builtin_bit_cast(bf2, q.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p.z), __builtin_bit_cast(bf2, q.z), acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p.w), __builtin_bit_cast(bf2, q.w), acc, false);
  };
  auto fma8 = [&](const uint4 wa, const uint4 wb, const int k) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < M) {
        const uint4 xs = *reinterpret_cast<const uint4*>(x + (size_t)m * ldx + k);
        a0[m] = dot8(xs, wa, a0[m]);
        a1[m] = dot8(xs, wb, a1[m]);
      }
    }
  };
  typedef unsigned v4u __attribute__((ext_vector_type(4)));      // weights are read once per step: streamed past the caches
  auto ldw8 = [](const bf16* ptr) __attribute__((always_inline)) {
    const v4u a = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(ptr));
    return make_uint4(a[0], a[1], a[2], a[3]);
  };
  int k = wave * 512 + lane * 8;
  for (; k + 3 * 2048 < K; k += 4 * 2048) {      // four steps per trip, their eight weight loads issued before the first is used
    uint4 wa[4], wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wa[j] = ldw8(w0 + k + 2048 * j); wb[j] = ldw8(w1 + k + 2048 * j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) fma8(wa[j], wb[j], k + 2048 * j);
  }
  for (; k + 2048 < K; k += 2 * 2048) {
    const uint4 p0 = ldw8(w0 + k), q0 = ldw8(w1 + k), p1 = ldw8(w0 + k + 2048), q1 = ldw8(w1 + k + 2048);
    fma8(p0, q0, k); fma8(p1, q1, k + 2048);
  }
  for (; k < K; k += 2048) fma8(ldw8(w0 + k), ldw8(w1 + k), k);
#pragma unroll
  for (int m = 0; m < MR; ++m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0[m] += __shfl_xor(a0[m], o); a1[m] += __shfl_xor(a1[m], o); }
  }
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < MR; ++m) { red[wave][m][0] = a0[m]; red[wave][m][1] = a1[m]; }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 2 * MR) {
    const int m = t >> 1, c = t & 1;
    if (m < M && (c == 0 || two)) {
      const float v = (red[0][m][c] + red[1][m][c]) + (red[2][m][c] + red[3][m][c]) + (bias ? bias[n0 + c] : 0.f);
      stf(y + (size_t)m * ldy + n0 + c, v);
    }
  }
}
template <class TO>
static void vlm_gemv(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, TO* y, int ldy) {
  FE_CHECK(w.wh && w.hprec == PREC_BF16 && w.KpH % 8 == 0 && ldx % 8 == 0 && M >= 1 && M <= 4 && !w.scale, "vlm_gemv: unsupported layer");
  const int K = w.CinPadH, blocks = (w.Cout + 1) / 2;
  if (M == 1) hipLaunchKernelGGL((vlm_gemv_kernel<1, TO>), dim3(blocks), dim3(256), 0, c.stream, x, ldx, (const bf16*)w.wh, w.KpH, (const float*)w.shift, y, ldy, M, w.Cout, K);
  else if (M == 2) hipLaunchKernelGGL((vlm_gemv_kernel<2, TO>), dim3(blocks), dim3(256), 0, c.stream, x, ldx, (const bf16*)w.wh, w.KpH, (const float*)w.shift, y, ldy, M, w.Cout, K);
  else hipLaunchKernelGGL((vlm_gemv_kernel<4, TO>), dim3(blocks), dim3(256), 0, c.stream, x, ldx, (const bf16*)w.wh, w.KpH, (const float*)w.shift, y, ldy, M, w.Cout, K);
  FE_HIP(hipGetLastError());
  c.flops_accum += 2.0 * M * (double)w.Cin * w.Cout;
}
// ---- decode GEMM for 5 .. 32 sequences: the weight rows are the A operand of v_mfma_f32_32x32x16_bf16 (32 output columns per wave),
// the activation rows the B operand (batch rows padded to 32 with zeros); every weight byte is read once per step. A workgroup takes
// FOUR column tiles (one per wave) over one K range (grid.y = the K split), in stages of 128 K-elements.
//   * Weights reach the matrix fragments through a wave-private LDS image: the fragment layout wants 16 bytes of each of 32 rows per
//     load instruction - 64 sectors touched for 1 KiB, every 64-byte sector fetched four times from L2 by the four instructions that share
//     it (measured: 1.6 TB/s with nontemporal loads, 2.1 with cached ones). The loads instead take 4 rows x 256 contiguous bytes per
//     instruction (whole sectors, once), the registers go to LDS (16-byte slots XORed with the row number: conflict-free for the writes'
//     8-lane groups and the fragment reads' 16-lane groups) and come back as fragments. Loads run one stage ahead in registers.
//   * The activation slice of a stage (32 rows x 256 B) is staged once per workgroup, double-buffered, for all four waves - fetched per
//     wave from L2 it was half of the load traffic of a "weight-streaming" kernel.
// Inside a 64-element block of K lane (r, h) owns elements 32 h .. 32 h + 31 of its row (four fragments; the contraction order is free as
// long as both operands use it). Partial sums [split][M][N] in fp32, added in a fixed order by the finishing kernel (no atomics: the
// same token ids on every run).
constexpr int VLM_G32_XP = 272;      // bytes per staged activation row: 128 elements + 16 (conflict-free 16-byte reads at stride 1 row)
// (w2 / part2: a second weight matrix of the same shape on the same activations - the gate and up projections of the MLP as ONE launch:
// workgroup columns [cols1, 2 cols1) take it)
__global__ __launch_bounds__(256) void vlm_gemm32_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ w, int ldw, float* __restrict__ part, int M, int N,
                                                         int K, int stages_per_wg, const bf16* __restrict__ w2, float* __restrict__ part2, int cols1) {
  __shared__ __attribute__((aligned(16))) char ws[4][32 * 256];
  __shared__ __attribute__((aligned(16))) char xs[2][32 * VLM_G32_XP];
  const bf16* const tag = nullptr;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
  const int ntiles = (N + 31) / 32;
  int bx = blockIdx.x;
  if (bx >= cols1) { bx -= cols1; w = w2; part = part2; }
  const int nt = min(bx * 4 + wave, ntiles - 1);      // a workgroup's waves past the last tile repeat it (and do not store)
  const int n0 = nt * 32;
  const int nst = K / 128;
  const int s0 = blockIdx.y * stages_per_wg, s1 = min(s0 + stages_per_wg, nst);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  // weight piece i of a stage: row 4 i + lane / 16 of the tile, 16 bytes at 16 (lane % 16) of the row's 256
  const int wrow = lane >> 4, wpc = lane & 15;
  const bf16* wsrc[8];
  int wdst[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + wrow, nr = min(n0 + row, N - 1);      // rows past N: a valid row, dropped at the store
    wsrc[i] = w + (size_t)nr * ldw + wpc * 8;
    wdst[i] = row * 256 + ((wpc ^ (row & 15)) << 4);
  }
  auto fetch_w = [&](const int st, v4u (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const v4u*>(wsrc[i] + (size_t)st * 128);
  };
  // activation piece i (of 2) of a stage: row (t + 256 i) / 16, 16 bytes at 16 ((t + 256 i) % 16)
  auto stage_x = [&](const int st, char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = t + 256 * i, row = p >> 4, col = (p & 15) * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row < M) v = *reinterpret_cast<const uint4*>(x + (size_t)row * ldx + (size_t)st * 128 + col);
      *reinterpret_cast<uint4*>(dst + row * VLM_G32_XP + col * 2) = v;
    }
  };
  fe_f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  v4u wr[2][8];
  char* const wl = ws[wave];
  if (s0 < s1) { fetch_w(s0, wr[0]); stage_x(s0, xs[0]); }
  __syncthreads();
  // (unrolled by two so the register sets and the activation buffers are compile-time choices)
  for (int st = s0; st < s1; st += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int cur = st + u;
      if (cur < s1) {
        if (cur + 1 < s1) fetch_w(cur + 1, wr[u ^ 1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<v4u*>(wl + wdst[i]) = wr[u][i];
        if (cur + 1 < s1) stage_x(cur + 1, xs[u ^ 1]);      // the other buffer: its readers passed the barrier one stage ago
        const char* const xb = xs[u] + r * VLM_G32_XP + 64 * h;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int pc = kb * 8 + h * 4 + q;      // the 16-byte piece of the row this fragment is
            const fe_v4f a = *reinterpret_cast<const fe_v4f*>(wl + r * 256 + ((pc ^ (r & 15)) << 4));
            acc = fe_mfma16(tag, a, *reinterpret_cast<const fe_v4f*>(xb + kb * 128 + 16 * q), acc);
          }
      }
      __syncthreads();
    }
  }
  // acc: lane (r, h) holds column m = r (batch row) of the 32 x 32 tile, rows n = (e & 3) + 8 (e >> 2) + 4 h
  if (bx * 4 + wave < ntiles && r < M) {
    float* out = part + ((size_t)blockIdx.y * M + r) * N + n0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int nn = (e & 3) + 8 * (e >> 2) + 4 * h;
      if (n0 + nn < N) out[nn] = acc[e];
    }
  }
}
template <class TO>
__global__ void vlm_gemm32_finish_kernel(const float* __restrict__ part, int splits, int M, int N, const float* __restrict__ bias, TO* __restrict__ y, int ldy) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
  float v = bias ? bias[n] : 0.f;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[((size_t)k * M + m) * N + n];
  stf(y + (size_t)m * ldy + n, s + v);
}
// launches the partial products of y = x W^T into `part` ([splits][M][N] fp32, taken from the arena: the CALLER rewinds); returns the split count
// `direct` (optional): where the result may be written straight when the layer needs no K split - a single split's partial sums ARE the
// product (the 152064-row lm_head: 1188 workgroup columns already fill the chip), so its finishing pass is skipped by the caller
static int vlm_gemm32_partials(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float** part, const ConvW* w2 = nullptr, float** part2 = nullptr,
                               float* direct = nullptr) {
  const int K = w.CinPadH, N = w.Cout;
  FE_CHECK(w.wh && w.hprec == PREC_BF16 && !w.scale && K % 128 == 0 && w.KpH % 8 == 0 && ldx % 8 == 0 && M >= 1 && M <= 32, "vlm_gemm32: unsupported layer");
  FE_CHECK(!w2 || (w2->wh && w2->hprec == PREC_BF16 && !w2->scale && w2->CinPadH == K && w2->Cout == N && w2->KpH == w.KpH && part2), "vlm_gemm32: the paired matrix differs in shape");
  const int ntiles = (N + 31) / 32, cols = (ntiles + 3) / 4, nst = K / 128, allcols = w2 ? 2 * cols : cols;
  // ~3 workgroups per CU stream from all of HBM; the partial sums of a split cost 4 M N bytes each way: no more splits than that needs,
  // and at least 4 stages per workgroup
  int splits = std::max(1, std::min(nst / 4, (768 + allcols - 1) / allcols));
  const int per = (nst + splits - 1) / splits;
  splits = (nst + per - 1) / per;
  *part = splits == 1 && direct ? direct : c.arena.array<float>((size_t)splits * M * N);
  if (w2) *part2 = c.arena.array<float>((size_t)splits * M * N);
  hipLaunchKernelGGL(vlm_gemm32_kernel, dim3(allcols, splits), dim3(256), 0, c.stream, x, ldx, (const bf16*)w.wh, w.KpH, *part, M, N, K, per,
                     w2 ? (const bf16*)w2->wh : (const bf16*)nullptr, w2 ? *part2 : (float*)nullptr, cols);
  FE_HIP(hipGetLastError());
  const double fl = 2.0 * M * (double)w.Cin * N * (w2 ? 2 : 1);
  c.flops_accum += fl; c.flops_half += fl;
  return splits;
}
template <class TO>
static void vlm_gemm32(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, TO* y, int ldy) {
  const size_t mark = c.arena.mark();
  float* part = nullptr;
  float* direct = nullptr;
  if constexpr (std::is_same<TO, float>::value) { if (!w.shift && ldy == w.Cout) direct = y; }
  const int splits = vlm_gemm32_partials(c, w, x, ldx, M, &part, nullptr, nullptr, direct), N = w.Cout;
  if (!(direct && part == direct)) {
    hipLaunchKernelGGL((vlm_gemm32_finish_kernel<TO>), dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, c.stream, (const float*)part, splits, M, N,
                       (const float*)w.shift, y, ldy);
    FE_HIP(hipGetLastError());
  }
  c.arena.rewind(mark);
}
static bool vlm_uses_gemm32(const ConvW& w, int M) {
  static const bool no32 = getenv("FE_VLM_NO_GEMM32") != nullptr;      // A/B hook
  return M > 3 && M <= 32 && w.CinPadH % 128 == 0 && !no32;      // measured: 4 sequences 0.94 ms / step here, 1.02 through the GEMV
}
// Finishing passes of the decode GEMM fused with what follows them (a 32-sequence step was ~30 launches per layer of which the 5-us ones -
// split sums, residual sums, norms, the gate product - were a quarter of the time). One workgroup per sequence row.
//   x = bf16(x + bf16(sum_k part + bias)); n = RMSNorm(x) * w   (w == nullptr: the sum only)
__global__ __launch_bounds__(1024) void vlm_finish_add_rmsnorm_kernel(const float* __restrict__ part, int splits, int M, int d, const float* __restrict__ bias, bf16* __restrict__ x,
                                                                     const bf16* __restrict__ w, bf16* __restrict__ n, float eps) {
  __shared__ float red[16];
  const int row = blockIdx.x, t = threadIdx.x;      // 1024 threads: with 256 a row of 25 splits was 100 dependent-latency loads per thread
  bf16* const xr = x + (size_t)row * d;
  float ss = 0.f;
  for (int i = t * 4; i < d; i += 4096) {
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight loads in flight, added in split order (the order of the stand-alone finishing pass: same bits)
    for (int k0 = 0; k0 < splits; k0 += 8) {
      float4 p4[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        p4[j] = k0 + j < splits ? *reinterpret_cast<const float4*>(part + ((size_t)(k0 + j) * M + row) * d + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < splits) { sum.x += p4[j].x; sum.y += p4[j].y; sum.z += p4[j].z; sum.w += p4[j].w; }
    }
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + i); sum.x += b4.x; sum.y += b4.y; sum.z += b4.z; sum.w += b4.w; }
    const float4 a = ld4(xr + i);
    // the projection's output is a bf16 tensor, and so is the residual stream
    const float4 v = make_float4((float)(bf16)(a.x + (float)(bf16)sum.x), (float)(bf16)(a.y + (float)(bf16)sum.y), (float)(bf16)(a.z + (float)(bf16)sum.z),
                                 (float)(bf16)(a.w + (float)(bf16)sum.w));
    st4(xr + i, v);
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (!w) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((t & 63) == 0) red[t >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += red[k];
  const float rs = rsqrtf(tot / (float)d + eps);
  bf16* const nr = n + (size_t)row * d;
  for (int i = t * 4; i < d; i += 4096) {
    const float4 v = ld4(xr + i), g = ld4(w + i);      // (this thread's own stores of the first loop)
    st4(nr + i, make_float4(g.x * (float)(bf16)(v.x * rs), g.y * (float)(bf16)(v.y * rs), g.z * (float)(bf16)(v.z * rs), g.w * (float)(bf16)(v.w * rs)));
  }
}
//   h = bf16(silu(bf16(sum_k pg))) * bf16(sum_k pu)   (the two projections of the SwiGLU MLP, same split count)
__global__ void vlm_finish_silu_mul_kernel(const float* __restrict__ pg, const float* __restrict__ pu, int splits, size_t mn4, bf16* __restrict__ hout) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn4; i += (size_t)gridDim.x * blockDim.x) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), u = g;
    for (int k = 0; k < splits; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(pg + (k * mn4 + i) * 4), b = *reinterpret_cast<const float4*>(pu + (k * mn4 + i) * 4);
      g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w;
      u.x += b.x; u.y += b.y; u.z += b.z; u.w += b.w;
    }
    auto f = [](float x, float y) { x = (float)(bf16)x; y = (float)(bf16)y; return (float)(bf16)(x / (1.f + expf(-x))) * y; };
    st4(hout + 4 * i, make_float4(f(g.x, u.x), f(g.y, u.y), f(g.z, u.z), f(g.w, u.w)));
  }
}
// y = x W^T (+ b) in bf16 for any row count: the streaming GEMV for up to 4 rows, the weight-streaming matrix-core GEMM up to 32, the
// shared layer wrapper above that
static void vlm_linear(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, bf16* y, int ldy) {
  if (M <= 4 && !vlm_uses_gemm32(w, M)) vlm_gemv(c, w, x, ldx, M, y, ldy);
  else if (vlm_uses_gemm32(w, M)) vlm_gemm32(c, w, x, ldx, M, y, ldy);
  else linear_forward(c, w, x, ldx, M, y, ldy, ACT_NONE);
}

// device-resident decode state (graph replay): after a step, the chosen tokens become the next step's input, every position and the
// cache length advance by one, and the tokens are appended to the output table [step][B]
__global__ void vlm_advance_kernel(const int* __restrict__ next, int* __restrict__ tok, int* __restrict__ pos, int* __restrict__ len, int* __restrict__ step,
                                   int* __restrict__ out, int B) {
  const int b = threadIdx.x;
  const int st = *step;
  if (b < B) {
    const int t = next[b];
    tok[b] = t;
    out[(size_t)st * B + b] = t;
    pos[b] += 1; pos[B + b] += 1; pos[2 * B + b] += 1;
  }
  __syncthreads();
  if (b == 0) { *len += 1; *step = st + 1; }
}

// x = bf16(x + y) and, in the same pass, n = RMSNorm(x) * w (w == nullptr: the sum only). One wave per row: a decode step is a chain of
// ~12 launches per layer whose small ones cost their launch latency, so the residual sum rides with the norm that follows it.
__global__ void vlm_add_rmsnorm_kernel(bf16* __restrict__ x, const bf16* __restrict__ y, const bf16* __restrict__ w, bf16* __restrict__ n, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    bf16* xr = x + (size_t)row * d;
    const bf16* yr = y + (size_t)row * d;
    float ss = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
      const float4 a = ld4(xr + i), b = ld4(yr + i);
      // the sum is rounded to bf16 first (the residual stream is a bf16 tensor), and the norm reads the rounded value
      const float4 v = make_float4((float)(bf16)(a.x + b.x), (float)(bf16)(a.y + b.y), (float)(bf16)(a.z + b.z), (float)(bf16)(a.w + b.w));
      st4(xr + i, v);
      ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (!w) continue;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rs = rsqrtf(ss / (float)d + eps);
    bf16* nr = n + (size_t)row * d;
    for (int i = lane * 4; i < d; i += 256) {
      const float4 v = ld4(xr + i), g = ld4(w + i);      // (this lane's own stores of the first loop)
      st4(nr + i, make_float4(g.x * (float)(bf16)(v.x * rs), g.y * (float)(bf16)(v.y * rs), g.z * (float)(bf16)(v.z * rs), g.w * (float)(bf16)(v.w * rs)));
    }
  }
}

// argmax over a vocabulary row in two launches (one 256-thread block walked 152064 logits in 180 us): 64 chunks per row, then one wave
// per row over the 64 partial results. Logits are rounded to bf16 in place first; ties go to the lowest index (torch.argmax).
constexpr int VLM_AM_CHUNKS = 64;
__global__ void vlm_argmax_part_kernel(float* __restrict__ lg, int vocab, float* __restrict__ pv, int* __restrict__ pi) {
  float* row = lg + (size_t)blockIdx.y * vocab;
  const int per = (vocab + VLM_AM_CHUNKS - 1) / VLM_AM_CHUNKS, i0 = blockIdx.x * per, i1 = min(i0 + per, vocab);
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float v = (float)(bf16)row[i];
    row[i] = v;
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
  __shared__ float sv[256]; __shared__ int si[256];
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float v = sv[threadIdx.x + o]; const int j = si[threadIdx.x + o];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && j < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = j; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { pv[blockIdx.y * VLM_AM_CHUNKS + blockIdx.x] = sv[0]; pi[blockIdx.y * VLM_AM_CHUNKS + blockIdx.x] = si[0]; }
}
__global__ void vlm_argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int* __restrict__ next) {
  const int b = blockIdx.x, t = threadIdx.x;      // 64 threads = one wave
  float v = pv[b * VLM_AM_CHUNKS + t]; int j = pi[b * VLM_AM_CHUNKS + t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o); const int j2 = __shfl_xor(j, o);
    if (v2 > v || (v2 == v && j2 < j)) { v = v2; j = j2; }
  }
  if (t == 0) next[b] = j;
}

// Decode attention, grouped-query aware: workgroup (sequence b, KV head kvh, chunk sp) takes keys [128 sp, 128 sp + 128) for ALL nh / nkv
// query heads that share the KV head - the first split form ran one workgroup per QUERY head, so every key / value row crossed L2 -> L1
// seven times at the 7B geometry, and it read K one row per lane (64 rows x 16 bytes per instruction: each 64-byte sector fetched four
// times): 54 us per layer at 32 sequences x 528 keys, a fifth of the step. Here 16 lanes share a key (1 KiB contiguous per instruction),
// the G dot products of a key come from one load, and a value row is multiplied into G outputs from one load. Writes the unnormalised
// outputs, running maxima and sums per (head, chunk); vlm_attn_combine_kernel merges the chunks. Lk by value or, for graph replay, from
// device memory (then the grid covers the whole cache and chunks past Lk write the neutral element).
// sum over the 16 lanes of a DPP row, result in every lane: four VALU adds with lane-permute modifiers (quad xor 1, quad xor 2, mirror of 8,
// mirror of 16) - __shfl_xor goes through the LDS crossbar (ds_bpermute), ~20x the latency, and the score phase needs 28 sums per step
__device__ __forceinline__ float vlm_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));       // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));      // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));      // row_mirror
  return v;
}
constexpr int VLM_DEC_CHUNK = 128;
template <int G>
__global__ __launch_bounds__(256, 3) void vlm_attn_decode_gqa_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kc, const bf16* __restrict__ vc, float* __restrict__ po,
                                                                  float* __restrict__ pm, float* __restrict__ pl, int nh, int nkv, int Lk, int max_seq, float scale,
                                                                  const int* __restrict__ len_dev, int nsplit) {
  __shared__ float sc[G][VLM_DEC_CHUNK];      // scores, then probabilities
  __shared__ float part[4][G][128];
  if (len_dev) Lk = *len_dev + 1;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bk = blockIdx.x, sp = blockIdx.y;
  const int b = bk / nkv, kvh = bk - b * nkv, head0 = kvh * G;
  const int k0 = sp * VLM_DEC_CHUNK, k1 = min(k0 + VLM_DEC_CHUNK, Lk);
  if (k0 >= Lk) {      // a chunk past the cache length: neutral element of the merge
    if (t < G) { pm[((size_t)b * nh + head0 + t) * nsplit + sp] = -INFINITY; pl[((size_t)b * nh + head0 + t) * nsplit + sp] = 0.f; }
    return;
  }
  const bf16* const K = kc + ((size_t)b * nkv + kvh) * max_seq * 128;
  const bf16* const V = vc + ((size_t)b * nkv + kvh) * max_seq * 128;
  // ---- scores: 16 lanes per key (8 dims each), 4 keys per wave instruction, 32 keys per wave ---------------------------------------------
  {
    const int sub = lane & 15, kk = lane >> 4;
    float qf[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) h_unpack8_bf16(*reinterpret_cast<const uint4*>(q + ((size_t)b * nh + head0 + g) * 128 + 8 * sub), qf[g]);
    uint4 kr[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int key = k0 + wave * 32 + it * 4 + kk;
      kr[it] = *reinterpret_cast<const uint4*>(K + (size_t)min(key, k1 - 1) * 128 + 8 * sub);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int kl = wave * 32 + it * 4 + kk;
      float kf[8];
      h_unpack8_bf16(kr[it], kf);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a += kf[e] * qf[g][e];
        a = vlm_row16_sum(a);
        if (sub == 0) sc[g][kl] = k0 + kl < k1 ? a * scale : -INFINITY;
      }
    }
  }
  __syncthreads();
  // ---- softmax pieces per head: wave w takes heads w, w + 4; 2 keys per lane ---------------------------------------------------------------
  for (int g = wave; g < G; g += 4) {
    const float s0 = sc[g][lane], s1 = sc[g][64 + lane];
    float mx = fmaxf(s0, s1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    // probabilities as bf16 (the P operand of the reference's P V product)
    const float e0 = (float)(bf16)__expf(s0 - mx), e1 = (float)(bf16)__expf(s1 - mx);
    float sum = __expf(s0 - mx) + __expf(s1 - mx);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    sc[g][lane] = e0; sc[g][64 + lane] = e1;
    if (lane == 0) { pm[((size_t)b * nh + head0 + g) * nsplit + sp] = mx; pl[((size_t)b * nh + head0 + g) * nsplit + sp] = sum; }
  }
  __syncthreads();
  // ---- O partials: the same 16-lanes-per-row pattern on V (eight 1-KiB loads per wave in flight): lane (sub, kk) accumulates dims
  // 8 sub .. 8 sub + 7 over keys 32 wave + 4 it + kk for every head; the four kk groups are then added with two shuffles -----------------------
  {
    const int sub = lane & 15, kk = lane >> 4;
    uint4 vr[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int key = k0 + wave * 32 + it * 4 + kk;
      vr[it] = *reinterpret_cast<const uint4*>(V + (size_t)min(key, k1 - 1) * 128 + 8 * sub);
    }
    float a[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[g][e] = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int kl = wave * 32 + it * 4 + kk;      // probabilities of keys past the chunk's end are zero
      float vf[8];
      h_unpack8_bf16(vr[it], vf);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float pj = sc[g][kl];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[g][e] += pj * vf[e];
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = a[g][e];
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (kk == 0) part[wave][g][8 * sub + e] = v;
      }
  }
  __syncthreads();
  for (int i = t; i < G * 128; i += 256) {
    const int g = i >> 7, dm = i & 127;
    po[(((size_t)b * nh + head0 + g) * nsplit + sp) * 128 + dm] = (part[0][g][dm] + part[1][g][dm]) + (part[2][g][dm] + part[3][g][dm]);
  }
}
__global__ void vlm_attn_combine_kernel(const float* __restrict__ po, const float* __restrict__ pm, const float* __restrict__ pl, bf16* __restrict__ o, int nsplit) {
  const int bh = blockIdx.x, t = threadIdx.x;      // 128 threads: one per output dimension
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pm[(size_t)bh * nsplit + s]);
  float l = 0.f, acc = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float mm = pm[(size_t)bh * nsplit + s];
    if (mm == -INFINITY) continue;
    const float f = __expf(mm - M);
    l += f * pl[(size_t)bh * nsplit + s];
    acc += f * po[((size_t)bh * nsplit + s) * 128 + t];
  }
  o[(size_t)bh * 128 + t] = (bf16)(acc / l);
}

// ---- model ---------------------------------------------------------------------------------------------------------------------------
static bf16* upload_bf16(DeviceWeights& dw, const std::vector<float>& v) {
  std::vector<uint16_t> h(v.size());
  for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16_bits(v[i]);
  return (bf16*)dw.upload_raw(h.data(), h.size() * sizeof(uint16_t));
}

void build_vlm(VlmModel& m, const WeightStore& ws, const VlmConfig& cfg) {
  m.cfg = cfg;
  m.dw.prec = PREC_BF16;
  m.dw.half_only = true;      // 7.6 G parameters: no fp32 / Winograd copies beside the bf16 ones
  const std::string P = "model.language_model.";
  const HostTensor& E = ws.get(P + "embed_tokens.weight");
  m.vocab = (int)E.shape[0]; m.hidden = (int)E.shape[1];
  FE_CHECK(cfg.head_dim == 128, "vlm: head_dim %d (the attention kernels are built for 128)", cfg.head_dim);
  FE_CHECK(cfg.n_heads > 0 && cfg.n_kv_heads > 0 && cfg.n_heads % cfg.n_kv_heads == 0, "vlm: %d heads / %d kv heads", cfg.n_heads, cfg.n_kv_heads);
  FE_CHECK(cfg.mrope[0] + cfg.mrope[1] + cfg.mrope[2] == 64, "vlm: mrope sections must sum to head_dim / 2");
  FE_CHECK(m.hidden % 64 == 0, "vlm: hidden size %d must be a multiple of 64", m.hidden);
  m.embed = upload_bf16(m.dw, E.data);
  m.layers.clear();
  const int qd = cfg.n_heads * 128, kd = cfg.n_kv_heads * 128;
  for (int i = 0;; ++i) {
    const std::string L = P + "layers." + std::to_string(i);
    if (!ws.has(L + ".self_attn.q_proj.weight")) break;
    VlmLayerW w;
    // q | k | v as ONE projection: rows concatenated, biases concatenated
    HostTensor W, Bv;
    W.shape = {qd + 2 * kd, m.hidden}; Bv.shape = {qd + 2 * kd};
    for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
      const HostTensor& a = ws.get(L + ".self_attn." + n + ".weight");
      const HostTensor& bb = ws.get(L + ".self_attn." + n + ".bias");
      FE_CHECK((int)a.shape[1] == m.hidden, "vlm: %s input width", n);
      W.data.insert(W.data.end(), a.data.begin(), a.data.end());
      Bv.data.insert(Bv.data.end(), bb.data.begin(), bb.data.end());
    }
    FE_CHECK((int64_t)W.data.size() == W.shape[0] * W.shape[1], "vlm: layer %d q/k/v shapes do not match %d heads / %d kv heads of 128", i, cfg.n_heads, cfg.n_kv_heads);
    w.qkv = build_linear_rows(m.dw, W, &Bv, 0, qd + 2 * kd);
    w.o = build_linear(m.dw, ws, L + ".self_attn.o_proj", false);
    w.gate = build_linear(m.dw, ws, L + ".mlp.gate_proj", false);
    w.up = build_linear(m.dw, ws, L + ".mlp.up_proj", false);
    w.down = build_linear(m.dw, ws, L + ".mlp.down_proj", false);
    w.ln1 = upload_bf16(m.dw, ws.get(L + ".input_layernorm.weight").data);
    w.ln2 = upload_bf16(m.dw, ws.get(L + ".post_attention_layernorm.weight").data);
    m.layers.push_back(w);
  }
  FE_CHECK(!m.layers.empty(), "vlm: no decoder layers found");
  m.norm = upload_bf16(m.dw, ws.get(P + "norm.weight").data);
  m.lm_head = build_linear(m.dw, ws, "lm_head", false);
  m.inter = m.layers[0].gate.Cout;
  // inv_freq as Qwen2_5_VLRotaryEmbedding.compute_default_rope_parameters: 1 / base^(2i / dim), fp32
  std::vector<float> inv(64);
  for (int i = 0; i < 64; ++i) inv[i] = 1.0f / powf(cfg.rope_theta, (float)(2 * i) / 128.0f);
  m.inv_freq = m.dw.upload(inv);
  build_vlm_vision(m, ws);      // model.visual.* when the checkpoint carries it
}

void VlmModel::reserve_cache(int B, int max_seq_) {
  const size_t per = (size_t)B * cfg.n_kv_heads * max_seq_ * 128;
  if (B == cache_B && max_seq_ == max_seq && !kcache.empty()) return;
  release_cache();
  for (size_t i = 0; i < layers.size(); ++i) {
    void *k = nullptr, *v = nullptr;
    FE_HIP(hipMalloc(&k, per * sizeof(bf16)));
    kcache.push_back((bf16*)k);
    FE_HIP(hipMalloc(&v, per * sizeof(bf16)));
    vcache.push_back((bf16*)v);
  }
  cache_B = B; max_seq = max_seq_; cur_len = 0;
}
void VlmModel::release_cache() {
  for (bf16* p : kcache) (void)hipFree(p);
  for (bf16* p : vcache) (void)hipFree(p);
  kcache.clear(); vcache.clear();
  cache_B = 0; max_seq = 0; cur_len = 0;
}

static inline int grid_n(size_t n, int per = 256) { size_t g = (n + per - 1) / per; return (int)(g > 65535 * 4 ? 65535 * 4 : (g ? g : 1)); }

void vlm_rmsnorm(Ctx& c, const bf16* x, int ldx, const bf16* w, bf16* y, int ldy, int rows, int d, float eps) {
  hipLaunchKernelGGL(vlm_rmsnorm_kernel, dim3(grid_n((size_t)rows * 64)), dim3(256), 0, c.stream, x, ldx, w, y, ldy, rows, d, eps);
  FE_HIP(hipGetLastError());
}
void vlm_add(Ctx& c, bf16* x, const bf16* y, size_t n) {
  hipLaunchKernelGGL(vlm_add_kernel, dim3(grid_n(n / 4)), dim3(256), 0, c.stream, x, y, n / 4);
  FE_HIP(hipGetLastError());
}
void vlm_silu_mul(Ctx& c, const bf16* g, const bf16* u, bf16* h, size_t n) {
  hipLaunchKernelGGL(vlm_silu_mul_kernel, dim3(grid_n(n / 4)), dim3(256), 0, c.stream, g, u, h, n / 4);
  FE_HIP(hipGetLastError());
}
void vlm_put_rows(Ctx& c, bf16* x, const bf16* rows, const int* index, int n, int d) {
  hipLaunchKernelGGL(vlm_put_rows_kernel, dim3(grid_n((size_t)n * d / 8)), dim3(256), 0, c.stream, x, rows, index, n, d);
  FE_HIP(hipGetLastError());
}

// x: [B*L][hidden] bf16 rows (token embeddings, image rows already in place), pos: device [3][B*L]. Appends L positions to the cache of
// every sequence, leaves the next token of every sequence in next_dev [B] and (optionally) the bf16-rounded logits in logits_dev [B][vocab].
// len_dev (decode steps only): the cache length in device memory, read by the kernels instead of the host's cur_len - the form a
// captured graph replays.
void vlm_forward(Ctx& c, VlmModel& m, bf16* x, const int* pos, int B, int L, int* next_dev, float* logits_dev, const int* len_dev) {
  const VlmConfig& g = m.cfg;
  const int rows = B * L, d = m.hidden, nh = g.n_heads, nkv = g.n_kv_heads, qd = nh * 128, qkvd = (nh + 2 * nkv) * 128;
  const int start = m.cur_len, Lk = start + L;
  FE_CHECK(B == m.cache_B && Lk <= m.max_seq, "vlm: %d sequences x %d positions do not fit the cache (%d x %d)", B, Lk, m.cache_B, m.max_seq);
  const size_t mark = c.arena.mark();
  bf16* n = c.arena.array<bf16>((size_t)rows * d);
  bf16* qkv = c.arena.array<bf16>((size_t)rows * qkvd);
  bf16* qr = c.arena.array<bf16>((size_t)rows * qd);
  bf16* ao = c.arena.array<bf16>((size_t)rows * qd);
  bf16* br = c.arena.array<bf16>((size_t)rows * d);
  bf16* gg = c.arena.array<bf16>((size_t)rows * m.inter);
  bf16* uu = c.arena.array<bf16>((size_t)rows * m.inter);
  const float scale = 1.0f / sqrtf(128.f);
  // decode attention: keys split over workgroups, merged by a second launch (graph replay: the grid covers the whole cache)
  const int nsplit = L == 1 ? ((len_dev ? m.max_seq : Lk) + VLM_DEC_CHUNK - 1) / VLM_DEC_CHUNK : 0;
  float* po = nsplit ? c.arena.array<float>((size_t)B * nh * nsplit * 128) : nullptr;
  float* pm = nsplit ? c.arena.array<float>((size_t)B * nh * nsplit) : nullptr;
  float* pl = nsplit ? c.arena.array<float>((size_t)B * nh * nsplit) : nullptr;
  // (few rows: one 1024-thread workgroup per row - the wave-per-row kernel walks a 3584-wide row in 14 dependent steps, 11 us for 32 rows;
  // with no split sums to take, the finishing pass is x = bf16(x + 0), n = RMSNorm(x) w)
  const bool few_rows = rows <= 64 && d % 4 == 0;
  if (few_rows)
    hipLaunchKernelGGL(vlm_finish_add_rmsnorm_kernel, dim3(rows), dim3(1024), 0, c.stream, (const float*)nullptr, 0, rows, d, (const float*)nullptr, x, (const bf16*)m.layers[0].ln1, n, g.rms_eps);
  else
    hipLaunchKernelGGL(vlm_rmsnorm_kernel, dim3(grid_n((size_t)rows * 64)), dim3(256), 0, c.stream, (const bf16*)x, d, (const bf16*)m.layers[0].ln1, n, d, rows, d, g.rms_eps);
  for (size_t li = 0; li < m.layers.size(); ++li) {
    const VlmLayerW& w = m.layers[li];
    vlm_linear(c, w.qkv, (const bf16*)n, d, rows, qkv, qkvd);
    hipLaunchKernelGGL(vlm_rope_cache_kernel, dim3(grid_n((size_t)rows * (nh + 2 * nkv) * 64)), dim3(256), 0, c.stream, (const bf16*)qkv, pos, (const float*)m.inv_freq, qr,
                       m.kcache[li], m.vcache[li], rows, L, nh, nkv, g.mrope[0], g.mrope[1], start, m.max_seq, L == 1 ? len_dev : (const int*)nullptr);
    if (L == 1) {
      const int G = nh / nkv;
#define VLM_DEC_LAUNCH(GG)                                                                                                                            \
  hipLaunchKernelGGL(vlm_attn_decode_gqa_kernel<GG>, dim3(B * nkv, nsplit), dim3(256), 0, c.stream, (const bf16*)qr, (const bf16*)m.kcache[li],       \
                     (const bf16*)m.vcache[li], po, pm, pl, nh, nkv, Lk, m.max_seq, scale, len_dev, nsplit)
      switch (G) {
        case 1: VLM_DEC_LAUNCH(1); break;
        case 2: VLM_DEC_LAUNCH(2); break;
        case 4: VLM_DEC_LAUNCH(4); break;
        case 7: VLM_DEC_LAUNCH(7); break;
        case 8: VLM_DEC_LAUNCH(8); break;
        default: FE_CHECK(false, "vlm decode attention: %d query heads per KV head (1, 2, 4, 7, 8 are built)", G);
      }
#undef VLM_DEC_LAUNCH
      hipLaunchKernelGGL(vlm_attn_combine_kernel, dim3(B * nh), dim3(128), 0, c.stream, (const float*)po, (const float*)pm, (const float*)pl, ao, nsplit);
    } else {
      VlmAttnParams ap{qr, qd, m.kcache[li], m.vcache[li], ao, qd, B, nh, nkv, L, Lk, m.max_seq, start, scale};
      hipLaunchKernelGGL(vlm_attn_prefill_kernel, dim3((L + 127) / 128, B * nh), dim3(256), 0, c.stream, ap);
    }
    FE_HIP(hipGetLastError());
    c.flops_accum += 4.0 * B * nh * (double)L * (start + (L + 1) * 0.5) * 128;
    c.flops_half += 4.0 * B * nh * (double)L * (start + (L + 1) * 0.5) * 128;
    const bf16* const next_ln = li + 1 < m.layers.size() ? (const bf16*)m.layers[li + 1].ln1 : (const bf16*)nullptr;
    // 5 .. 32 sequences: the split sums of the weight-streaming GEMM are taken by the pass that follows the projection (residual +
    // norm, SwiGLU product) instead of a finishing launch of their own
    const bool fused = d % 4 == 0 && m.inter % 4 == 0 && vlm_uses_gemm32(w.o, rows) && vlm_uses_gemm32(w.gate, rows) && vlm_uses_gemm32(w.up, rows) &&
                       vlm_uses_gemm32(w.down, rows) && !w.o.shift && !w.gate.shift && !w.up.shift && !w.down.shift && w.gate.KpH == w.up.KpH;
    if (fused) {
      const size_t pm = c.arena.mark();
      float *po_ = nullptr, *pg = nullptr, *pu = nullptr, *pd = nullptr;
      int sp = vlm_gemm32_partials(c, w.o, (const bf16*)ao, qd, rows, &po_);
      hipLaunchKernelGGL(vlm_finish_add_rmsnorm_kernel, dim3(rows), dim3(1024), 0, c.stream, (const float*)po_, sp, rows, d, (const float*)nullptr, x, (const bf16*)w.ln2, n, g.rms_eps);
      const int sg = vlm_gemm32_partials(c, w.gate, (const bf16*)n, d, rows, &pg, &w.up, &pu);      // gate and up: one launch
      const size_t mn4 = (size_t)rows * m.inter / 4;
      hipLaunchKernelGGL(vlm_finish_silu_mul_kernel, dim3(grid_n(mn4)), dim3(256), 0, c.stream, (const float*)pg, (const float*)pu, sg, mn4, gg);
      sp = vlm_gemm32_partials(c, w.down, (const bf16*)gg, m.inter, rows, &pd);
      hipLaunchKernelGGL(vlm_finish_add_rmsnorm_kernel, dim3(rows), dim3(1024), 0, c.stream, (const float*)pd, sp, rows, d, (const float*)nullptr, x, next_ln, n, g.rms_eps);
      FE_HIP(hipGetLastError());
      c.arena.rewind(pm);
      continue;
    }
    vlm_linear(c, w.o, (const bf16*)ao, qd, rows, br, d);
    hipLaunchKernelGGL(vlm_add_rmsnorm_kernel, dim3(grid_n((size_t)rows * 64)), dim3(256), 0, c.stream, x, (const bf16*)br, (const bf16*)w.ln2, n, rows, d, g.rms_eps);
    vlm_linear(c, w.gate, (const bf16*)n, d, rows, gg, m.inter);
    vlm_linear(c, w.up, (const bf16*)n, d, rows, uu, m.inter);
    hipLaunchKernelGGL(vlm_silu_mul_kernel, dim3(grid_n((size_t)rows * m.inter / 4)), dim3(256), 0, c.stream, (const bf16*)gg, (const bf16*)uu, gg, (size_t)rows * m.inter / 4);
    vlm_linear(c, w.down, (const bf16*)gg, m.inter, rows, br, d);
    // x += down(...) and, in the same launch, the next layer's input norm (none after the last layer)
    hipLaunchKernelGGL(vlm_add_rmsnorm_kernel, dim3(grid_n((size_t)rows * 64)), dim3(256), 0, c.stream, x, (const bf16*)br, next_ln, n, rows, d, g.rms_eps);
    FE_HIP(hipGetLastError());
  }
  // final norm + lm_head on the last position of every sequence
  bf16* last = c.arena.array<bf16>((size_t)B * d);
  bf16* lastn = c.arena.array<bf16>((size_t)B * d);
  hipLaunchKernelGGL(vlm_last_rows_kernel, dim3((B * d + 255) / 256), dim3(256), 0, c.stream, (const bf16*)x, last, B, L, d);
  if (B <= 64 && d % 4 == 0)
    hipLaunchKernelGGL(vlm_finish_add_rmsnorm_kernel, dim3(B), dim3(1024), 0, c.stream, (const float*)nullptr, 0, B, d, (const float*)nullptr, last, (const bf16*)m.norm, lastn, g.rms_eps);
  else
    hipLaunchKernelGGL(vlm_rmsnorm_kernel, dim3(grid_n((size_t)B * 64)), dim3(256), 0, c.stream, (const bf16*)last, d, (const bf16*)m.norm, lastn, d, B, d, g.rms_eps);
  float* lg = logits_dev ? logits_dev : c.arena.array<float>((size_t)B * m.vocab);
  if (B <= 4 && !vlm_uses_gemm32(m.lm_head, B)) vlm_gemv(c, m.lm_head, (const bf16*)lastn, d, B, lg, m.vocab);
  else if (B <= 32 && d % 128 == 0) vlm_gemm32(c, m.lm_head, (const bf16*)lastn, d, B, lg, m.vocab);
  else linear_forward_f32(c, m.lm_head, (const bf16*)lastn, d, B, lg, m.vocab, ACT_NONE);
  float* apv = c.arena.array<float>((size_t)B * VLM_AM_CHUNKS);
  int* api = c.arena.array<int>((size_t)B * VLM_AM_CHUNKS);
  hipLaunchKernelGGL(vlm_argmax_part_kernel, dim3(VLM_AM_CHUNKS, B), dim3(256), 0, c.stream, lg, m.vocab, apv, api);
  hipLaunchKernelGGL(vlm_argmax_final_kernel, dim3(B), dim3(64), 0, c.stream, (const float*)apv, (const int*)api, next_dev);
  FE_HIP(hipGetLastError());
  m.cur_len = Lk;
  c.arena.rewind(mark);
}

// n_steps greedy decode steps with NO host round trip: token ids, positions and the cache length live in device memory, one step is
// captured into a HIP graph (~12 launches per layer: at the reference's batch sizes a step is launch-bound otherwise) and replayed.
// tok_dev [B] holds the tokens to feed first (the prefill's choice), pos_dev [3][B] their positions; out_dev [n_steps][B] receives the
// tokens chosen by the steps. Leaves cur_len advanced by n_steps.
void vlm_decode_steps(Ctx& c, VlmModel& m, int* tok_dev, int* pos_dev, int B, int n_steps, int* out_dev) {
  if (n_steps <= 0) return;
  FE_CHECK(B == m.cache_B && m.cur_len > 0 && m.cur_len + n_steps <= m.max_seq, "vlm: %d more positions do not fit the cache (%d of %d used)", n_steps, m.cur_len, m.max_seq);
  const size_t mark = c.arena.mark();
  int* st = c.arena.array<int>(4);                 // [0] cache length, [1] step counter
  int* next = c.arena.array<int>((size_t)B);
  bf16* x = c.arena.array<bf16>((size_t)B * m.hidden);
  const int init[2] = {m.cur_len, 0};
  FE_HIP(hipMemcpyAsync(st, init, sizeof init, hipMemcpyHostToDevice, c.stream));
  auto one_step = [&]() {
    vlm_embed(c, m, tok_dev, B, x);
    vlm_forward(c, m, x, pos_dev, B, 1, next, nullptr, st);
    hipLaunchKernelGGL(vlm_advance_kernel, dim3(1), dim3(64 * ((B + 63) / 64)), 0, c.stream, (const int*)next, tok_dev, pos_dev, st, st + 1, out_dev, B);
    FE_HIP(hipGetLastError());
  };
  const int len0 = m.cur_len;
  one_step();                                      // outside the graph: first-use attributes (dynamic LDS sizes) are set here
  static const bool no_graph = getenv("FE_VLM_NO_GRAPH") != nullptr;      // A/B hook
  // replayed from a graph the ~60 launches of a step carry a dependency edge each: worth it while the step is launch-bound (1-2 sequences:
  // 0.68 vs 0.70 ms per 4-layer step), slower than back-to-back stream launches above that (32 sequences: 1.21 vs 1.10 ms; profiles/r03_vlm_perf.txt)
  if (n_steps > 1 && !no_graph && B <= 2) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    FE_HIP(hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal));
    m.cur_len = len0 + 1;                          // (host copy: only the cache-capacity check reads it during capture)
    try { one_step(); } catch (...) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(c.stream, &g); if (g) (void)hipGraphDestroy(g); throw; }
    FE_HIP(hipStreamEndCapture(c.stream, &graph));
    FE_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int s = 1; s < n_steps; ++s) FE_HIP(hipGraphLaunch(exec, c.stream));
    FE_HIP(hipStreamSynchronize(c.stream));
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
  } else {
    for (int s = 1; s < n_steps; ++s) { m.cur_len = len0 + s; one_step(); }
  }
  m.cur_len = len0 + n_steps;
  c.arena.rewind(mark);
}

void vlm_embed(Ctx& c, const VlmModel& m, const int* tok_dev, int rows, bf16* x) {
  hipLaunchKernelGGL(vlm_embed_kernel, dim3(grid_n((size_t)rows * m.hidden / 8)), dim3(256), 0, c.stream, tok_dev, (const bf16*)m.embed, x, rows, m.hidden, m.vocab);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

// VLM tagger, slice 2: the vision tower of Qwen2.5-VL (SURVEY 8(f)-4 / BASELINE configs[4]) - pixel patches in, merged image embeddings out.
//
// Stands behind `Qwen2_5_VLForConditionalGeneration.get_image_features` = `model.visual(pixel_values, grid_thw).pooler_output`, which
// `generate(**inputs)` runs on the processor's `pixel_values [n_patches, 3*2*14*14]` / `image_grid_thw` (reference models/vlm_tagger.py:
// 245-259, 346-360). transformers' Qwen2_5_VisionTransformerPretrainedModel [modeling_qwen2_5_vl.py]: Conv3d patch embedding (a
// [n, 1176] x [1176, hidden] product) -> rows regrouped window by window (`window_index`, units of the 2x2 merge block) -> `depth` blocks
// { x += proj(attn(rope2d(qkv(RMSNorm(x))))) ; x += down(silu(gate(n)) * up(n)), n = RMSNorm(x) } where attention runs inside 112-pixel
// windows (<= 64 patches) except in the `fullatt_block_indexes` blocks (whole image) -> patch merger (RMSNorm, 4 rows -> 1, Linear - GELU -
// Linear to the decoder width) -> rows back in raster order. bf16 with the rounding points of the bf16 torch modules (each Linear output,
// RMSNorm before the weight multiply, SiLU before the gate multiply, residual sums, GELU); the rotary embedding is applied in fp32 on the
// bf16 q / k and rounded once, as apply_rotary_pos_emb_vision does. head_dim is 80 (1280 / 16): the attention kernel below runs 5 k-steps
// per S tile and three 32-row d-tiles of O (rows 80..95 of V^T are zeros). The index arrays (positions, window order, segment bounds) are
// the host's (facet_amd/vlm_tagger.py: numpy restatements of transformers.vision_utils, pinned by the golden vectors).
// Parity: tests/test_vlm_gpu.py against tests/golden/vlm_vision_golden.npz (the reference's own class) - pinned.
#include "engine.h"
#include <algorithm>
#include <cmath>

namespace fe {

// dst rows (4 i + j) = src rows (4 index[i] + j): the window regrouping (scatter = false) and its inverse on merged rows (units of 1 row)
__global__ void vlm_vis_gather_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, const int* __restrict__ index, int groups, int unit, int d, int scatter) {
  const size_t total = (size_t)groups * unit * (d / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (d / 8)) * 8;
    const size_t row = i / (d / 8);
    const int g = (int)(row / unit), j = (int)(row % unit);
    const size_t other = (size_t)index[g] * unit + j;
    const size_t s = scatter ? row : other, t = scatter ? other : row;
    *reinterpret_cast<uint4*>(dst + t * d + c) = *reinterpret_cast<const uint4*>(src + s * d + c);
  }
}

// 2-D rotary embedding of the vision tower on the q and k thirds of a fused qkv row block. Frequencies: inv_freq[j], j < hd/4; dimension
// i of a head (pairs (i, i + hd/2)) takes, with jj = i % (hd/2): the row position and inv_freq[jj] for jj < hd/4, the column position and
// inv_freq[jj - hd/4] otherwise. fp32 arithmetic on the bf16 values, one rounding (apply_rotary_pos_emb_vision).
__global__ void vlm_vis_rope_kernel(const bf16* __restrict__ qkv, const int* __restrict__ pos, const float* __restrict__ inv_freq, bf16* __restrict__ q_out,
                                    bf16* __restrict__ k_out, int rows, int heads, int hd) {
  const int half = hd / 2, quarter = hd / 4, dim = heads * hd;
  const size_t total = (size_t)rows * 2 * heads * half;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % half), hh = (int)((i / half) % (2 * heads)), row = (int)(i / ((size_t)half * 2 * heads));
    const int which = hh / heads, head = hh % heads;                    // 0: q, 1: k
    const bf16* src = qkv + (size_t)row * 3 * dim + which * dim + head * hd;
    const float p = (float)pos[2 * row + (d < quarter ? 0 : 1)];
    const float ang = p * inv_freq[d < quarter ? d : d - quarter];
    const float c = cosf(ang), s = sinf(ang);
    const float x1 = (float)src[d], x2 = (float)src[d + half];
    bf16* dst = (which ? k_out : q_out) + (size_t)row * dim + head * hd;
    dst[d] = (bf16)(x1 * c - x2 * s);
    dst[d + half] = (bf16)(x2 * c + x1 * s);
  }
}

// y = bf16(gelu_erf(x)) elementwise (nn.GELU() of the patch merger on a bf16 tensor)
__global__ void vlm_gelu_kernel(bf16* __restrict__ x, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = ld4(x + 4 * i);
    auto f = [](float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); };
    st4(x + 4 * i, make_float4(f(a.x), f(a.y), f(a.z), f(a.w)));
  }
}

// ---- attention over packed variable-length segments, head_dim 80, non-causal ---------------------------------------------------------------
// Segment s = rows cu[s] .. cu[s+1]-1 of the packed sequence (a window, or a whole image in the full-attention blocks). One workgroup =
// NW waves x 32 queries of one (segment, head); K / V tiles of 32 keys through LDS (V transposed on the way in). Same accumulate-in-place
// scheme as the decoder's prefill kernel (model_vlm.hip).
constexpr int VV_KS = 176;      // K tile row stride in bytes (160 + 16)
constexpr int VV_VS = 72;       // V^T tile row stride in bytes (64 + 8)
struct VisAttnParams {
  const bf16* q; const bf16* k; int ldqk;       // rotated q / k: [N][heads*80]
  const bf16* v; int ldv;                       // V third of the fused projection: [N][3*heads*80] + 2*heads*80
  bf16* o; int ldo;
  const int* cu; int heads; float scale;
};
union VV8 { uint4 u; fe_v4f f; };

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void vlm_vis_attn_kernel(const VisAttnParams p) {
  __shared__ __attribute__((aligned(16))) char Ks[2][32 * VV_KS];
  __shared__ __attribute__((aligned(16))) char Vs[2][96 * VV_VS];
  const bf16* const tag = nullptr;
  constexpr int NT = NW * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int seg = blockIdx.y, head = blockIdx.z;
  const int s0 = p.cu[seg], len = p.cu[seg + 1] - s0;
  const int q0 = blockIdx.x * NW * 32;
  if (q0 >= len) return;
  for (int i = t; i < 2 * 16 * VV_VS / 4; i += NT) {      // rows 80..95 of both V^T buffers: zeros, never written again
    const int b = i / (16 * VV_VS / 4), o = i % (16 * VV_VS / 4);
    reinterpret_cast<unsigned*>(&Vs[b][80 * VV_VS])[o] = 0u;
  }
  const bf16* Qp = p.q + (size_t)s0 * p.ldqk + head * 80;
  const bf16* Kp = p.k + (size_t)s0 * p.ldqk + head * 80;
  const bf16* Vp = p.v + (size_t)s0 * p.ldv + head * 80;
  const int q = q0 + wave * 32 + r;
  const bool qok = q < len;
  const int qc = qok ? q : len - 1;
  VV8 qf[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) qf[s].u = *reinterpret_cast<const uint4*>(Qp + (size_t)qc * p.ldqk + 16 * s + 8 * h);
  constexpr int PIECES = (320 + NT - 1) / NT;      // 32 keys x 10 chunks of 16 B
  uint4 kr[PIECES], vr[PIECES];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int c = t + i * NT;
      if (c < 320) {
        int key = kt * 32 + c / 10;
        if (key > len - 1) key = len - 1;
        kr[i] = *reinterpret_cast<const uint4*>(Kp + (size_t)key * p.ldqk + (c % 10) * 8);
        vr[i] = *reinterpret_cast<const uint4*>(Vp + (size_t)key * p.ldv + (c % 10) * 8);
      }
    }
  };
  auto store_tile = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int c = t + i * NT;
      if (c < 320) {
        const int key = c / 10, d0 = (c % 10) * 8;
        *reinterpret_cast<uint4*>(&Ks[buf][key * VV_KS + d0 * 2]) = kr[i];
        const bool live = kt * 32 + key < len;
        const unsigned w[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned short v = live ? (unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu)) : (unsigned short)0;
          *reinterpret_cast<unsigned short*>(&Vs[buf][(d0 + e) * VV_VS + key * 2]) = v;
        }
      }
    }
  };
  fe_f32x16 o[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int nt = (len + 31) / 32;
  load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nt) load_tile(kt + 1);
    fe_f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    const char* kb = &Ks[buf][r * VV_KS + 16 * h];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      VV8 kf;
      kf.u = *reinterpret_cast<const uint4*>(kb + 32 * s);
      st = fe_mfma16(tag, kf.f, qf[s].f, st);
    }
    const int kbase = kt * 32 + 4 * h;
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kbase + (e & 3) + 8 * (e >> 2);
      st[e] = key >= len ? -INFINITY : st[e] * p.scale;
      tmax = fmaxf(tmax, st[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = __expf(st[e] - mn); psum += st[e]; }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    const char* vb = &Vs[buf][r * VV_VS + 8 * h];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      VV8 pf;
      pf.u = make_uint4(fe_pack2(tag, st[8 * s], st[8 * s + 1]), fe_pack2(tag, st[8 * s + 2], st[8 * s + 3]),
                        fe_pack2(tag, st[8 * s + 4], st[8 * s + 5]), fe_pack2(tag, st[8 * s + 6], st[8 * s + 7]));
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(vb + dt * 32 * VV_VS + 32 * s), a1 = *reinterpret_cast<const uint2*>(vb + dt * 32 * VV_VS + 32 * s + 16);
        VV8 v;
        v.u = make_uint4(a0.x, a0.y, a1.x, a1.y);
        o[dt] = fe_mfma16(tag, v.f, pf.f, o[dt]);
      }
    }
    if (kt + 1 < nt) store_tile(buf ^ 1, kt + 1);
    __syncthreads();
  }
  if (qok) {
    const float inv = 1.f / l;
    bf16* op = p.o + (size_t)(s0 + q) * p.ldo + head * 80;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        if (d0 < 80) st4(op + d0, make_float4(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv));
      }
  }
}

// ---- model ---------------------------------------------------------------------------------------------------------------------------------
static bf16* vis_upload_bf16(DeviceWeights& dw, const std::vector<float>& v) {
  std::vector<uint16_t> h(v.size());
  for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16_bits(v[i]);
  return (bf16*)dw.upload_raw(h.data(), h.size() * sizeof(uint16_t));
}

void build_vlm_vision(VlmModel& m, const WeightStore& ws) {
  VlmVisionW& v = m.vis;
  const std::string P = "model.visual.";
  v.present = false;
  if (!ws.has(P + "patch_embed.proj.weight")) return;
  const HostTensor& pe = ws.get(P + "patch_embed.proj.weight");      // [hidden][3][t][p][p]
  HostTensor flat;
  flat.shape = {pe.shape[0], (int64_t)(pe.numel() / (size_t)pe.shape[0])};
  flat.data = pe.data;
  v.hidden = (int)flat.shape[0]; v.patch_dim = (int)flat.shape[1];
  v.heads = m.cfg.vis_heads;
  FE_CHECK(v.patch_dim % 8 == 0 && v.hidden % v.heads == 0 && v.hidden / v.heads == 80 && v.hidden % 32 == 0,
           "vlm vision: hidden %d over %d heads (the attention kernel is built for head_dim 80), patch vector %d", v.hidden, v.heads, v.patch_dim);
  v.patch = build_linear_rows(m.dw, flat, nullptr, 0, v.hidden);
  v.blocks.clear();
  for (int i = 0;; ++i) {
    const std::string B = P + "blocks." + std::to_string(i);
    if (!ws.has(B + ".attn.qkv.weight")) break;
    VlmVisionBlockW w;
    w.qkv = build_linear(m.dw, ws, B + ".attn.qkv", true);
    w.proj = build_linear(m.dw, ws, B + ".attn.proj", true);
    w.gate = build_linear(m.dw, ws, B + ".mlp.gate_proj", true);
    w.up = build_linear(m.dw, ws, B + ".mlp.up_proj", true);
    w.down = build_linear(m.dw, ws, B + ".mlp.down_proj", true);
    w.n1 = vis_upload_bf16(m.dw, ws.get(B + ".norm1.weight").data);
    w.n2 = vis_upload_bf16(m.dw, ws.get(B + ".norm2.weight").data);
    v.blocks.push_back(w);
  }
  FE_CHECK(!v.blocks.empty(), "vlm vision: no blocks found");
  v.inter = v.blocks[0].gate.Cout;
  v.ln_q = vis_upload_bf16(m.dw, ws.get(P + "merger.ln_q.weight").data);
  v.m0 = build_linear(m.dw, ws, P + "merger.mlp.0", true);
  v.m2 = build_linear(m.dw, ws, P + "merger.mlp.2", true);
  v.out_hidden = v.m2.Cout;
  FE_CHECK(v.m0.Cin == 4 * v.hidden && v.out_hidden == m.hidden, "vlm vision: merger %d -> %d does not fit the tower (%d) / decoder (%d)", v.m0.Cin, v.out_hidden, v.hidden, m.hidden);
  // Qwen2_5_VisionRotaryEmbedding(head_dim // 2): inv_freq = 1 / 10000^(arange(0, dim, 2) / dim), dim = head_dim / 2
  std::vector<float> inv(20);
  for (int i = 0; i < 20; ++i) inv[i] = 1.0f / powf(10000.0f, (float)(2 * i) / 40.0f);
  v.inv_freq = m.dw.upload(inv);
  v.fullatt.assign(m.cfg.fullatt, m.cfg.fullatt + m.cfg.n_fullatt);
  v.present = true;
}

static inline int vgrid(size_t n, int per = 256) { size_t g = (n + per - 1) / per; return (int)(g > 262140 ? 262140 : (g ? g : 1)); }

static void vis_linear(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, bf16* y, int ldy) { linear_forward(c, w, x, ldx, M, y, ldy, ACT_NONE); }

// pv: device fp32 [N][patch_dim]; pos: device int [N][2] (row, column of every patch, ALREADY in window order); widx: device int [N/4]
// (window order -> raster group); cu_win / cu_full: device segment bounds (window order) with their host counts and longest segment;
// out: device bf16 [N/4][out_hidden], raster order.
void vlm_vision_forward(Ctx& c, VlmModel& m, const float* pv, int N, const int* pos, const int* widx, const int* cu_win, int n_win, int max_win,
                        const int* cu_full, int n_full, int max_full, bf16* out) {
  VlmVisionW& v = m.vis;
  FE_CHECK(v.present, "vlm: the checkpoint had no vision tower (model.visual.*)");
  FE_CHECK(N > 0 && N % 4 == 0, "vlm vision: %d patches (whole 2x2 merge blocks expected)", N);
  const int d = v.hidden, H = v.heads;
  const size_t mark = c.arena.mark();
  bf16* pvh = c.arena.array<bf16>((size_t)N * v.patch_dim);
  bf16* h0 = c.arena.array<bf16>((size_t)N * d);
  bf16* x = c.arena.array<bf16>((size_t)N * d);
  bf16* n = c.arena.array<bf16>((size_t)N * d);
  bf16* qkv = c.arena.array<bf16>((size_t)N * 3 * d);
  bf16* qr = c.arena.array<bf16>((size_t)N * d);
  bf16* kr = c.arena.array<bf16>((size_t)N * d);
  bf16* ao = c.arena.array<bf16>((size_t)N * d);
  bf16* br = c.arena.array<bf16>((size_t)N * d);
  // the MLP width (3420 at the 7B geometry) is not a multiple of 8: rows are padded to the 8 columns the 2-byte GEMM reads per chunk,
  // the padding zeroed once (silu(0) * 0 = 0 keeps it zero; the packed down_proj weights are zero there too)
  const int ip = (v.inter + 7) & ~7;
  bf16* gg = c.arena.array<bf16>((size_t)N * ip);
  bf16* uu = c.arena.array<bf16>((size_t)N * ip);
  if (ip != v.inter) {
    FE_HIP(hipMemsetAsync(gg, 0, (size_t)N * ip * sizeof(bf16), c.stream));
    FE_HIP(hipMemsetAsync(uu, 0, (size_t)N * ip * sizeof(bf16), c.stream));
  }
  launch_convert(pv, pvh, (size_t)N * v.patch_dim, c.stream);      // pixel_values.to(bfloat16), as the patch embedding does
  vis_linear(c, v.patch, pvh, v.patch_dim, N, h0, d);
  hipLaunchKernelGGL(vlm_vis_gather_kernel, dim3(vgrid((size_t)N * d / 8)), dim3(256), 0, c.stream, (const bf16*)h0, x, widx, N / 4, 4, d, 0);
  FE_HIP(hipGetLastError());
  const float scale = 1.0f / sqrtf(80.f);
  for (size_t li = 0; li < v.blocks.size(); ++li) {
    const VlmVisionBlockW& w = v.blocks[li];
    const bool full = std::find(v.fullatt.begin(), v.fullatt.end(), (int)li) != v.fullatt.end();
    vlm_rmsnorm(c, x, d, w.n1, n, d, N, d, 1e-6f);
    vis_linear(c, w.qkv, n, d, N, qkv, 3 * d);
    hipLaunchKernelGGL(vlm_vis_rope_kernel, dim3(vgrid((size_t)N * 2 * H * 40)), dim3(256), 0, c.stream, (const bf16*)qkv, pos, (const float*)v.inv_freq, qr, kr, N, H, 80);
    VisAttnParams ap{qr, kr, d, qkv + 2 * d, 3 * d, ao, d, full ? cu_full : cu_win, H, scale};
    const int nseg = full ? n_full : n_win, mx = full ? max_full : max_win;
    if (mx <= 64) hipLaunchKernelGGL(vlm_vis_attn_kernel<2>, dim3((mx + 63) / 64, nseg, H), dim3(128), 0, c.stream, ap);
    else hipLaunchKernelGGL(vlm_vis_attn_kernel<4>, dim3((mx + 127) / 128, nseg, H), dim3(256), 0, c.stream, ap);
    FE_HIP(hipGetLastError());
    vis_linear(c, w.proj, ao, d, N, br, d);
    vlm_add(c, x, br, (size_t)N * d);
    vlm_rmsnorm(c, x, d, w.n2, n, d, N, d, 1e-6f);
    vis_linear(c, w.gate, n, d, N, gg, ip);
    vis_linear(c, w.up, n, d, N, uu, ip);
    vlm_silu_mul(c, gg, uu, gg, (size_t)N * ip);
    vis_linear(c, w.down, gg, ip, N, br, d);
    vlm_add(c, x, br, (size_t)N * d);
  }
  // merger: RMSNorm per patch row, four consecutive rows = one merged row, Linear - GELU - Linear, then raster order
  vlm_rmsnorm(c, x, d, v.ln_q, n, d, N, d, 1e-6f);
  bf16* t0 = c.arena.array<bf16>((size_t)(N / 4) * 4 * d);
  bf16* e = c.arena.array<bf16>((size_t)(N / 4) * v.out_hidden);
  vis_linear(c, v.m0, n, 4 * d, N / 4, t0, 4 * d);
  hipLaunchKernelGGL(vlm_gelu_kernel, dim3(vgrid((size_t)N * d / 4)), dim3(256), 0, c.stream, t0, (size_t)N * d / 4);
  vis_linear(c, v.m2, t0, 4 * d, N / 4, e, v.out_hidden);
  hipLaunchKernelGGL(vlm_vis_gather_kernel, dim3(vgrid((size_t)(N / 4) * v.out_hidden / 8)), dim3(256), 0, c.stream, (const bf16*)e, out, widx, N / 4, 1, v.out_hidden, 1);
  FE_HIP(hipGetLastError());
  c.arena.rewind(mark);
}

}  // namespace fe

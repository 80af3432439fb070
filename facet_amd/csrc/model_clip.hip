// CLIP ViT-L/14 image tower (open_clip `VisionTransformer`, pretrained tag laion2b_s32b_b82k) + the LAION aesthetic MLP.
//
// Stands behind reference processing/scorer.py:661-665 / processing/multi_pass.py:522-529:
//   features = model.encode_image(inputs); F.normalize(features); aesthetic_head(features.float()).
// open_clip is not vendored in the reference (requirements.txt:8): the graph below follows its published
// VisionTransformer [DEP-KNOWLEDGE]: conv1 14x14/14 (no bias) -> [cls | patches] + positional_embedding -> ln_pre ->
// 24 x { x += MHA(ln_1(x)); x += c_proj(GELU(c_fc(ln_2(x)))) } -> ln_post(x[:,0]) @ proj. Exact-erf GELU, LN eps 1e-5.
// Parity is checked against oracle/clip_vit.py (same restatement in torch-CPU): "parity unpinned".
#include "engine.h"
#include <algorithm>
#include <cmath>

namespace fe {

// Split-operand weight of a linear layer W [N][K] (+ bias): rows [Wh | Wh | Wl] (terms = 3) or [Wh | Wl] (terms = 2) in fp16, where
// Wh = fp16(2^s W), Wl = fp16(2^s W - Wh). The power of two keeps Wl out of fp16's subnormal range (weights ~ 1/sqrt(K)); it is undone
// by the epilogue scale, which also carries `col_scale` (the 1/sqrt(head_dim) of the q rows). Consumed by gemm_split below.
static ConvW pack_split(DeviceWeights& dw, int N, int K, const float* W, const float* bias, int terms, const std::vector<float>* col_scale = nullptr) {
  FE_CHECK(K % 64 == 0 && (terms == 2 || terms == 3), "pack_split: K = %d", K);
  float amax = 0.f;
  for (size_t i = 0; i < (size_t)N * K; ++i) amax = std::max(amax, std::fabs(W[i]));
  int s = 8;
  while (s > 0 && amax * std::ldexp(1.0f, s) > 16384.f) --s;
  const float up = std::ldexp(1.0f, s);
  ConvW c;
  c.Cout = N; c.Cin = K; c.KH = c.KW = 1; c.CinPad = K; c.K = K; c.Kp = K;
  c.CinPadH = terms * K; c.KpH = terms * K; c.cb = 32; c.hprec = PREC_F16;
  std::vector<uint16_t> packed((size_t)N * terms * K);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float w = W[(size_t)n * K + k] * up;
      const uint16_t hb = f32_to_f16_bits(w);
      // the fp16 value back as float (exact): sign, exponent, mantissa
      const int e = (hb >> 10) & 31, mant = hb & 1023;
      const float hf = (hb & 0x8000 ? -1.f : 1.f) * (e == 0 ? std::ldexp((float)mant, -24) : std::ldexp((float)(mant | 1024), e - 25));
      const uint16_t lb = f32_to_f16_bits(w - hf);
      uint16_t* row = &packed[(size_t)n * terms * K];
      row[k] = hb;
      if (terms == 3) { row[K + k] = hb; row[2 * K + k] = lb; }
      else row[K + k] = lb;
    }
  c.wh = dw.upload_raw(packed.data(), packed.size() * sizeof(uint16_t));
  std::vector<float> sc(N, 1.0f / up);
  if (col_scale) for (int n = 0; n < N; ++n) sc[n] *= (*col_scale)[n];
  c.scale = dw.upload(sc);
  if (bias) { std::vector<float> bv(bias, bias + N); if (col_scale) for (int n = 0; n < N; ++n) bv[n] *= (*col_scale)[n]; c.shift = dw.upload(bv); }
  return c;
}

void build_clip(ClipModel& m, const WeightStore& ws) {
  const std::string p = "visual.";
  m.split3 = m.dw.split3 && m.dw.prec == PREC_F16;
  m.patch = build_conv(m.dw, ws, p + "conv1", "", false);
  m.width = m.patch.Cout;
  m.patch_size = m.patch.KH;
  const HostTensor& pos = ws.get(p + "positional_embedding");
  m.tokens = (int)pos.shape[0];
  FE_CHECK((int)pos.shape[1] == m.width, "clip: positional_embedding width mismatch");
  m.pos = m.dw.upload(pos.data);
  m.cls = m.dw.upload(ws.get(p + "class_embedding").data);
  m.ln_pre = build_ln(m.dw, ws, p + "ln_pre");
  m.ln_post = build_ln(m.dw, ws, p + "ln_post");
  m.heads = m.width / 64;
  m.blocks.clear();
  for (int i = 0;; ++i) {
    const std::string b = p + "transformer.resblocks." + std::to_string(i);
    if (!ws.has(b + ".ln_1.weight")) break;
    ClipBlockW w;
    w.ln1 = build_ln(m.dw, ws, b + ".ln_1");
    w.ln2 = build_ln(m.dw, ws, b + ".ln_2");
    w.attn = build_mha(m.dw, ws, b + ".attn", m.heads);
    w.fc = build_linear(m.dw, ws, b + ".mlp.c_fc", true);
    w.proj = build_linear(m.dw, ws, b + ".mlp.c_proj", true);
    if (m.split3) {
      const int d = m.width;
      const HostTensor& Wi = ws.get(b + ".attn.in_proj_weight");
      const HostTensor& Bi = ws.get(b + ".attn.in_proj_bias");
      std::vector<float> cs(3 * d, 1.0f);      // q = (x Wq^T + bq) / sqrt(head_dim): scaled after the bias, like build_mha
      for (int i = 0; i < d; ++i) cs[i] = 1.0f / std::sqrt((float)(d / m.heads));
      w.qkv3 = pack_split(m.dw, 3 * d, d, Wi.data.data(), Bi.data.data(), 3, &cs);
      w.out2 = pack_split(m.dw, d, d, ws.get(b + ".attn.out_proj.weight").data.data(), ws.get(b + ".attn.out_proj.bias").data.data(), 3);
      const HostTensor& Wf = ws.get(b + ".mlp.c_fc.weight");
      w.fc3 = pack_split(m.dw, (int)Wf.shape[0], d, Wf.data.data(), ws.get(b + ".mlp.c_fc.bias").data.data(), 3);
      const HostTensor& Wp = ws.get(b + ".mlp.c_proj.weight");
      w.proj3 = pack_split(m.dw, d, (int)Wp.shape[1], Wp.data.data(), ws.get(b + ".mlp.c_proj.bias").data.data(), 3);
    }
    m.blocks.push_back(w);
  }
  FE_CHECK(!m.blocks.empty(), "clip: no transformer blocks found");
  if (m.split3) m.zero_bias = m.dw.upload(std::vector<float>((size_t)m.width, 0.f));
  // proj is [width][out]: features = pooled @ proj  ==  Linear with weight proj^T
  const HostTensor& pr = ws.get(p + "proj");
  HostTensor prt;
  prt.shape = {pr.shape[1], pr.shape[0]};
  prt.data.resize(pr.data.size());
  for (int i = 0; i < pr.shape[0]; ++i)
    for (int j = 0; j < pr.shape[1]; ++j) prt.data[(size_t)j * pr.shape[0] + i] = pr.data[(size_t)i * pr.shape[1] + j];
  m.out_dim = (int)pr.shape[1];
  m.proj = build_linear_rows(m.dw, prt, nullptr, 0, m.out_dim);
}

void build_clip_text(ClipTextModel& m, const WeightStore& ws) {
  const HostTensor& te = ws.get("token_embedding.weight");
  m.vocab = (int)te.shape[0]; m.width = (int)te.shape[1];
  m.tok_emb = m.dw.upload(te.data);
  const HostTensor& pe = ws.get("positional_embedding");
  m.ctx = (int)pe.shape[0];
  FE_CHECK((int)pe.shape[1] == m.width && m.width % 64 == 0, "clip text: positional_embedding width mismatch");
  m.pos = m.dw.upload(pe.data);
  m.heads = m.width / 64;
  for (int i = 0;; ++i) {
    const std::string b = "transformer.resblocks." + std::to_string(i);
    if (!ws.has(b + ".ln_1.weight")) break;
    ClipBlockW w;
    w.ln1 = build_ln(m.dw, ws, b + ".ln_1");
    w.ln2 = build_ln(m.dw, ws, b + ".ln_2");
    w.attn = build_mha(m.dw, ws, b + ".attn", m.heads);
    w.fc = build_linear(m.dw, ws, b + ".mlp.c_fc", true);
    w.proj = build_linear(m.dw, ws, b + ".mlp.c_proj", true);
    m.blocks.push_back(w);
  }
  FE_CHECK(!m.blocks.empty(), "clip text: no transformer blocks found");
  m.ln_final = build_ln(m.dw, ws, "ln_final");
  const HostTensor& pr = ws.get("text_projection");   // [width][out]: features = pooled @ text_projection
  HostTensor prt;
  prt.shape = {pr.shape[1], pr.shape[0]};
  prt.data.resize(pr.data.size());
  for (int i = 0; i < pr.shape[0]; ++i)
    for (int j = 0; j < pr.shape[1]; ++j) prt.data[(size_t)j * pr.shape[0] + i] = pr.data[(size_t)i * pr.shape[1] + j];
  m.out_dim = (int)pr.shape[1];
  m.proj = build_linear_rows(m.dw, prt, nullptr, 0, m.out_dim);
}

// x[b][t][:] = tok_emb[tokens[b][t]][:] + pos[t][:]
__global__ void text_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ emb, const float* __restrict__ pos,
                                  float* __restrict__ x, int rows, int ctx, int d, int vocab) {
  const size_t total = (size_t)rows * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % d;
    const int row = i / d;
    int tk = tokens[row];
    tk = tk < 0 ? 0 : (tk >= vocab ? vocab - 1 : tk);
    x[i] = emb[(size_t)tk * d + c] + pos[(size_t)(row % ctx) * d + c];
  }
}
// pooled[b][:] = x[b][eot[b]][:]
__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ eot, float* __restrict__ y, int B,
                                   int ctx, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * d) return;
  const int b = i / d, c = i % d;
  int e = eot[b];
  e = e < 0 ? 0 : (e >= ctx ? ctx - 1 : e);
  y[i] = x[((size_t)b * ctx + e) * d + c];
}

void clip_text_forward(Ctx& c, const ClipTextModel& m, const int* tokens, const int* eot, int B, float* feat) {
  const size_t mark = c.arena.mark();
  const int d = m.width, T = m.ctx, rows = B * T;
  float* xa = (float*)c.arena.alloc((size_t)rows * d * sizeof(float));
  float* xb = (float*)c.arena.alloc((size_t)rows * d * sizeof(float));
  float* nb = (float*)c.arena.alloc((size_t)rows * d * sizeof(float));
  float* hb = (float*)c.arena.alloc((size_t)rows * 4 * d * sizeof(float));
  hipLaunchKernelGGL(text_embed_kernel, dim3(1024), dim3(256), 0, c.stream, tokens, m.tok_emb, m.pos, xa, rows, T, d, m.vocab);
  FE_HIP(hipGetLastError());
  float* cur = xa;
  float* other = xb;
  for (const ClipBlockW& w : m.blocks) {
    launch_layernorm(cur, d, nb, d, w.ln1.g, w.ln1.b, rows, d, w.ln1.eps, c.stream);
    mha_forward(c, w.attn, nb, d, nb, d, B, T, T, cur, d, other, d, /*causal=*/true);
    launch_layernorm(other, d, nb, d, w.ln2.g, w.ln2.b, rows, d, w.ln2.eps, c.stream);
    linear_forward(c, w.fc, nb, d, rows, hb, w.fc.Cout, ACT_GELU);
    linear_forward(c, w.proj, hb, w.fc.Cout, rows, cur, d, ACT_NONE, other, d);
  }
  float* pooled = (float*)c.arena.alloc((size_t)B * d * sizeof(float));
  float* pn = (float*)c.arena.alloc((size_t)B * d * sizeof(float));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((B * d + 255) / 256), dim3(256), 0, c.stream, cur, eot, pooled, B, T, d);
  FE_HIP(hipGetLastError());
  launch_layernorm(pooled, d, pn, d, m.ln_final.g, m.ln_final.b, B, d, m.ln_final.eps, c.stream);   // LN is per row: gather first
  linear_forward(c, m.proj, pn, d, B, feat, m.out_dim, ACT_NONE);
  c.arena.rewind(mark);
}

void build_aesthetic(AestheticModel& m, const WeightStore& ws) {
  m.l0 = build_linear(m.dw, ws, "0", true);
  m.l2 = build_linear(m.dw, ws, "2", true);
}

// tok[b][0] = cls + pos[0]; tok[b][1+p] = patch[b][1+p] + pos[1+p]. patch = the fp32 output of the patch-embed GEMM (rows 1..T-1
// of every image); tok has the activation type of the tower (in place when both are fp32).
template <class T>
__global__ void clip_embed_kernel(const float* patch, T* tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                  int B, int Tk, int d) {
  const size_t total = (size_t)B * Tk * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % d;
    const int tkn = (i / d) % Tk;
    const float pe = pos[(size_t)tkn * d + c];
    stf(tok + i, tkn == 0 ? cls[c] + pe : patch[i] + pe);
  }
}

// y[b][:] = x[b][:] / max(||x[b]||_2, eps)  (torch.nn.functional.normalize), one wave per row
__global__ void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) { const float v = x[(size_t)row * d + i]; s += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  for (int i = lane; i < d; i += 64) y[(size_t)row * d + i] = x[(size_t)row * d + i] * inv;
}

// x: [B,224,224,4] fp32 (CLIP-normalised, 4th channel zero) -> feat: device fp32 [B][out_dim] (un-normalised features).
// T = type of the GEMM operands (float | bf16 | f16), RT = type of the token stream (T, or float: FE_PRECISION_RES32 - the residual
// stream x and the input of every LayerNorm stay fp32, LayerNorm writes the 2-byte operand of the next GEMM, the two projections that
// add to the stream read and write it in fp32). The 3-channel patch embedding always runs on the fp32 kernel, the last projection
// always emits fp32.
template <class T, class RT>
void clip_forward(Ctx& c, const ClipModel& m, const Tensor& x, float* feat) {
  const size_t mark = c.arena.mark();
  const int B = x.n, d = m.width, Tk = m.tokens, P = m.patch_size;
  const int gh = x.h / P, gw = x.w / P;
  FE_CHECK(gh * gw + 1 == Tk, "clip: %dx%d input gives %d patches, positional_embedding has %d tokens", x.h, x.w, gh * gw, Tk);
  RT* tok = c.arena.array<RT>((size_t)B * Tk * d);
  float* patch = sizeof(RT) == 4 ? reinterpret_cast<float*>(tok) : c.arena.array<float>((size_t)B * Tk * d);
  {  // patch embed, one batch entry per image so rows land at patch[b][1 + p]
    ConvParams p{};
    p.x = x.p; p.ldx = x.ld; p.w = m.patch.w; p.y = patch + d; p.ldy = d;
    p.N = 1; p.H = x.h; p.W = x.w; p.Cin = m.patch.CinPad; p.Ho = gh; p.Wo = gw; p.Cout = d;
    p.KH = p.KW = P; p.sh = p.sw = P; p.dh = p.dw = 1;
    p.K = m.patch.K; p.Kp = m.patch.Kp; p.M = gh * gw;
    p.batch = B; p.nb1 = 1; p.xs2 = (long long)x.h * x.w * x.ld; p.ys2 = (long long)Tk * d;
    launch_conv(p, c.stream);
    c.flops_accum += 2.0 * B * p.M * (double)(P * P * m.patch.Cin) * d;
  }
  hipLaunchKernelGGL(clip_embed_kernel<RT>, dim3(2048), dim3(256), 0, c.stream, patch, tok, m.cls, m.pos, B, Tk, d);
  FE_HIP(hipGetLastError());
  const int rows = B * Tk;
  RT* xa = c.arena.array<RT>((size_t)rows * d);
  T* nb = c.arena.array<T>((size_t)rows * d);
  T* hb = c.arena.array<T>((size_t)rows * 4 * d);
  launch_layernorm(tok, d, xa, d, m.ln_pre.g, m.ln_pre.b, rows, d, m.ln_pre.eps, c.stream);
  RT* cur = xa;
  RT* other = tok;
  for (const ClipBlockW& w : m.blocks) {
    launch_layernorm(cur, d, nb, d, w.ln1.g, w.ln1.b, rows, d, w.ln1.eps, c.stream);
    mha_forward<T, RT>(c, w.attn, nb, d, nb, d, B, Tk, Tk, cur, d, other, d);          // other = cur + attn(ln1(cur))
    launch_layernorm(other, d, nb, d, w.ln2.g, w.ln2.b, rows, d, w.ln2.eps, c.stream);
    linear_forward(c, w.fc, nb, d, rows, hb, w.fc.Cout, ACT_GELU);
    linear_forward_res(c, w.proj, (const T*)hb, w.fc.Cout, rows, cur, d, ACT_NONE, (const RT*)other, d);  // cur = other + mlp(ln2(other))
  }
  // ln_post on the class token of every image (row stride Tk*d), then the projection (fp32 out)
  T* pooled = c.arena.array<T>((size_t)B * d);
  launch_layernorm(cur, Tk * d, pooled, d, m.ln_post.g, m.ln_post.b, B, d, m.ln_post.eps, c.stream);
  linear_forward_f32(c, m.proj, pooled, d, B, feat, m.out_dim, ACT_NONE);
  c.arena.rewind(mark);
}
template void clip_forward<float, float>(Ctx&, const ClipModel&, const Tensor&, float*);
template void clip_forward<bf16, bf16>(Ctx&, const ClipModel&, const Tensor&, float*);
template void clip_forward<f16, f16>(Ctx&, const ClipModel&, const Tensor&, float*);
template void clip_forward<bf16, float>(Ctx&, const ClipModel&, const Tensor&, float*);
template void clip_forward<f16, float>(Ctx&, const ClipModel&, const Tensor&, float*);

// ---- split-operand fp16 tower (FE_PRECISION_F16 | FE_PRECISION_SPLIT3) ----------------------------------------------------------------
// y = act(A . W3^T * scale + shift) (+ res32): A = fp16 rows holding a_cols columns ([xh | xl], or the plain x), W3 = pack_split's rows
// of K' = w.KpH columns; the kernel reads A's columns 0 .. a_cols-1 and then, wrapped, 0 .. K' - a_cols - 1. Output: fp16 rows (y16) or
// fp32 rows (y32, with an optional fp32 residual): the fp32-stream form of the kernel.
// pair_off > 0: y16 rows receive the result as a split pair (hi at column n, lo at column pair_off + n).
static void gemm_split(Ctx& c, const ConvW& w, const f16* a, int lda, int a_cols, int M, f16* y16, int ldy16, float* y32, int ldy32, const float* res32, int ldr32,
                       int act, int pair_off = 0) {
  ConvParamsT<f16> p{};
  p.x = a; p.ldx = lda; p.w = (const f16*)w.wh; p.ldw = w.KpH; p.scale = w.scale; p.shift = w.shift;
  p.y = y16; p.ldy = ldy16; p.y32 = y32; p.ldy32 = ldy32; p.res32 = res32; p.ldr32 = ldr32;
  p.N = 1; p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.M = M; p.Cin = w.KpH; p.Cout = w.Cout;
  p.KH = p.KW = 1; p.sh = p.sw = p.dh = p.dw = 1; p.K = w.KpH; p.Kp = w.KpH; p.cb = 32;
  p.act = act; p.a_wrap = a_cols / 32; p.exact_act = 1; p.split_lo_off = pair_off;
  FE_CHECK(a_cols % 64 == 0 && a_cols < w.KpH, "gemm_split: operand widths");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c.profile) {
    FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1));
    FE_HIP(hipEventRecord(e0, c.stream));
  }
  launch_conv_bf16(p, c.stream);
  if (c.profile) {      // per-launch record (tools/perf_clip.py): EXECUTED FLOPs of the two or three operand products
    FE_HIP(hipEventRecord(e1, c.stream));
    FE_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FE_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    char nm[160];
    snprintf(nm, sizeof nm, "f16 split gemm M=%d K'=%d (K=%d) N=%d %s", M, w.KpH, w.K, w.Cout, pair_off ? "pair out" : (y32 ? "fp32 out" : "f16 out"));
    c.timings.push_back({nm, 2.0 * M * (double)w.KpH * w.Cout, 0.0, ms});
  }
  // algorithmic = the layer's 2 M K N; executed = the two or three operand products the matrix cores really ran (flops_saved < 0)
  const double alg = 2.0 * M * (double)w.K * w.Cout, exec = 2.0 * M * (double)w.KpH * w.Cout;
  c.flops_accum += alg; c.flops_saved -= exec - alg; c.flops_half += exec;
}

// vt[b][c][t] = src[b*T + t][col0 + c] for t < T, 0 for T <= t < Lp   (a V plane of the fused projection, transposed for the attention kernel)
__global__ void clip_v_transpose_kernel(const f16* __restrict__ src, int ld, int col0, f16* __restrict__ vt, int B, int T, int Lp, int d) {
  __shared__ f16 tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r;
    tile[r][tx] = t < T ? src[((size_t)b * T + t) * ld + col0 + c0 + tx] : (f16)0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + tx;
    if (t < Lp) vt[((size_t)b * d + c0 + r) * Lp + t] = tile[tx][r];
  }
}

// Every tensor between two layers is either the fp32 token stream or an fp16 PAIR (hi | lo): LayerNorm writes pairs, the fused q|k|v
// projection and c_fc write pairs from their epilogues, attention (kernels_attn_split.hip) reads and writes pairs, out_proj and c_proj
// add into the fp32 stream. No activation is ever rounded to a single fp16.
void clip_forward_split3(Ctx& c, const ClipModel& m, const Tensor& x, float* feat) {
  FE_CHECK(m.split3, "clip: the tower was not committed under FE_PRECISION_SPLIT3");
  const size_t mark = c.arena.mark();
  const int B = x.n, d = m.width, Tk = m.tokens, P = m.patch_size, H = m.heads;
  const int gh = x.h / P, gw = x.w / P, rows = B * Tk, Lp = (Tk + 31) / 32 * 32, ff = m.blocks[0].fc3.Cout;
  FE_CHECK(gh * gw + 1 == Tk && d % 64 == 0 && d / H == 64, "clip(split3): geometry");
  float* tok = c.arena.array<float>((size_t)rows * d);
  {  // patch embed (fp32 kernel, as in every precision)
    ConvParams p{};
    p.x = x.p; p.ldx = x.ld; p.w = m.patch.w; p.y = tok + d; p.ldy = d;
    p.N = 1; p.H = x.h; p.W = x.w; p.Cin = m.patch.CinPad; p.Ho = gh; p.Wo = gw; p.Cout = d;
    p.KH = p.KW = P; p.sh = p.sw = P; p.dh = p.dw = 1;
    p.K = m.patch.K; p.Kp = m.patch.Kp; p.M = gh * gw;
    p.batch = B; p.nb1 = 1; p.xs2 = (long long)x.h * x.w * x.ld; p.ys2 = (long long)Tk * d;
    launch_conv(p, c.stream);
    c.flops_accum += 2.0 * B * p.M * (double)(P * P * m.patch.Cin) * d;
  }
  hipLaunchKernelGGL(clip_embed_kernel<float>, dim3(2048), dim3(256), 0, c.stream, (const float*)tok, tok, m.cls, m.pos, B, Tk, d);
  FE_HIP(hipGetLastError());
  float* xa = c.arena.array<float>((size_t)rows * d);
  f16* nb = c.arena.array<f16>((size_t)rows * 2 * d);          // LayerNorm output [hi(d) | lo(d)]
  f16* qkv = c.arena.array<f16>((size_t)rows * 6 * d);         // [q k v hi (3d) | q k v lo (3d)]
  f16* vth = c.arena.array<f16>((size_t)B * d * Lp);
  f16* vtl = c.arena.array<f16>((size_t)B * d * Lp);
  f16* ao = c.arena.array<f16>((size_t)rows * 2 * d);          // attention output [hi | lo]
  f16* hb = c.arena.array<f16>((size_t)rows * 2 * ff);         // GELU(c_fc) [hi(ff) | lo(ff)]
  launch_layernorm(tok, d, xa, d, m.ln_pre.g, m.ln_pre.b, rows, d, m.ln_pre.eps, c.stream);
  float* cur = xa;
  float* other = tok;
  for (const ClipBlockW& w : m.blocks) {
    launch_layernorm_split(cur, d, nb, 2 * d, w.ln1.g, w.ln1.b, rows, d, w.ln1.eps, c.stream);
    gemm_split(c, w.qkv3, nb, 2 * d, 2 * d, rows, qkv, 6 * d, nullptr, 0, nullptr, 0, ACT_NONE, 3 * d);      // q (pre-scaled) | k | v, biases included
    hipLaunchKernelGGL(clip_v_transpose_kernel, dim3(Lp / 32, d / 32, B), dim3(256), 0, c.stream, (const f16*)qkv, 6 * d, 2 * d, vth, B, Tk, Lp, d);
    hipLaunchKernelGGL(clip_v_transpose_kernel, dim3(Lp / 32, d / 32, B), dim3(256), 0, c.stream, (const f16*)qkv, 6 * d, 5 * d, vtl, B, Tk, Lp, d);
    launch_attention_split((const f16*)qkv, (const f16*)(qkv + d), 6 * d, 3 * d, (const f16*)vth, (const f16*)vtl, Lp, ao, 2 * d, d, B, H, Tk, Tk, d, c.stream);
    c.flops_accum += 4.0 * B * H * (double)Tk * Tk * 64; c.flops_saved -= 8.0 * B * H * (double)Tk * Tk * 64; c.flops_half += 12.0 * B * H * (double)Tk * Tk * 64;
    gemm_split(c, w.out2, ao, 2 * d, 2 * d, rows, nullptr, 0, other, d, cur, d, ACT_NONE);                  // other = cur + out_proj(attn)
    launch_layernorm_split(other, d, nb, 2 * d, w.ln2.g, w.ln2.b, rows, d, w.ln2.eps, c.stream);
    gemm_split(c, w.fc3, nb, 2 * d, 2 * d, rows, hb, 2 * ff, nullptr, 0, nullptr, 0, ACT_GELU, ff);          // erf GELU, pair out
    gemm_split(c, w.proj3, hb, 2 * ff, 2 * ff, rows, nullptr, 0, cur, d, other, d, ACT_NONE);               // cur = other + c_proj(h)
  }
  float* pooled = c.arena.array<float>((size_t)B * d);
  launch_layernorm(cur, Tk * d, pooled, d, m.ln_post.g, m.ln_post.b, B, d, m.ln_post.eps, c.stream);
  // the projection of the pooled token runs on the fp32 copy of its weights: ONE layer of singly-rounded fp16 weights alone costs
  // 2.5e-4 on the features (measured) - relative weight errors of a dot product do not average out
  linear_forward(c, m.proj, (const float*)pooled, d, B, feat, m.out_dim, ACT_NONE);
  c.arena.rewind(mark);
}

// raw[b] = Linear(256,1)(relu(Linear(768,256)(feat[b])))   (reference scorer.py:579-583; (x+1)*5 clamp stays on host)
void aesthetic_forward(Ctx& c, const AestheticModel& m, const float* feat, int B, float* raw) {
  const size_t mark = c.arena.mark();
  float* h = (float*)c.arena.alloc((size_t)B * m.l0.Cout * sizeof(float));
  linear_forward(c, m.l0, feat, m.l0.CinPad, B, h, m.l0.Cout, ACT_RELU);
  linear_forward(c, m.l2, h, m.l0.Cout, B, raw, 1, ACT_NONE);
  c.arena.rewind(mark);
}

void l2_normalize(Ctx& c, const float* x, float* y, int rows, int d) {
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((rows * 64 + 255) / 256), dim3(256), 0, c.stream, x, y, rows, d);
  FE_HIP(hipGetLastError());
}

}  // namespace fe

// CLIP ViT-L/14 image tower (open_clip `VisionTransformer`, pretrained tag laion2b_s32b_b82k) + the LAION aesthetic MLP.
//
// Stands behind reference processing/scorer.py:661-665 / processing/multi_pass.py:522-529:
//   features = model.encode_image(inputs); F.normalize(features); aesthetic_head(features.float()).
// open_clip is not vendored in the reference (requirements.txt:8): the graph below follows its published
// VisionTransformer [DEP-KNOWLEDGE]: conv1 14x14/14 (no bias) -> [cls | patches] + positional_embedding -> ln_pre ->
// 24 x { x += MHA(ln_1(x)); x += c_proj(GELU(c_fc(ln_2(x)))) } -> ln_post(x[:,0]) @ proj. Exact-erf GELU, LN eps 1e-5.
// Parity is checked against oracle/clip_vit.py (same restatement in torch-CPU): "parity unpinned".
#include "engine.h"
#include <cmath>

namespace fe {

void build_clip(ClipModel& m, const WeightStore& ws) {
  const std::string p = "visual.";
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
    m.blocks.push_back(w);
  }
  FE_CHECK(!m.blocks.empty(), "clip: no transformer blocks found");
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

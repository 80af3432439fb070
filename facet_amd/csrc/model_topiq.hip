// TOPIQ-NR head (pyiqa CFANet on a ResNet-50 pyramid): gated local pooling, per-level self-attention,
// coarse-to-fine cross-attention, attention pooling and the MOS MLP.
//
// Stands behind reference models/pyiqa_scorer.py:212 (`self.model(t)`, pyiqa `topiq_nr`). The arithmetic is
// pyiqa's, which is not vendored in the reference: this follows the published CFANet definition
// (use_ref=False, inter_dim=256, 4 heads, 1 layer per block, GELU, pre-norm) [DEP-KNOWLEDGE] and is
// parity-checked against oracle/topiq.py, the same restatement in torch-CPU ("parity unpinned").
#include "engine.h"
#include <algorithm>
#include <cmath>

namespace fe {

static EncLayerW build_enc(DeviceWeights& dw, const WeightStore& ws, const std::string& p, int heads) {
  EncLayerW e;
  e.attn = build_mha(dw, ws, p + ".self_attn", heads);
  e.lin1 = build_linear(dw, ws, p + ".linear1", true);
  e.lin2 = build_linear(dw, ws, p + ".linear2", true);
  e.n1 = build_ln(dw, ws, p + ".norm1");
  e.n2 = build_ln(dw, ws, p + ".norm2");
  return e;
}

static DecLayerW build_dec(DeviceWeights& dw, const WeightStore& ws, const std::string& p, int heads) {
  DecLayerW d;
  d.cross = build_mha(dw, ws, p + ".multihead_attn", heads);
  d.lin1 = build_linear(dw, ws, p + ".linear1", true);
  d.lin2 = build_linear(dw, ws, p + ".linear2", true);
  d.n1 = build_ln(dw, ws, p + ".norm1");
  d.n2 = build_ln(dw, ws, p + ".norm2");
  d.n3 = build_ln(dw, ws, p + ".norm3");
  return d;
}

void build_topiq_head(TopiqModel& m, const WeightStore& ws) {
  const int heads = 4;
  for (int i = 0; i < 5; ++i) {
    const std::string g = "weight_pool." + std::to_string(i);
    // splitconv: Conv2d(dim, 2*dim, 1); chunk(2, dim=1) -> x1 = rows [0,dim), x2 = rows [dim, 2dim)
    const HostTensor& W = ws.get(g + ".splitconv.weight");
    const HostTensor& B = ws.get(g + ".splitconv.bias");
    const int dim = (int)W.shape[1];
    FE_CHECK(W.shape[0] == 2 * dim, "splitconv %d shape", i);
    HostTensor W2; W2.shape = {2 * dim, dim}; W2.data = W.data;  // [2dim][dim][1][1] == [2dim][dim]
    m.gate[i].split_x1 = build_linear_rows(m.dw, W2, &B, 0, dim);
    {
      // x2 = splitconv(x)[dim:2dim] feeds weight_blk[0] (a 1x1 conv) with no nonlinearity in between, so the two
      // linear maps are composed once here (in double): w0(x2) = (W0 . Wx2) x + (W0 . bx2 + b0). Saves one
      // dim x dim 1x1 convolution and a full-resolution tensor round trip per level.
      const HostTensor& W0 = ws.get(g + ".weight_blk.0.weight");   // [64][dim][1][1]
      const HostTensor& B0 = ws.get(g + ".weight_blk.0.bias");
      const int mid = (int)W0.shape[0];
      FE_CHECK((int)W0.shape[1] == dim, "weight_blk.0 of level %d expects %d inputs", i, dim);
      HostTensor We; We.shape = {mid, dim}; We.data.resize((size_t)mid * dim);
      HostTensor Be; Be.shape = {mid}; Be.data.resize(mid);
      std::vector<double> row(dim);
      for (int o = 0; o < mid; ++o) {
        std::fill(row.begin(), row.end(), 0.0);
        double bacc = B0.data[o];
        for (int k = 0; k < dim; ++k) {
          const double w0k = W0.data[(size_t)o * dim + k];
          const float* wx = &W.data[(size_t)(dim + k) * dim];   // row (dim + k) of splitconv = x2 channel k
          for (int c = 0; c < dim; ++c) row[c] += w0k * wx[c];
          bacc += w0k * B.data[dim + k];
        }
        for (int c = 0; c < dim; ++c) We.data[(size_t)o * dim + c] = (float)row[c];
        Be.data[o] = (float)bacc;
      }
      m.gate[i].w0 = build_linear_rows(m.dw, We, &Be, 0, mid);
      // the 64-channel level of a 2-byte model also gets the one-launch form of the whole gate + pool (kernels_gate.hip)
      const HostTensor& W2g = ws.get(g + ".weight_blk.2.weight");
      const HostTensor& W4g = ws.get(g + ".weight_blk.4.weight");
      if (m.dw.prec != PREC_F32 && dim == 64 && mid == 64 && W2g.shape == std::vector<int64_t>{64, 64, 3, 3} && W4g.shape == std::vector<int64_t>{1, 64, 3, 3})
        build_gate64_fragments(m.dw, m.gate[i], We.data.data(), Be.data.data(), W2g.data.data(), ws.get(g + ".weight_blk.2.bias").data.data(), W4g.data.data(),
                               ws.get(g + ".weight_blk.4.bias").data[0], W.data.data(), B.data.data());
    }
    m.gate[i].w2 = build_conv(m.dw, ws, g + ".weight_blk.2", "", true);
    {      // 2-byte models: the 64 -> 64 3x3 also in the halo-tiled kernel's fragment order (kernels_c64.hip)
      const HostTensor& W2h = ws.get(g + ".weight_blk.2.weight");
      if (m.dw.prec != PREC_F32 && W2h.shape == std::vector<int64_t>{64, 64, 3, 3}) build_c64_fragments(m.dw, W2h.data.data(), nullptr, &m.gate[i].w2_frag, nullptr);
    }
    m.gate[i].w4 = build_conv(m.dw, ws, g + ".weight_blk.4", "", true);
    m.dim_reduce[i] = build_conv(m.dw, ws, "dim_reduce." + std::to_string(i) + ".0", "", true);
    m.sa[i] = build_enc(m.dw, ws, "sa_attn_blks." + std::to_string(i) + ".layers.0", heads);
  }
  for (int i = 0; i < 4; ++i) m.cross[i] = build_dec(m.dw, ws, "attn_blks." + std::to_string(i) + ".layers.0", heads);
  m.pool = build_enc(m.dw, ws, "attn_pool", heads);
  m.s_ln0 = build_ln(m.dw, ws, "score_linear.0");
  m.s_l1 = build_linear(m.dw, ws, "score_linear.1", true);
  m.s_ln3 = build_ln(m.dw, ws, "score_linear.3");
  m.s_l4 = build_linear(m.dw, ws, "score_linear.4", true);
  m.s_l6 = build_linear(m.dw, ws, "score_linear.6", true);
  m.h_emb = ws.get("h_emb").data;  // [1][128][32][1]
  m.w_emb = ws.get("w_emb").data;  // [1][128][1][32]
  FE_CHECK(m.h_emb.size() == 128 * 32 && m.w_emb.size() == 128 * 32, "pos emb size");
  m.has_head = true;
}

// torch upsample_bicubic2d (A=-0.75, align_corners=False): taps i-1..i+2 around floor(src), clamped.
static void cubic_taps(int out, int in, int o, int idx[4], float wt[4]) {
  const float scale = (float)in / (float)out;
  const float real = scale * ((float)o + 0.5f) - 0.5f;
  const float fl = std::floor(real);
  const float t = real - fl;
  const int i0 = (int)fl;
  const float A = -0.75f;
  auto c1 = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
  auto c2 = [&](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
  wt[0] = c2(t + 1.f); wt[1] = c1(t); wt[2] = c1(1.f - t); wt[3] = c2(2.f - t);
  for (int k = 0; k < 4; ++k) {
    int j = i0 - 1 + k;
    idx[k] = j < 0 ? 0 : (j > in - 1 ? in - 1 : j);
  }
}

// pos[(y*tw + x)][c]: channels 0..127 from h_emb (varies with y), 128..255 from w_emb (varies with x).
static float* topiq_pos(TopiqModel& m, int th, int tw) {
  auto key = std::make_pair(th, tw);
  auto it = m.pos.find(key);
  if (it != m.pos.end()) return it->second;
  std::vector<float> hy((size_t)th * 128), wx((size_t)tw * 128);
  int idx[4]; float wt[4];
  for (int y = 0; y < th; ++y) {
    cubic_taps(th, 32, y, idx, wt);
    for (int c = 0; c < 128; ++c) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += wt[k] * m.h_emb[(size_t)c * 32 + idx[k]];
      hy[(size_t)y * 128 + c] = acc;
    }
  }
  for (int x = 0; x < tw; ++x) {
    cubic_taps(tw, 32, x, idx, wt);
    for (int c = 0; c < 128; ++c) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += wt[k] * m.w_emb[(size_t)c * 32 + idx[k]];
      wx[(size_t)x * 128 + c] = acc;
    }
  }
  std::vector<float> pos((size_t)th * tw * 256);
  for (int y = 0; y < th; ++y)
    for (int x = 0; x < tw; ++x) {
      float* d = &pos[((size_t)y * tw + x) * 256];
      for (int c = 0; c < 128; ++c) { d[c] = hy[(size_t)y * 128 + c]; d[128 + c] = wx[(size_t)x * 128 + c]; }
    }
  float* dev = m.dw.upload(pos);
  m.pos[key] = dev;
  return dev;
}

template <class T>
static void ln(Ctx& c, const LayerNormW& l, const T* x, T* y, int rows) {
  launch_layernorm(x, l.d, y, l.d, l.g, l.b, rows, l.d, l.eps, c.stream);
}

// x, y: [B*L][d]; y may alias x. Pre-norm encoder layer (q = k = v = norm1(x)).
template <class T>
static void enc_forward(Ctx& c, const EncLayerW& e, T* x, T* y, int B, int L) {
  const int d = e.n1.d, rows = B * L, ff = e.lin1.Cout;
  const size_t mark = c.arena.mark();
  T* n = c.arena.array<T>((size_t)rows * d);
  T* a = c.arena.array<T>((size_t)rows * d);
  T* hbuf = c.arena.array<T>((size_t)rows * ff);
  ln(c, e.n1, x, n, rows);
  mha_forward<T>(c, e.attn, n, d, n, d, B, L, L, x, d, a, d);
  ln(c, e.n2, a, n, rows);
  linear_forward(c, e.lin1, n, d, rows, hbuf, ff, ACT_GELU);
  linear_forward(c, e.lin2, hbuf, ff, rows, y, d, ACT_NONE, a, d);
  c.arena.rewind(mark);
}

// tgt: [B*Lq][d] (updated in place), memory: [B*Lk][d]
template <class T>
static void dec_forward(Ctx& c, const DecLayerW& w, T* tgt, const T* memory, int B, int Lq, int Lk) {
  const int d = w.n1.d, ff = w.lin1.Cout;
  const size_t mark = c.arena.mark();
  T* mem = c.arena.array<T>((size_t)B * Lk * d);
  T* t2 = c.arena.array<T>((size_t)B * Lq * d);
  T* a = c.arena.array<T>((size_t)B * Lq * d);
  T* hbuf = c.arena.array<T>((size_t)B * Lq * ff);
  ln(c, w.n2, memory, mem, B * Lk);
  ln(c, w.n1, (const T*)tgt, t2, B * Lq);
  mha_forward<T>(c, w.cross, t2, d, mem, d, B, Lq, Lk, tgt, d, a, d);
  ln(c, w.n3, (const T*)a, t2, B * Lq);
  linear_forward(c, w.lin1, t2, d, B * Lq, hbuf, ff, ACT_GELU);
  linear_forward(c, w.lin2, hbuf, ff, B * Lq, tgt, d, ACT_NONE, a, d);
  c.arena.rewind(mark);
}

// T = activation type (float | bf16). With bf16 the head stays bf16 up to the token mean; the MOS MLP (LayerNorm - Linear - GELU -
// LayerNorm - Linear - GELU - Linear on one 256-vector per image) always runs in fp32 on the fp32 copies of its weights.
template <class T>
void topiq_head_forward(Ctx& c, TopiqModel& m, const std::vector<TensorT<T>>& feats, float* scores_dev) {
  FE_CHECK(m.has_head && feats.size() == 5, "topiq head: not built / bad pyramid");
  const int B = feats[0].n;
  const int th = feats[4].h, tw = feats[4].w, L = th * tw, d = 256;
  const float* pos = topiq_pos(m, th, tw);
  T* tok[5];
  for (int i = 0; i < 5; ++i) tok[i] = c.arena.array<T>((size_t)B * L * d);
  for (int i = 4; i >= 0; --i) {
    const size_t mark = c.arena.mark();
    const TensorT<T>& f = feats[i];
    const GatedConvW& g = m.gate[i];
    if constexpr (sizeof(T) == 2) {
      const bool no_fused = getenv("FE_NO_FUSED_GATE") != nullptr;      // A/B hook, read per call (tests switch it inside one process)
      if (g.fused && !no_fused && f.c == 64 && f.h == 16 * th && f.w == 16 * tw && f.ld % 8 == 0) {
        // the 64-channel level: gate + 16 x 16 pool in one launch; FLOPs counted as the four convolutions it stands for
        TensorT<T> pooled = c.arena.tensor_t<T>(B, th, tw, 64);
        const double px = (double)f.pixels(), flops = 2.0 * px * (64.0 * 64 + 576.0 * 64 + 576.0 + 64.0 * 64);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c.profile) { FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1)); FE_HIP(hipEventRecord(e0, c.stream)); }
        launch_topiq_gate64(f, pooled, g.fused, g.fused_bias, m.wblk_act, m.gate_act, c.stream);
        if (c.profile) {
          FE_HIP(hipEventRecord(e1, c.stream)); FE_HIP(hipEventSynchronize(e1));
          float ms = 0.f;
          FE_HIP(hipEventElapsedTime(&ms, e0, e1));
          (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
          char nm[128];
          snprintf(nm, sizeof nm, "%s fused gate64 + pool16 M=%d", PrecOf<T>::value == PREC_F16 ? "f16" : "bf16", (int)f.pixels());
          c.timings.push_back({nm, flops, 2.0 * px * 64, ms});
        }
        c.flops_accum += flops; c.flops_half += flops;
        TensorT<T> t = mat_view(tok[i], B * L, d, d);
        t.n = B; t.h = th; t.w = tw;
        ConvOptsT<T> od; od.act = ACT_GELU;
        conv_forward(c, m.dim_reduce[i], pooled, t, od);
        launch_add_rows_bcast(tok[i], d, pos, B * L, L, d, c.stream);
        enc_forward(c, m.sa[i], tok[i], tok[i], B, L);
        c.arena.rewind(mark);
        continue;
      }
    }
    ConvOptsT<T> o0; o0.act = m.wblk_act;
    TensorT<T> wa = conv_new(c, g.w0, f, o0);    // = act(weight_blk[0](x2)) with x2 folded in (see build_topiq_head)
    ConvOptsT<T> o2; o2.act = m.wblk_act; o2.ph = o2.pw = 1;
    TensorT<T> wb;
    bool tiled = false;
    if constexpr (sizeof(T) == 2) {
      // (a launch loads the 72 KB of weights into every workgroup's LDS: only where the map is large enough to amortise it)
      if (g.w2_frag && wa.c == 64 && wa.ld == 64 && !g.w2.scale && wa.pixels() >= 131072 && !getenv("FE_NO_FUSED_C64")) {
        wb = c.arena.tensor_t<T>(wa.n, wa.h, wa.w, 64);
        const double flops = 2.0 * (double)wa.pixels() * 576.0 * 64;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c.profile) { FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1)); FE_HIP(hipEventRecord(e0, c.stream)); }
        launch_conv3x3_c64(wa, wb, (const TensorT<T>*)nullptr, g.w2_frag, nullptr, nullptr, g.w2.shift, nullptr, nullptr, m.wblk_act, c.stream);
        if (c.profile) {
          FE_HIP(hipEventRecord(e1, c.stream)); FE_HIP(hipEventSynchronize(e1));
          float ms = 0.f;
          FE_HIP(hipEventElapsedTime(&ms, e0, e1));
          (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
          char nm[128];
          snprintf(nm, sizeof nm, "%s halo-tiled conv3x3 64->64 M=%d", PrecOf<T>::value == PREC_F16 ? "f16" : "bf16", (int)wa.pixels());
          c.timings.push_back({nm, flops, 2.0 * (double)wa.pixels() * 128, ms});
        }
        c.flops_accum += flops; c.flops_half += flops;
        tiled = true;
      }
    }
    if (!tiled) wb = conv_new(c, g.w2, wa, o2);
    ConvOptsT<T> o4; o4.act = ACT_SIGMOID; o4.ph = o4.pw = 1;
    TensorT<T> wc = conv_new(c, g.w4, wb, o4);
    ConvOptsT<T> og; og.act = m.gate_act; og.gate = &wc;   // act(x1) * weight
    TensorT<T> gated = conv_new(c, g.split_x1, f, og);
    if (gated.h > th && gated.w > tw) {
      TensorT<T> pooled = c.arena.tensor_t<T>(B, th, tw, gated.c);
      launch_adaptive_avgpool(gated, pooled, c.stream);
      gated = pooled;
    }
    FE_CHECK(gated.h == th && gated.w == tw, "topiq head: level %d is %dx%d, expected %dx%d", i, gated.h, gated.w, th, tw);
    TensorT<T> t = mat_view(tok[i], B * L, d, d);
    t.n = B; t.h = th; t.w = tw;
    ConvOptsT<T> od; od.act = ACT_GELU;
    conv_forward(c, m.dim_reduce[i], gated, t, od);
    launch_add_rows_bcast(tok[i], d, pos, B * L, L, d, c.stream);
    enc_forward(c, m.sa[i], tok[i], tok[i], B, L);
    c.arena.rewind(mark);
  }
  T* query = tok[4];
  for (int i = 0; i < 4; ++i) dec_forward(c, m.cross[i], query, (const T*)tok[3 - i], B, L, L);
  enc_forward(c, m.pool, query, query, B, L);
  // mean over tokens, then the MOS MLP (fp32)
  TensorT<T> fin; fin.p = query; fin.n = B; fin.h = 1; fin.w = L; fin.c = d; fin.ld = d;
  TensorT<T> mean = c.arena.tensor_t<T>(B, 1, 1, d);
  launch_adaptive_avgpool(fin, mean, c.stream);
  float* a = c.arena.array<float>((size_t)B * d);
  float* b = c.arena.array<float>((size_t)B * d);
  const float* mean_f = to_f32(c, mean.p, (size_t)B * d);
  launch_layernorm(mean_f, d, a, d, m.s_ln0.g, m.s_ln0.b, B, d, m.s_ln0.eps, c.stream);
  linear_forward(c, m.s_l1, (const float*)a, d, B, b, d, ACT_GELU);
  launch_layernorm((const float*)b, d, a, d, m.s_ln3.g, m.s_ln3.b, B, d, m.s_ln3.eps, c.stream);
  linear_forward(c, m.s_l4, (const float*)a, d, B, b, d, ACT_GELU);
  linear_forward(c, m.s_l6, (const float*)b, d, B, scores_dev, 1, ACT_NONE);
}
template void topiq_head_forward<float>(Ctx&, TopiqModel&, const std::vector<Tensor>&, float*);
template void topiq_head_forward<bf16>(Ctx&, TopiqModel&, const std::vector<TensorH>&, float*);
template void topiq_head_forward<f16>(Ctx&, TopiqModel&, const std::vector<TensorF16>&, float*);

}  // namespace fe

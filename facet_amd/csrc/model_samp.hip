// U2-Net-P saliency + SAMP-Net composition (reference models/samp_net.py, in-tree definitions).
//   REBNCONV :45-54, RSU7/6/5/4/4F :62-255, U2NETP.forward :298-342, SAMPPModule :429-645, SAMPNet.forward :760-791.
// Every torch.cat of the reference is a channel slice of one NHWC buffer here (producers write straight into
// their slice, consumers read it with the wider pixel stride); BN+bias+ReLU and the RSU residual ride in the
// conv epilogue. Parity: PINNED against the reference's own classes through tests/golden/samp_golden.npz.
#include "engine.h"
#include <cmath>

namespace fe {

// ---------------------------------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------------------------------
static RSUW build_rsu(DeviceWeights& dw, const WeightStore& ws, const std::string& p, int depth, bool dilated) {
  RSUW r;
  r.depth = depth; r.dilated = dilated;
  auto cbr = [&](const std::string& n) { return build_conv(dw, ws, p + "." + n + ".conv_s1", p + "." + n + ".bn_s1", true); };
  r.in = cbr("rebnconvin");
  for (int k = 1; k <= depth; ++k) r.enc.push_back(cbr("rebnconv" + std::to_string(k)));
  for (int k = depth - 1; k >= 1; --k) r.dec.push_back(cbr("rebnconv" + std::to_string(k) + "d"));  // dec[0] = (L-1)d ... dec[L-2] = 1d
  return r;
}

void build_u2netp(U2NetPModel& m, const WeightStore& ws) {
  static const struct { const char* name; int depth; bool dil; } S[11] = {
      {"stage1", 7, false}, {"stage2", 6, false}, {"stage3", 5, false}, {"stage4", 4, false}, {"stage5", 4, true},
      {"stage6", 4, true}, {"stage5d", 4, true}, {"stage4d", 4, false}, {"stage3d", 5, false}, {"stage2d", 6, false},
      {"stage1d", 7, false}};
  for (int i = 0; i < 11; ++i) m.stage[i] = build_rsu(m.dw, ws, S[i].name, S[i].depth, S[i].dil);
  for (int k = 0; k < 6; ++k) m.side[k] = build_conv(m.dw, ws, "side" + std::to_string(k + 1), "", true);
  m.outconv = build_conv(m.dw, ws, "outconv", "", true);
}

static inline int ceil_half(int v) { return (v + 1) / 2; }  // MaxPool2d(2, stride=2, ceil_mode=True)

static void cbr(Ctx& c, const ConvW& w, const Tensor& x, const Tensor& y, int dil, const Tensor* res_after = nullptr) {
  ConvOpts o; o.ph = o.pw = dil; o.dh = o.dw = dil; o.act = ACT_RELU;
  if (res_after) { o.res = res_after; o.res_after_act = 1; }
  conv_forward(c, w, x, y, o);
}

// One RSU block. x: input view, out: output view (may be a slice of an outer concat buffer).
static void rsu_forward(Ctx& c, const RSUW& r, const Tensor& x, const Tensor& out) {
  const size_t mark = c.arena.mark();
  const int B = x.n, L = r.depth, mid = r.enc[0].Cout, oc = r.in.Cout;
  Tensor hin = c.arena.tensor(B, x.h, x.w, oc);
  cbr(c, r.in, x, hin, 1);
  if (r.dilated) {
    // RSU4F: all at one resolution, dilations 1,2,4,8 then 4,2,1 (samp_net.py:232-255)
    Tensor cat[3];  // cat[k] = [decoder-side | h_{k+1}] for k = 0..2
    for (int k = 0; k < 3; ++k) cat[k] = c.arena.tensor(B, x.h, x.w, 2 * mid);
    const int dil[4] = {1, 2, 4, 8};
    Tensor prev = hin;
    for (int k = 0; k < 3; ++k) {
      Tensor hk = cat[k].slice(mid, mid);
      cbr(c, r.enc[k], prev, hk, dil[k]);
      prev = hk;
    }
    cbr(c, r.enc[3], prev, cat[2].slice(0, mid), 8);       // h4 -> first half of cat(h4, h3)
    cbr(c, r.dec[0], cat[2], cat[1].slice(0, mid), 4);     // h3d
    cbr(c, r.dec[1], cat[1], cat[0].slice(0, mid), 2);     // h2d
    cbr(c, r.dec[2], cat[0], out, 1, &hin);                // h1d + hin
    c.arena.rewind(mark);
    return;
  }
  // pooled RSU-L: level k (1-based) has resolution ceil-halved k-1 times; cat[k] = [up/deeper | enc_k]
  std::vector<Tensor> cat(L);  // index 1..L-1 used
  int h = x.h, w = x.w;
  for (int k = 1; k <= L - 1; ++k) {
    cat[k] = c.arena.tensor(B, h, w, 2 * mid);
    if (k < L - 1) { h = ceil_half(h); w = ceil_half(w); }
  }
  cbr(c, r.enc[0], hin, cat[1].slice(mid, mid), 1);
  for (int k = 2; k <= L - 1; ++k) {
    Tensor pooled = c.arena.tensor(B, cat[k].h, cat[k].w, mid);
    launch_maxpool(cat[k - 1].slice(mid, mid), pooled, 2, 2, 0, c.stream);
    cbr(c, r.enc[k - 1], pooled, cat[k].slice(mid, mid), 1);
  }
  cbr(c, r.enc[L - 1], cat[L - 1].slice(mid, mid), cat[L - 1].slice(0, mid), 2);  // dilated bottom
  // decoder: dec[0] = rebnconv(L-1)d on cat[L-1]; its output is upsampled into cat[L-2][:mid], ...
  for (int k = L - 1; k >= 2; --k) {
    Tensor d = c.arena.tensor(B, cat[k].h, cat[k].w, mid);
    cbr(c, r.dec[L - 1 - k], cat[k], d, 1);
    launch_bilinear(d, cat[k - 1].slice(0, mid), c.stream);
  }
  cbr(c, r.dec[L - 2], cat[1], out, 1, &hin);
  c.arena.rewind(mark);
}

// x: [B,H,W,4] (normalised image, 4th channel zero) -> sal: [B,H,W,1] (sigmoid of the fused map d0)
void u2netp_forward(Ctx& c, const U2NetPModel& m, const Tensor& x, const Tensor& sal) {
  const size_t mark = c.arena.mark();
  const int B = x.n;
  int hs[6], wsz[6];
  hs[0] = x.h; wsz[0] = x.w;
  for (int i = 1; i < 6; ++i) { hs[i] = ceil_half(hs[i - 1]); wsz[i] = ceil_half(wsz[i - 1]); }
  // decoder concat buffers: catd[i] = [up(deeper decoder out) | encoder out hx_{i+1}] at level i (0..4)
  Tensor catd[5];
  for (int i = 0; i < 5; ++i) catd[i] = c.arena.tensor(B, hs[i], wsz[i], 128);
  Tensor hx6 = c.arena.tensor(B, hs[5], wsz[5], 64);
  Tensor in = x;
  for (int i = 0; i < 6; ++i) {
    Tensor o = i < 5 ? catd[i].slice(64, 64) : hx6;
    rsu_forward(c, m.stage[i], in, o);
    if (i < 5) {
      Tensor p = c.arena.tensor(B, hs[i + 1], wsz[i + 1], 64);
      launch_maxpool(o, p, 2, 2, 0, c.stream);
      in = p;
    }
  }
  // side maps into one 8-channel buffer (6 used; outconv's packed weight has zeros for channels 6,7)
  Tensor sides = c.arena.tensor(B, x.h, x.w, 8);
  FE_HIP(hipMemsetAsync(sides.p, 0, sides.numel() * sizeof(float), c.stream));
  auto side = [&](int k, const Tensor& feat) {
    ConvOpts o; o.ph = o.pw = 1;
    if (feat.h == x.h && feat.w == x.w) {
      conv_forward(c, m.side[k], feat, sides.slice(k, 1), o);
    } else {
      Tensor s = c.arena.tensor(B, feat.h, feat.w, 1);
      conv_forward(c, m.side[k], feat, s, o);
      launch_bilinear(s, sides.slice(k, 1), c.stream);
    }
  };
  side(5, hx6);
  Tensor deeper = hx6;
  Tensor dec[5];
  for (int i = 4; i >= 0; --i) {  // stage5d, 4d, 3d, 2d, 1d = stage[6 + (4 - i)]
    launch_bilinear(deeper, catd[i].slice(0, 64), c.stream);
    dec[i] = c.arena.tensor(B, hs[i], wsz[i], 64);
    rsu_forward(c, m.stage[6 + (4 - i)], catd[i], dec[i]);
    side(i, dec[i]);
    deeper = dec[i];
  }
  ConvOpts oo; oo.act = ACT_SIGMOID;
  conv_forward(c, m.outconv, sides, sal, oo);
  c.arena.rewind(mark);
}

// ---------------------------------------------------------------------------------------------------
// SAMP-Net
// ---------------------------------------------------------------------------------------------------
static const int kPatLen[8] = {2592, 2592, 2746, 2746, 2592, 5184, 5296, 7524};  // samp_net.py:486

void build_sampnet(SampModel& m, const WeightStore& ws) {
  const int blocks[4] = {2, 2, 2, 2};
  build_resnet(m.backbone, m.dw, ws, "backbone.", false, blocks, true);
  m.pattern_weight = build_linear(m.dw, ws, "pattern_weight_layer.3", false);
  for (int i = 0; i < 8; ++i) {
    const HostTensor& W = ws.get("pattern_module.conv_list." + std::to_string(i) + ".0.weight");
    // Conv2d(c, 1024, (kh,kw)) over a [c,kh,kw] input == Linear over the flattened (c,kh,kw) vector
    HostTensor flat; flat.data = W.data;
    flat.shape = {W.shape[0], W.shape[1] * W.shape[2] * W.shape[3]};
    FE_CHECK(flat.shape[1] == kPatLen[i], "pattern conv %d has %lld inputs, expected %d", i, (long long)flat.shape[1], kPatLen[i]);
    m.pattern[i] = build_linear_rows(m.dw, flat, nullptr, 0, (int)flat.shape[0]);
  }
  m.att_feat = build_linear(m.dw, ws, "att_feature_layer.0", false);
  m.att_pred = build_linear(m.dw, ws, "att_pred_layer.0", false);
  m.com0 = build_linear(m.dw, ws, "com_pred_layer.0", false);
  m.com3 = build_linear(m.dw, ws, "com_pred_layer.3", false);
  m.com5 = build_linear(m.dw, ws, "com_pred_layer.5", false);
}

// Region pooling of SAMPPModule._get_regional_features (samp_net.py:463-596) for a 7x7x512 map.
// One block per image; thread t owns channels t and t+256. Writes the 8 padded/truncated vectors and relu(gavg).
struct PatOut { float* f[8]; int ld[8]; float* relu_gavg; };

__device__ inline void region(const float* fm, int c, int r0, int r1, int c0, int c1, float& mx, float& av) {
  float m = -INFINITY, s = 0.f;
  for (int y = r0; y < r1; ++y)
    for (int x = c0; x < c1; ++x) { const float v = fm[(y * 7 + x) * 512 + c]; m = fmaxf(m, v); s += v; }
  mx = m; av = s / (float)((r1 - r0) * (c1 - c0));
}

__global__ void samp_patterns_kernel(const float* __restrict__ fmap, const float* __restrict__ sal7, PatOut o) {
  const int b = blockIdx.x, t = threadIdx.x;
  const float* fm = fmap + (size_t)b * 49 * 512;
  const float* sal = sal7 + (size_t)b * 49;
  __shared__ float sal_small[16];
  if (t < 16) {  // adaptive_avg_pool2d(7x7 -> 4x4)
    const int i = t / 4, j = t % 4;
    const int hs = (i * 7) / 4, he = ((i + 1) * 7 + 3) / 4, ws = (j * 7) / 4, we = ((j + 1) * 7 + 3) / 4;
    float s = 0.f;
    for (int y = hs; y < he; ++y) for (int x = ws; x < we; ++x) s += sal[y * 7 + x];
    sal_small[t] = s / (float)((he - hs) * (we - ws));
  }
  __syncthreads();
  float* F[8];
  for (int i = 0; i < 8; ++i) F[i] = o.f[i] + (size_t)b * o.ld[i];
  for (int c = t; c < 512; c += 256) {
    float gmx, gav, tmx, tav, bmx, bav, lmx, lav, rmx, rav, cmx, cav, smx, sav;
    region(fm, c, 0, 7, 0, 7, gmx, gav);
    region(fm, c, 0, 3, 0, 7, tmx, tav);   // top    rows [0, H//2)
    region(fm, c, 3, 7, 0, 7, bmx, bav);   // bottom rows [H//2, H)
    region(fm, c, 0, 7, 0, 3, lmx, lav);   // left
    region(fm, c, 0, 7, 3, 7, rmx, rav);   // right
    region(fm, c, 1, 5, 1, 5, cmx, cav);   // centre [H//4, 3H//4)
    region(fm, c, 1, 6, 1, 6, smx, sav);   // pattern 4 inner region [H//4, H - H//4)
    o.relu_gavg[(size_t)b * 512 + c] = gav > 0.f ? gav : 0.f;
    // patterns 0,1,4: [r1max, r1avg, r2max, r2avg, sal_small(16)] + gmax tiling up to 2592
    float* p0 = F[0]; p0[c] = tmx; p0[512 + c] = tav; p0[1024 + c] = bmx; p0[1536 + c] = bav; p0[2064 + c] = gmx;
    float* p1 = F[1]; p1[c] = lmx; p1[512 + c] = lav; p1[1024 + c] = rmx; p1[1536 + c] = rav; p1[2064 + c] = gmx;
    float* p4 = F[4]; p4[c] = smx; p4[512 + c] = sav; p4[1024 + c] = gmx; p4[1536 + c] = gav; p4[2064 + c] = gmx;
    if (c < 16) { p0[2576 + c] = gmx; p1[2576 + c] = gmx; p4[2576 + c] = gmx; }
    // patterns 2,3: [top max/avg, bottom max/avg, centre max, sal_small] (2576) + 170 of gmax
    for (int q = 2; q <= 3; ++q) {
      float* p = F[q]; p[c] = tmx; p[512 + c] = tav; p[1024 + c] = bmx; p[1536 + c] = bav; p[2048 + c] = cmx;
      if (c < 170) p[2576 + c] = gmx;
    }
    // pattern 5: 4 quadrant max, 4 quadrant avg, centre max, sal_small (4624) + 560 of gmax tiling
    {
      float* p = F[5];
      float m, a;
      region(fm, c, 0, 3, 0, 3, m, a); p[c] = m; p[2048 + c] = a;
      region(fm, c, 0, 3, 3, 7, m, a); p[512 + c] = m; p[2560 + c] = a;
      region(fm, c, 3, 7, 0, 3, m, a); p[1024 + c] = m; p[3072 + c] = a;
      region(fm, c, 3, 7, 3, 7, m, a); p[1536 + c] = m; p[3584 + c] = a;
      p[4096 + c] = cmx;
      p[4624 + c] = gmx;
      if (c < 48) p[4624 + 512 + c] = gmx;
    }
    // pattern 6: 3x3 grid of 2x2 cells (h3 = w3 = 2) max (4608) + sal_small + 672 of gmax tiling
    {
      float* p = F[6];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { float m, a; region(fm, c, 2 * i, 2 * i + 2, 2 * j, 2 * j + 2, m, a); p[(i * 3 + j) * 512 + c] = m; }
      p[4624 + c] = gmx;
      if (c < 160) p[4624 + 512 + c] = gmx;
    }
    // pattern 7: [gmax, gavg, adaptive_avg 2x2 (c*4+i*2+j), adaptive_avg 3x3 (c*9+...), ...] truncated to 7524
    {
      float* p = F[7];
      p[c] = gmx; p[512 + c] = gav;
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          float m, a;
          region(fm, c, (i * 7) / 2, ((i + 1) * 7 + 1) / 2, (j * 7) / 2, ((j + 1) * 7 + 1) / 2, m, a);
          p[1024 + c * 4 + i * 2 + j] = a;
        }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          const int idx = 3072 + c * 9 + i * 3 + j;
          if (idx < 7524) {
            float m, a;
            region(fm, c, (i * 7) / 3, ((i + 1) * 7 + 2) / 3, (j * 7) / 3, ((j + 1) * 7 + 2) / 3, m, a);
            p[idx] = a;
          }
        }
    }
  }
  if (t < 16) {
    const float s = sal_small[t];
    F[0][2048 + t] = s; F[1][2048 + t] = s; F[4][2048 + t] = s;
    F[2][2560 + t] = s; F[3][2560 + t] = s;
    F[5][4608 + t] = s; F[6][4608 + t] = s;
  }
}

// agg[b][j] = sum_i softmax(pw[b])_i * patt[i][b][j]
__global__ void samp_aggregate_kernel(const float* __restrict__ pw, const float* __restrict__ patt, float* __restrict__ agg, int B) {
  const int b = blockIdx.x;
  float w[8], mx = -INFINITY, s = 0.f;
  for (int i = 0; i < 8; ++i) { w[i] = pw[b * 8 + i]; mx = fmaxf(mx, w[i]); }
  for (int i = 0; i < 8; ++i) { w[i] = expf(w[i] - mx); s += w[i]; }
  for (int j = threadIdx.x; j < 1024; j += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += patt[((size_t)i * B + b) * 1024 + j] * (w[i] / s);
    agg[(size_t)b * 1024 + j] = acc;
  }
}

// x: [B,224,224,4], sal: [B,224,224,1]; outputs are device pointers [B*8], [B*6], [B*5]
void sampnet_forward(Ctx& c, const SampModel& m, const Tensor& x, const Tensor& sal, float* pw, float* attrs, float* dist) {
  const size_t mark = c.arena.mark();
  const int B = x.n;
  Tensor fm = resnet_forward(c, m.backbone, x, nullptr);
  FE_CHECK(fm.h == 7 && fm.w == 7 && fm.c == 512 && fm.ld == 512, "SAMP-Net expects a 7x7x512 feature map (224x224 input), got %dx%dx%d", fm.h, fm.w, fm.c);
  // saliency: two 3x3/s2/p1 max pools (samp_net.py:692-695), then bilinear to 7x7 (:613-618)
  Tensor s1 = c.arena.tensor(B, conv_out_dim(sal.h, 3, 2, 1, 1), conv_out_dim(sal.w, 3, 2, 1, 1), 1);
  launch_maxpool(sal, s1, 3, 2, 1, c.stream);
  Tensor s2 = c.arena.tensor(B, conv_out_dim(s1.h, 3, 2, 1, 1), conv_out_dim(s1.w, 3, 2, 1, 1), 1);
  launch_maxpool(s1, s2, 3, 2, 1, c.stream);
  Tensor s7 = c.arena.tensor(B, 7, 7, 1);
  launch_bilinear(s2, s7, c.stream);
  PatOut po;
  for (int i = 0; i < 8; ++i) {
    po.ld[i] = m.pattern[i].Kp;
    po.f[i] = (float*)c.arena.alloc((size_t)B * po.ld[i] * sizeof(float));
    FE_HIP(hipMemsetAsync(po.f[i], 0, (size_t)B * po.ld[i] * sizeof(float), c.stream));
  }
  po.relu_gavg = (float*)c.arena.alloc((size_t)B * 512 * sizeof(float));
  hipLaunchKernelGGL(samp_patterns_kernel, dim3(B), dim3(256), 0, c.stream, fm.p, s7.p, po);
  FE_HIP(hipGetLastError());
  linear_forward(c, m.pattern_weight, po.relu_gavg, 512, B, pw, 8, ACT_NONE);
  float* patt = (float*)c.arena.alloc((size_t)8 * B * 1024 * sizeof(float));
  for (int i = 0; i < 8; ++i)
    linear_forward(c, m.pattern[i], po.f[i], po.ld[i], B, patt + (size_t)i * B * 1024, 1024, ACT_NONE);
  float* agg = (float*)c.arena.alloc((size_t)B * 1024 * sizeof(float));
  hipLaunchKernelGGL(samp_aggregate_kernel, dim3(B), dim3(256), 0, c.stream, pw, patt, agg, B);
  FE_HIP(hipGetLastError());
  float* h1 = (float*)c.arena.alloc((size_t)B * 1024 * sizeof(float));
  float* h2 = (float*)c.arena.alloc((size_t)B * 512 * sizeof(float));
  linear_forward(c, m.att_feat, agg, 1024, B, h2, 512, ACT_RELU);
  linear_forward(c, m.att_pred, h2, 512, B, attrs, 6, ACT_SIGMOID);
  linear_forward(c, m.com0, agg, 1024, B, h1, 1024, ACT_RELU);
  linear_forward(c, m.com3, h1, 1024, B, h2, 512, ACT_RELU);
  linear_forward(c, m.com5, h2, 512, B, dist, 5, ACT_NONE);
  launch_softmax_rows(dist, 5, B, 5, c.stream);
  c.arena.rewind(mark);
}

}  // namespace fe

// U2-Net-P saliency + SAMP-Net composition (reference models/samp_net.py, in-tree definitions).
//   REBNCONV :45-54, RSU7/6/5/4/4F :62-255, U2NETP.forward :298-342, SAMPPModule :429-645, SAMPNet.forward :760-791.
// Every torch.cat of the reference is a channel slice of one NHWC buffer here (producers write straight into
// their slice, consumers read it with the wider pixel stride); BN+bias+ReLU and the RSU residual ride in the
// conv epilogue. Parity: PINNED against the reference's own classes through tests/golden/samp_golden.npz.
#include "engine.h"
#include <cmath>
#include <type_traits>

namespace fe {

// ---------------------------------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------------------------------
static RSUW build_rsu(DeviceWeights& dw, const WeightStore& ws, const std::string& p, int depth, bool dilated) {
  RSUW r;
  r.depth = depth; r.dilated = dilated;
  auto cbr = [&](const std::string& n) {
    ConvW w = build_conv(dw, ws, p + "." + n + ".conv_s1", p + "." + n + ".bn_s1", true);
    const HostTensor& W = ws.get(p + "." + n + ".conv_s1.weight");
    if (dw.prec == PREC_F32 && W.shape.size() == 4 && W.shape[0] == 16 && W.shape[2] == 3 && W.shape[3] == 3 && (W.shape[1] == 16 || W.shape[1] == 32 || W.shape[1] == 64))
      w.wn16 = build_n16_weights(dw, W.data.data(), (int)W.shape[1]);
    return w;
  };
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

template <class T>
static void cbr(Ctx& c, const ConvW& w, const TensorT<T>& x, const TensorT<T>& y, int dil, const TensorT<T>* res_after = nullptr) {
  if constexpr (sizeof(T) == 4) {
    // 16-output-channel layers on large maps: the 16-column matrix instruction, halo-tiled (kernels_n16.hip)
    if (w.wn16 && dil == 1 && !res_after && x.c == w.Cin && y.c == 16 && x.pixels() >= 65536 && x.ld % 4 == 0 && y.ld % 4 == 0 && !getenv("FE_NO_N16")) {
      const double flops = 2.0 * (double)y.pixels() * 9.0 * w.Cin * 16;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c.profile) { FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1)); FE_HIP(hipEventRecord(e0, c.stream)); }
      launch_conv3x3_n16_f32(x, y, w.wn16, w.scale, w.shift, ACT_RELU, c.stream);
      if (c.profile) {
        FE_HIP(hipEventRecord(e1, c.stream)); FE_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        FE_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        char nm[128];
        snprintf(nm, sizeof nm, "halo-tiled conv3x3 -> 16 (16x16x4 f32) M=%d K=%d", (int)y.pixels(), 9 * w.Cin);
        c.timings.push_back({nm, flops, 4.0 * (double)y.pixels() * (w.Cin + 16), ms});
      }
      c.flops_accum += flops;
      return;
    }
  }
  ConvOptsT<T> o; o.ph = o.pw = dil; o.dh = o.dw = dil; o.act = ACT_RELU;
  if (res_after) { o.res = res_after; o.res_after_act = 1; }
  conv_forward(c, w, x, y, o);
}

// One RSU block. x: input view, out: output view (may be a slice of an outer concat buffer). image: the fp32 NHWC4 pixels when
// this is the first block of the network (its 3-channel input convolution runs on the fp32 stem kernel whatever T is).
template <class T>
static void rsu_forward(Ctx& c, const RSUW& r, const TensorT<T>& x, const TensorT<T>& out, const Tensor* image = nullptr) {
  const size_t mark = c.arena.mark();
  const int B = out.n, L = r.depth, mid = r.enc[0].Cout, oc = r.in.Cout;
  TensorT<T> hin;
  if (image) {
    ConvOpts o; o.ph = o.pw = 1; o.act = ACT_RELU;
    hin = first_conv<T>(c, r.in, *image, o);
  } else {
    hin = c.arena.tensor_t<T>(B, x.h, x.w, oc);
    cbr(c, r.in, x, hin, 1);
  }
  const int H0 = hin.h, W0 = hin.w;
  if (r.dilated) {
    // RSU4F: all at one resolution, dilations 1,2,4,8 then 4,2,1 (samp_net.py:232-255)
    TensorT<T> cat[3];  // cat[k] = [decoder-side | h_{k+1}] for k = 0..2
    for (int k = 0; k < 3; ++k) cat[k] = c.arena.tensor_t<T>(B, H0, W0, 2 * mid);
    const int dil[4] = {1, 2, 4, 8};
    TensorT<T> prev = hin;
    for (int k = 0; k < 3; ++k) {
      TensorT<T> hk = cat[k].slice(mid, mid);
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
  std::vector<TensorT<T>> cat(L);  // index 1..L-1 used
  int h = H0, w = W0;
  for (int k = 1; k <= L - 1; ++k) {
    cat[k] = c.arena.tensor_t<T>(B, h, w, 2 * mid);
    if (k < L - 1) { h = ceil_half(h); w = ceil_half(w); }
  }
  cbr(c, r.enc[0], hin, cat[1].slice(mid, mid), 1);
  for (int k = 2; k <= L - 1; ++k) {
    TensorT<T> pooled = c.arena.tensor_t<T>(B, cat[k].h, cat[k].w, mid);
    launch_maxpool(cat[k - 1].slice(mid, mid), pooled, 2, 2, 0, c.stream);
    cbr(c, r.enc[k - 1], pooled, cat[k].slice(mid, mid), 1);
  }
  cbr(c, r.enc[L - 1], cat[L - 1].slice(mid, mid), cat[L - 1].slice(0, mid), 2);  // dilated bottom
  // decoder: dec[0] = rebnconv(L-1)d on cat[L-1]; its output is upsampled into cat[L-2][:mid], ...
  for (int k = L - 1; k >= 2; --k) {
    TensorT<T> d = c.arena.tensor_t<T>(B, cat[k].h, cat[k].w, mid);
    cbr(c, r.dec[L - 1 - k], cat[k], d, 1);
    launch_bilinear(d, cat[k - 1].slice(0, mid), c.stream);
  }
  cbr(c, r.dec[L - 2], cat[1], out, 1, &hin);
  c.arena.rewind(mark);
}

// x: fp32 [B,H,W,4] (normalised image, 4th channel zero) -> sal: [B,H,W,1] (sigmoid of the fused map d0), activations of type T
template <class T>
void u2netp_forward(Ctx& c, const U2NetPModel& m, const Tensor& x, const TensorT<T>& sal) {
  const size_t mark = c.arena.mark();
  const int B = x.n;
  int hs[6], wsz[6];
  hs[0] = x.h; wsz[0] = x.w;
  for (int i = 1; i < 6; ++i) { hs[i] = ceil_half(hs[i - 1]); wsz[i] = ceil_half(wsz[i - 1]); }
  // decoder concat buffers: catd[i] = [up(deeper decoder out) | encoder out hx_{i+1}] at level i (0..4)
  TensorT<T> catd[5];
  for (int i = 0; i < 5; ++i) catd[i] = c.arena.tensor_t<T>(B, hs[i], wsz[i], 128);
  TensorT<T> hx6 = c.arena.tensor_t<T>(B, hs[5], wsz[5], 64);
  TensorT<T> in;
  for (int i = 0; i < 6; ++i) {
    TensorT<T> o = i < 5 ? catd[i].slice(64, 64) : hx6;
    rsu_forward(c, m.stage[i], in, o, i == 0 ? &x : nullptr);
    if (i < 5) {
      TensorT<T> p = c.arena.tensor_t<T>(B, hs[i + 1], wsz[i + 1], 64);
      launch_maxpool(o, p, 2, 2, 0, c.stream);
      in = p;
    }
  }
  // side maps into one 8-channel buffer (6 used; outconv's packed weight has zeros for channels 6,7)
  TensorT<T> sides = c.arena.tensor_t<T>(B, x.h, x.w, 8);
  FE_HIP(hipMemsetAsync(sides.p, 0, sides.numel() * sizeof(T), c.stream));
  auto side = [&](int k, const TensorT<T>& feat) {
    ConvOptsT<T> o; o.ph = o.pw = 1;
    if (feat.h == x.h && feat.w == x.w) {
      conv_forward(c, m.side[k], feat, sides.slice(k, 1), o);
    } else {
      TensorT<T> s = c.arena.tensor_t<T>(B, feat.h, feat.w, 1);
      conv_forward(c, m.side[k], feat, s, o);
      launch_bilinear(s, sides.slice(k, 1), c.stream);
    }
  };
  side(5, hx6);
  TensorT<T> deeper = hx6;
  TensorT<T> dec[5];
  for (int i = 4; i >= 0; --i) {  // stage5d, 4d, 3d, 2d, 1d = stage[6 + (4 - i)]
    launch_bilinear(deeper, catd[i].slice(0, 64), c.stream);
    dec[i] = c.arena.tensor_t<T>(B, hs[i], wsz[i], 64);
    rsu_forward(c, m.stage[6 + (4 - i)], catd[i], dec[i]);
    side(i, dec[i]);
    deeper = dec[i];
  }
  ConvOptsT<T> oo; oo.act = ACT_SIGMOID;
  conv_forward(c, m.outconv, sides, sal, oo);
  c.arena.rewind(mark);
}
template void u2netp_forward<float>(Ctx&, const U2NetPModel&, const Tensor&, const Tensor&);
template void u2netp_forward<bf16>(Ctx&, const U2NetPModel&, const Tensor&, const TensorH&);
template void u2netp_forward<f16>(Ctx&, const U2NetPModel&, const Tensor&, const TensorF16&);

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
template <class T>
struct PatOut { T* f[8]; int ld[8]; float* relu_gavg; };

template <class T>
__device__ inline void region(const T* fm, int c, int r0, int r1, int c0, int c1, float& mx, float& av) {
  float m = -INFINITY, s = 0.f;
  for (int y = r0; y < r1; ++y)
    for (int x = c0; x < c1; ++x) { const float v = ldf(fm + (y * 7 + x) * 512 + c); m = fmaxf(m, v); s += v; }
  mx = m; av = s / (float)((r1 - r0) * (c1 - c0));
}

// TF: element type of the feature map, TS: of the saliency map, T: of the pattern vectors written
template <class TF, class TS, class T>
__global__ void samp_patterns_kernel(const TF* __restrict__ fmap, const TS* __restrict__ sal7, PatOut<T> o) {
  const int b = blockIdx.x, t = threadIdx.x;
  const TF* fm = fmap + (size_t)b * 49 * 512;
  const TS* sal = sal7 + (size_t)b * 49;
  __shared__ float sal_small[16];
  if (t < 16) {  // adaptive_avg_pool2d(7x7 -> 4x4)
    const int i = t / 4, j = t % 4;
    const int hs = (i * 7) / 4, he = ((i + 1) * 7 + 3) / 4, ws = (j * 7) / 4, we = ((j + 1) * 7 + 3) / 4;
    float s = 0.f;
    for (int y = hs; y < he; ++y) for (int x = ws; x < we; ++x) s += ldf(sal + y * 7 + x);
    sal_small[t] = s / (float)((he - hs) * (we - ws));
  }
  __syncthreads();
  T* F[8];
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
    T* p0 = F[0]; stf(p0 + c, tmx); stf(p0 + 512 + c, tav); stf(p0 + 1024 + c, bmx); stf(p0 + 1536 + c, bav); stf(p0 + 2064 + c, gmx);
    T* p1 = F[1]; stf(p1 + c, lmx); stf(p1 + 512 + c, lav); stf(p1 + 1024 + c, rmx); stf(p1 + 1536 + c, rav); stf(p1 + 2064 + c, gmx);
    T* p4 = F[4]; stf(p4 + c, smx); stf(p4 + 512 + c, sav); stf(p4 + 1024 + c, gmx); stf(p4 + 1536 + c, gav); stf(p4 + 2064 + c, gmx);
    if (c < 16) { stf(p0 + 2576 + c, gmx); stf(p1 + 2576 + c, gmx); stf(p4 + 2576 + c, gmx); }
    // patterns 2,3: [top max/avg, bottom max/avg, centre max, sal_small] (2576) + 170 of gmax
    for (int q = 2; q <= 3; ++q) {
      T* p = F[q]; stf(p + c, tmx); stf(p + 512 + c, tav); stf(p + 1024 + c, bmx); stf(p + 1536 + c, bav); stf(p + 2048 + c, cmx);
      if (c < 170) stf(p + 2576 + c, gmx);
    }
    // pattern 5: 4 quadrant max, 4 quadrant avg, centre max, sal_small (4624) + 560 of gmax tiling
    {
      T* p = F[5];
      float m, a;
      region(fm, c, 0, 3, 0, 3, m, a); stf(p + c, m); stf(p + 2048 + c, a);
      region(fm, c, 0, 3, 3, 7, m, a); stf(p + 512 + c, m); stf(p + 2560 + c, a);
      region(fm, c, 3, 7, 0, 3, m, a); stf(p + 1024 + c, m); stf(p + 3072 + c, a);
      region(fm, c, 3, 7, 3, 7, m, a); stf(p + 1536 + c, m); stf(p + 3584 + c, a);
      stf(p + 4096 + c, cmx);
      stf(p + 4624 + c, gmx);
      if (c < 48) stf(p + 4624 + 512 + c, gmx);
    }
    // pattern 6: 3x3 grid of 2x2 cells (h3 = w3 = 2) max (4608) + sal_small + 672 of gmax tiling
    {
      T* p = F[6];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { float m, a; region(fm, c, 2 * i, 2 * i + 2, 2 * j, 2 * j + 2, m, a); stf(p + (i * 3 + j) * 512 + c, m); }
      stf(p + 4624 + c, gmx);
      if (c < 160) stf(p + 4624 + 512 + c, gmx);
    }
    // pattern 7: [gmax, gavg, adaptive_avg 2x2 (c*4+i*2+j), adaptive_avg 3x3 (c*9+...), ...] truncated to 7524
    {
      T* p = F[7];
      stf(p + c, gmx); stf(p + 512 + c, gav);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          float m, a;
          region(fm, c, (i * 7) / 2, ((i + 1) * 7 + 1) / 2, (j * 7) / 2, ((j + 1) * 7 + 1) / 2, m, a);
          stf(p + 1024 + c * 4 + i * 2 + j, a);
        }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          const int idx = 3072 + c * 9 + i * 3 + j;
          if (idx < 7524) {
            float m, a;
            region(fm, c, (i * 7) / 3, ((i + 1) * 7 + 2) / 3, (j * 7) / 3, ((j + 1) * 7 + 2) / 3, m, a);
            stf(p + idx, a);
          }
        }
    }
  }
  if (t < 16) {
    const float s = sal_small[t];
    stf(F[0] + 2048 + t, s); stf(F[1] + 2048 + t, s); stf(F[4] + 2048 + t, s);
    stf(F[2] + 2560 + t, s); stf(F[3] + 2560 + t, s);
    stf(F[5] + 4608 + t, s); stf(F[6] + 4608 + t, s);
  }
}

// agg[b][j] = sum_i softmax(pw[b])_i * patt[i][b][j]   (fp32 result whatever the activation type)
template <class T>
__global__ void samp_aggregate_kernel(const float* __restrict__ pw, const T* __restrict__ patt, float* __restrict__ agg, int B) {
  const int b = blockIdx.x;
  float w[8], mx = -INFINITY, s = 0.f;
  for (int i = 0; i < 8; ++i) { w[i] = pw[b * 8 + i]; mx = fmaxf(mx, w[i]); }
  for (int i = 0; i < 8; ++i) { w[i] = expf(w[i] - mx); s += w[i]; }
  for (int j = threadIdx.x; j < 1024; j += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += ldf(patt + ((size_t)i * B + b) * 1024 + j) * (w[i] / s);
    agg[(size_t)b * 1024 + j] = acc;
  }
}

// x: fp32 [B,224,224,4], sal: [B,224,224,1] (type TS); outputs are fp32 device pointers [B*8], [B*6], [B*5].
// T = operand type of the trunk and the pattern module; the attribute / composition heads behind the pattern aggregation
// (a 1024-vector per image) always run in fp32. Under FE_PRECISION_RES32 (2-byte T) the trunk keeps its skip stream in fp32 and the
// pattern module reads the fp32 feature map, pools in fp32 and multiplies the fp32 vectors with the 2-byte weights (fp32 out): no
// activation behind the trunk is rounded.
template <class T, class TS>
void sampnet_forward(Ctx& c, const SampModel& m, const Tensor& x, const TensorT<TS>& sal, float* pw, float* attrs, float* dist) {
  const size_t mark = c.arena.mark();
  const int B = x.n;
  const bool r32 = sizeof(T) == 2 && m.dw.res32;
  Tensor fm32;
  TensorT<T> fm = resnet_forward<T>(c, m.backbone, x, nullptr, r32, &fm32);
  FE_CHECK(fm.h == 7 && fm.w == 7 && fm.c == 512 && fm.ld == 512, "SAMP-Net expects a 7x7x512 feature map (224x224 input), got %dx%dx%d", fm.h, fm.w, fm.c);
  // saliency: two 3x3/s2/p1 max pools (samp_net.py:692-695), then bilinear to 7x7 (:613-618)
  TensorT<TS> s1 = c.arena.tensor_t<TS>(B, conv_out_dim(sal.h, 3, 2, 1, 1), conv_out_dim(sal.w, 3, 2, 1, 1), 1);
  launch_maxpool(sal, s1, 3, 2, 1, c.stream);
  TensorT<TS> s2 = c.arena.tensor_t<TS>(B, conv_out_dim(s1.h, 3, 2, 1, 1), conv_out_dim(s1.w, 3, 2, 1, 1), 1);
  launch_maxpool(s1, s2, 3, 2, 1, c.stream);
  TensorT<TS> s7 = c.arena.tensor_t<TS>(B, 7, 7, 1);
  launch_bilinear(s2, s7, c.stream);
  float* relu_gavg = c.arena.array<float>((size_t)B * 512);
  float* agg = c.arena.array<float>((size_t)B * 1024);
  auto pattern_stage = [&](auto* ftag, auto* ptag) {
    typedef std::remove_pointer_t<decltype(ftag)> TF;      // feature-map element
    typedef std::remove_pointer_t<decltype(ptag)> TP;      // pattern-vector element
    PatOut<TP> po;
    for (int i = 0; i < 8; ++i) {
      po.ld[i] = sizeof(T) == 2 ? m.pattern[i].KpH : m.pattern[i].Kp;   // zero padded up to the packed K of the pattern "convs"
      po.f[i] = c.arena.array<TP>((size_t)B * po.ld[i]);
      FE_HIP(hipMemsetAsync(po.f[i], 0, (size_t)B * po.ld[i] * sizeof(TP), c.stream));
    }
    po.relu_gavg = relu_gavg;
    const TF* fmp;
    if constexpr (sizeof(TF) == sizeof(T)) fmp = reinterpret_cast<const TF*>(fm.p); else fmp = reinterpret_cast<const TF*>(fm32.p);
    hipLaunchKernelGGL((samp_patterns_kernel<TF, TS, TP>), dim3(B), dim3(256), 0, c.stream, fmp, (const TS*)s7.p, po);
    FE_HIP(hipGetLastError());
    linear_forward(c, m.pattern_weight, (const float*)relu_gavg, 512, B, pw, 8, ACT_NONE);
    TP* patt = c.arena.array<TP>((size_t)8 * B * 1024);
    for (int i = 0; i < 8; ++i) {
      if constexpr (sizeof(TP) == 4 && sizeof(T) == 2) linear_forward_xf32(c, m.pattern[i], m.dw.prec, (const float*)po.f[i], po.ld[i], B, (float*)(patt + (size_t)i * B * 1024), 1024, ACT_NONE);
      else linear_forward(c, m.pattern[i], (const T*)po.f[i], po.ld[i], B, (T*)(patt + (size_t)i * B * 1024), 1024, ACT_NONE);
    }
    hipLaunchKernelGGL(samp_aggregate_kernel<TP>, dim3(B), dim3(256), 0, c.stream, pw, (const TP*)patt, agg, B);
    FE_HIP(hipGetLastError());
  };
  if (r32) pattern_stage((float*)nullptr, (float*)nullptr);
  else pattern_stage((T*)nullptr, (T*)nullptr);
  float* h1 = c.arena.array<float>((size_t)B * 1024);
  float* h2 = c.arena.array<float>((size_t)B * 512);
  linear_forward(c, m.att_feat, (const float*)agg, 1024, B, h2, 512, ACT_RELU);
  linear_forward(c, m.att_pred, (const float*)h2, 512, B, attrs, 6, ACT_SIGMOID);
  linear_forward(c, m.com0, (const float*)agg, 1024, B, h1, 1024, ACT_RELU);
  linear_forward(c, m.com3, (const float*)h1, 1024, B, h2, 512, ACT_RELU);
  linear_forward(c, m.com5, (const float*)h2, 512, B, dist, 5, ACT_NONE);
  launch_softmax_rows(dist, 5, B, 5, c.stream);
  c.arena.rewind(mark);
}
#define FE_SAMP_INST(T, TS) template void sampnet_forward<T, TS>(Ctx&, const SampModel&, const Tensor&, const TensorT<TS>&, float*, float*, float*);
FE_SAMP_INST(float, float) FE_SAMP_INST(float, bf16) FE_SAMP_INST(float, f16)
FE_SAMP_INST(bf16, float) FE_SAMP_INST(bf16, bf16) FE_SAMP_INST(bf16, f16)
FE_SAMP_INST(f16, float) FE_SAMP_INST(f16, bf16) FE_SAMP_INST(f16, f16)
#undef FE_SAMP_INST

}  // namespace fe

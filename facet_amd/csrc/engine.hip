// Engine runtime: weight packing, conv/linear wrappers and the ResNet graphs.
#include "engine.h"
#include "onnx_graph.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace fe {

void WeightStore::set(const std::string& name, const float* data, const int64_t* shape, int ndim) {
  HostTensor h;
  h.shape.assign(shape, shape + ndim);
  h.data.assign(data, data + h.numel());
  t_[name] = std::move(h);
}
const HostTensor& WeightStore::get(const std::string& name) const {
  auto it = t_.find(name);
  if (it == t_.end()) throw Error("missing weight tensor '" + name + "'");
  return it->second;
}

float* DeviceWeights::upload(const std::vector<float>& v) {
  void* p = nullptr;
  const size_t b = v.size() * sizeof(float);
  FE_HIP(hipMalloc(&p, b ? b : 16));
  if (b) FE_HIP(hipMemcpy(p, v.data(), b, hipMemcpyHostToDevice));
  ptrs_.push_back(p);
  bytes_ += b;
  return (float*)p;
}
void* DeviceWeights::upload_raw(const void* data, size_t bytes) {
  void* p = nullptr;
  FE_HIP(hipMalloc(&p, bytes ? bytes : 16));
  if (bytes) FE_HIP(hipMemcpy(p, data, bytes, hipMemcpyHostToDevice));
  ptrs_.push_back(p);
  bytes_ += bytes;
  return p;
}
void DeviceWeights::release() {
  for (void* p : ptrs_) (void)hipFree(p);
  ptrs_.clear();
  bytes_ = 0;
}

const float* to_f32(Ctx& c, const bf16* p, size_t n) {
  float* f = c.arena.array<float>(n);
  launch_convert(p, f, n, c.stream);
  return f;
}
const float* to_f32(Ctx& c, const f16* p, size_t n) {
  float* f = c.arena.array<float>(n);
  launch_convert(p, f, n, c.stream);
  return f;
}

Ctx::~Ctx() {
  topiq.reset();
  arena.release();
  if (stream) (void)hipStreamDestroy(stream);
}

ConvW pack_conv(DeviceWeights& dw, const HostTensor& w, const std::vector<float>* scale, const std::vector<float>* shift) {
  FE_CHECK(w.shape.size() == 4 || w.shape.size() == 2, "conv weight rank %zu", w.shape.size());
  ConvW c;
  c.Cout = (int)w.shape[0];
  c.Cin = (int)w.shape[1];
  c.KH = w.shape.size() == 4 ? (int)w.shape[2] : 1;
  c.KW = w.shape.size() == 4 ? (int)w.shape[3] : 1;
  c.CinPad = (c.Cin + 3) & ~3;
  c.K = c.KH * c.KW * c.CinPad;
  c.Kp = (c.K + CONV_KALIGN - 1) / CONV_KALIGN * CONV_KALIGN;
  std::vector<float> packed((size_t)c.Cout * c.Kp, 0.f);
  // K order: (tap, ci) for the generic kernel; for CinPad % 16 == 0 (the fast kernels) 16-channel block outer, tap
  // inner, ci-in-block innermost, so consecutive K slabs are the taps of one channel block.
  const bool blocked = (c.CinPad % 16 == 0);
  const int ntaps = c.KH * c.KW;
  for (int co = 0; co < c.Cout; ++co)
    for (int ci = 0; ci < c.Cin; ++ci)
      for (int kh = 0; kh < c.KH; ++kh)
        for (int kw = 0; kw < c.KW; ++kw) {
          const int tap = kh * c.KW + kw;
          const size_t k = blocked ? ((size_t)(ci / 16) * ntaps + tap) * 16 + (ci % 16) : (size_t)tap * c.CinPad + ci;
          packed[(size_t)co * c.Kp + k] = w.data[(((size_t)co * c.Cin + ci) * c.KH + kh) * c.KW + kw];
        }
  // (half_only models: the 2-byte form below is the only one their layers run on)
  const bool skip_f32 = dw.half_only && dw.prec != PREC_F32 && (c.KH * c.KW == 1 || ((c.Cin + 7) & ~7) % 16 == 0);
  if (!skip_f32) c.w = dw.upload(packed);
  if (!skip_f32 && c.Cout <= 2 && c.KH * c.KW > 1 && c.CinPad % 16 == 0) {
    // tap-decomposed form: z[pixel][tap*Cout+co] = <x[pixel], w[co][tap]> as a 1x1 conv, neighbours summed afterwards
    const int T = c.KH * c.KW * c.Cout;
    c.KpT = (c.CinPad + CONV_KALIGN - 1) / CONV_KALIGN * CONV_KALIGN;
    std::vector<float> tp((size_t)T * c.KpT, 0.f);
    for (int t = 0; t < c.KH * c.KW; ++t)
      for (int co = 0; co < c.Cout; ++co)
        for (int ci = 0; ci < c.Cin; ++ci)
          tp[(size_t)(t * c.Cout + co) * c.KpT + ci] = w.data[(((size_t)co * c.Cin + ci) * c.KH + t / c.KW) * c.KW + t % c.KW];
    c.wtap = dw.upload(tp);
  }
  if (c.CinPad == 4 && c.KH == c.KW && (c.KH == 7 || c.KH == 3) && (c.Cout == 32 || c.Cout == 64)) {
    // stem layout; a 4th input channel must be all-zero weights (padding), the stem kernel never multiplies it
    bool ch3_zero = true;
    const int taps = c.KH * c.KW, taps2 = (taps + 1) & ~1;
    for (int co = 0; co < c.Cout && c.Cin == 4; ++co)
      for (int t = 0; t < taps; ++t) ch3_zero = ch3_zero && w.data[((size_t)co * c.Cin + 3) * taps + t] == 0.f;
    if (ch3_zero) {
      std::vector<float> st((size_t)taps2 * c.Cout * 4, 0.f);
      for (int co = 0; co < c.Cout; ++co)
        for (int ci = 0; ci < c.Cin && ci < 3; ++ci)
          for (int t = 0; t < taps; ++t) st[((size_t)t * c.Cout + co) * 4 + ci] = w.data[((size_t)co * c.Cin + ci) * taps + t];
      c.wstem = dw.upload(st);
    }
  }
  static const int wino_min_cin = getenv("FE_WINO_MIN_CIN") ? atoi(getenv("FE_WINO_MIN_CIN")) : 96;   // tuning hooks; defaults measured best (profiles/r01_README.md)
  static const int wino_form = getenv("FE_WINO_FORM") ? atoi(getenv("FE_WINO_FORM")) : 4;
  // (Winograd forms serve the fp32 conv path only: a model committed under a 2-byte precision never runs them - its layers take the
  // 2-byte kernel, or the direct fp32 kernel through the conversion fallback)
  if (!skip_f32 && dw.prec == PREC_F32 && (wino_form == 2 || wino_form == 4) && c.KH == 3 && c.KW == 3 && c.Cin == c.CinPad && c.Cin % 32 == 0 && c.Cin >= wino_min_cin && c.Cout % 4 == 0) {
    // Winograd weights U = G g G^T per (cout, cin), in double. Layout [planes][Cout][Cin]; the epilogue scale / shift / activation
    // are applied by the output transform (ConvW.scale may be attached after packing).
    static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    static const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                    {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int R = wino_form == 2 ? 4 : 6;
    auto Gm = [&](int a, int q) { return wino_form == 2 ? G2[a][q] : G4[a][q]; };
    std::vector<float> U((size_t)R * R * c.Cout * c.Cin);
    for (int co = 0; co < c.Cout; ++co)
      for (int ci = 0; ci < c.Cin; ++ci) {
        const float* g = &w.data[((size_t)co * c.Cin + ci) * 9];
        double tmp[6][3];
        for (int a = 0; a < R; ++a)
          for (int q = 0; q < 3; ++q) tmp[a][q] = Gm(a, 0) * g[q] + Gm(a, 1) * g[3 + q] + Gm(a, 2) * g[6 + q];
        for (int a = 0; a < R; ++a)
          for (int b = 0; b < R; ++b) {
            const double u = tmp[a][0] * Gm(b, 0) + tmp[a][1] * Gm(b, 1) + tmp[a][2] * Gm(b, 2);
            U[((size_t)(R * a + b) * c.Cout + co) * c.Cin + ci] = (float)u;
          }
      }
    c.wino = dw.upload(U);
    c.wino_m = wino_form;
  }
  if (scale) c.scale = dw.upload(*scale);
  if (shift) c.shift = dw.upload(*shift);
  if (dw.prec != PREC_F32) pack_conv_bf16(dw, w, c);
  return c;
}

// 2-byte form (bf16 or f16, the precision of `dw`): [Cout][KpH], K order = channel block (cb) outer, tap inner, channel-in-block innermost (kernels_conv_bf16.hip);
// 1x1 kernels keep plain channel order with Cin rounded up to 8 (the kernel zero-fills the chunk past Cin).
void pack_conv_bf16(DeviceWeights& dw, const HostTensor& w, ConvW& c) {
  const int ntaps = c.KH * c.KW;
  c.CinPadH = (c.Cin + 7) & ~7;
  if (ntaps > 1) {
    if (c.CinPadH % 32 == 0) c.cb = 32;
    else if (c.CinPadH % 16 == 0) c.cb = 16;
    else return;                       // 3-channel first layers: no bf16 form (fp32 stem / generic kernel, bf16 output)
    if (ntaps >= 64) return;
  } else {
    c.cb = 32;
  }
  const int K = ntaps * c.CinPadH;
  c.KpH = (K + CONV_KALIGN_H - 1) / CONV_KALIGN_H * CONV_KALIGN_H;
  std::vector<uint16_t> packed((size_t)c.Cout * c.KpH, 0);
  for (int co = 0; co < c.Cout; ++co)
    for (int ci = 0; ci < c.Cin; ++ci)
      for (int t = 0; t < ntaps; ++t) {
        const size_t k = ntaps > 1 ? ((size_t)(ci / c.cb) * ntaps + t) * c.cb + (ci % c.cb) : (size_t)ci;
        packed[(size_t)co * c.KpH + k] = f32_to_half_bits(w.data[((size_t)co * c.Cin + ci) * ntaps + t], dw.prec);
      }
  c.wh = dw.upload_raw(packed.data(), packed.size() * sizeof(uint16_t));
  c.hprec = dw.prec;
  if (c.Cout <= 2 && ntaps > 1) {
    // tap-decomposed form (see pack_conv): rows (tap, co) of a 1x1 conv, row count rounded up to 8 for 16-byte stores
    const int T = ntaps * c.Cout, T8 = (T + 7) & ~7;
    c.KpTH = (c.CinPadH + CONV_KALIGN_H - 1) / CONV_KALIGN_H * CONV_KALIGN_H;
    std::vector<uint16_t> tp((size_t)T8 * c.KpTH, 0);
    for (int t = 0; t < ntaps; ++t)
      for (int co = 0; co < c.Cout; ++co)
        for (int ci = 0; ci < c.Cin; ++ci)
          tp[(size_t)(t * c.Cout + co) * c.KpTH + ci] = f32_to_half_bits(w.data[((size_t)co * c.Cin + ci) * ntaps + t], dw.prec);
    c.wtap_h = dw.upload_raw(tp.data(), tp.size() * sizeof(uint16_t));
  }
}

ConvW build_conv(DeviceWeights& dw, const WeightStore& ws, const std::string& conv_prefix,
                 const std::string& bn_prefix, bool conv_bias, float bn_eps) {
  const HostTensor& w = ws.get(conv_prefix + ".weight");
  const int cout = (int)w.shape[0];
  std::vector<float> scale, shift;
  const bool bn = !bn_prefix.empty();
  if (bn) {
    const auto& g = ws.get(bn_prefix + ".weight").data;
    const auto& b = ws.get(bn_prefix + ".bias").data;
    const auto& mu = ws.get(bn_prefix + ".running_mean").data;
    const auto& var = ws.get(bn_prefix + ".running_var").data;
    FE_CHECK((int)g.size() == cout && (int)var.size() == cout, "bn %s size mismatch", bn_prefix.c_str());
    scale.resize(cout); shift.resize(cout);
    for (int i = 0; i < cout; ++i) {
      const float inv = 1.0f / std::sqrt(var[i] + bn_eps);
      scale[i] = g[i] * inv;
      float sh = b[i] - mu[i] * scale[i];
      if (conv_bias) sh += ws.get(conv_prefix + ".bias").data[i] * scale[i];
      shift[i] = sh;
    }
  } else if (conv_bias) {
    shift = ws.get(conv_prefix + ".bias").data;
    FE_CHECK((int)shift.size() == cout, "bias %s size mismatch", conv_prefix.c_str());
  }
  return pack_conv(dw, w, scale.empty() ? nullptr : &scale, shift.empty() ? nullptr : &shift);
}

ConvW build_linear(DeviceWeights& dw, const WeightStore& ws, const std::string& prefix, bool bias) {
  const HostTensor& w = ws.get(prefix + ".weight");
  FE_CHECK(w.shape.size() == 2, "linear %s weight rank", prefix.c_str());
  if (bias) {
    const auto& b = ws.get(prefix + ".bias").data;
    return pack_conv(dw, w, nullptr, &b);
  }
  return pack_conv(dw, w, nullptr, nullptr);
}

ConvW build_linear_rows(DeviceWeights& dw, const HostTensor& w, const HostTensor* b, int row0, int rows) {
  FE_CHECK(w.shape.size() == 2 && row0 + rows <= w.shape[0], "linear_rows: bad range");
  HostTensor sub;
  const int in = (int)w.shape[1];
  sub.shape = {rows, in};
  sub.data.assign(w.data.begin() + (size_t)row0 * in, w.data.begin() + (size_t)(row0 + rows) * in);
  if (b) {
    std::vector<float> bb(b->data.begin() + row0, b->data.begin() + row0 + rows);
    return pack_conv(dw, sub, nullptr, &bb);
  }
  return pack_conv(dw, sub, nullptr, nullptr);
}

LayerNormW build_ln(DeviceWeights& dw, const WeightStore& ws, const std::string& prefix, float eps) {
  LayerNormW l;
  const auto& g = ws.get(prefix + ".weight").data;
  l.d = (int)g.size();
  l.g = dw.upload(g);
  l.b = dw.upload(ws.get(prefix + ".bias").data);
  l.eps = eps;
  return l;
}

void conv_forward(Ctx& c, const ConvW& w, const Tensor& x, const Tensor& y, const ConvOpts& o) {
  FE_CHECK(x.c == w.CinPad, "conv: input channels %d != packed Cin %d", x.c, w.CinPad);
  FE_CHECK(y.c == w.Cout && y.n == x.n, "conv: output view mismatch (c=%d Cout=%d)", y.c, w.Cout);
  ConvParams p{};
  p.x = x.p; p.ldx = x.ld;
  p.w = w.w; p.scale = w.scale; p.shift = w.shift; p.slope = w.slope;
  FE_CHECK(o.act != ACT_PRELU || w.slope, "conv: PReLU without slopes");
  if (o.res) {
    FE_CHECK(o.res->c == y.c && o.res->pixels() == y.pixels(), "conv: residual shape mismatch");
    p.res = o.res->p; p.ldr = o.res->ld;
  }
  if (o.gate) {
    FE_CHECK(o.gate->pixels() == y.pixels() && (o.gate->c == 1 || o.gate->c == y.c), "conv: gate shape mismatch");
    p.gate = o.gate->p; p.ldg = o.gate->ld; p.gate_c1 = o.gate->c == 1;
  }
  p.y = y.p; p.ldy = y.ld;
  p.N = x.n; p.H = x.h; p.W = x.w; p.Cin = w.CinPad;
  p.Ho = y.h; p.Wo = y.w; p.Cout = w.Cout;
  p.KH = w.KH; p.KW = w.KW; p.sh = o.sh; p.sw = o.sw; p.ph = o.ph; p.pw = o.pw; p.dh = o.dh; p.dw = o.dw;
  FE_CHECK(y.h == conv_out_dim(x.h, w.KH, o.sh, o.ph, o.dh) && y.w == conv_out_dim(x.w, w.KW, o.sw, o.pw, o.dw),
           "conv: output dims %dx%d inconsistent with input %dx%d", y.h, y.w, x.h, x.w);
  p.K = w.K; p.Kp = w.Kp;
  p.M = (int)y.pixels();
  FE_CHECK(y.pixels() < (1ull << 31), "conv: M too large");
  p.act = o.act; p.res_after_act = o.res_after_act;
  p.variant = c.force_variant;
  if (w.wtap && o.act != ACT_PRELU && o.sh == 1 && o.sw == 1 && !o.res && !o.gate && p.variant == 0 && y.h == x.h && y.w == x.w) {
    // narrow spatial conv (Cout <= 2): 1x1 conv to per-tap partials on the matrix cores + a gather-sum pass
    const int T = w.KH * w.KW * w.Cout, Tp = (T + 3) & ~3;
    const size_t mark = c.arena.mark();
    Tensor z = c.arena.tensor(x.n, x.h, x.w, Tp);
    ConvParams q{};
    q.x = x.p; q.ldx = x.ld; q.w = w.wtap; q.y = z.p; q.ldy = Tp;
    q.N = x.n; q.H = x.h; q.W = x.w; q.Cin = w.CinPad; q.Ho = x.h; q.Wo = x.w; q.Cout = T;
    q.KH = q.KW = 1; q.sh = q.sw = q.dh = q.dw = 1;
    q.K = w.CinPad; q.Kp = w.KpT; q.M = (int)x.pixels();
    launch_conv(q, c.stream);
    launch_tap_gather(z.p, Tp, x.n, x.h, x.w, w.KH, w.KW, o.ph, o.pw, o.dh, o.dw, w.Cout, w.scale, w.shift, o.act, y.p,
                      y.ld, y.h, y.w, c.stream);
    c.arena.rewind(mark);
    c.flops_accum += 2.0 * p.M * (double)(w.KH * w.KW * w.Cin) * (w.CoutAlg ? w.CoutAlg : p.Cout);
    return;
  }
  // Winograd F(4x4,3x3) (or F(2x2,3x3)) for the deep 3x3 stride-1 layers: input transform -> 36 (16) batched GEMMs in ONE launch
  // -> output transform with the epilogue
  static const bool no_wino = getenv("FE_NO_WINO") != nullptr;
  const bool wino_plain = !o.res && (o.act == ACT_NONE || o.act == ACT_RELU);     // F(2x2) handles only these; F(4x4) also residual / PReLU
  if (w.wino && !no_wino && p.variant == 0 && o.sh == 1 && o.sw == 1 && o.ph == 1 && o.pw == 1 && o.dh == 1 && o.dw == 1 &&
      !o.gate && (wino_plain || (w.wino_m == 4 && (o.act == ACT_NONE || o.act == ACT_RELU || o.act == ACT_PRELU))) && y.h == x.h && y.w == x.w && (((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0 &&
      x.ld % 4 == 0 && y.ld % 4 == 0) {
    const int ts = w.wino_m, planes = (ts + 2) * (ts + 2);       // output tile side 2 or 4; 16 or 36 planes
    const int th = (x.h + ts - 1) / ts, tw = (x.w + ts - 1) / ts;
    const size_t tiles = (size_t)x.n * th * tw;
    const size_t wino_bytes = (size_t)planes * tiles * ((size_t)w.CinPad + w.Cout) * sizeof(float) + 1024;
    // falls through to the direct kernel when the two plane buffers do not fit what is left of the arena
    if (tiles < (1u << 30) / 36 && c.arena.mark() + wino_bytes <= c.arena.capacity()) {
      const size_t mark = c.arena.mark();
      float* V = (float*)c.arena.alloc((size_t)planes * tiles * w.CinPad * sizeof(float));
      float* Mb = (float*)c.arena.alloc((size_t)planes * tiles * w.Cout * sizeof(float));
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c.profile) {
        FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1));
        FE_HIP(hipEventRecord(e0, c.stream));
      }
      if (ts == 2) launch_wino_input(x, th, tw, V, c.stream);
      else launch_wino4_input(x, th, tw, V, c.stream);
      ConvParams g{};
      // physical channel count everywhere below: ConvW.Cin may hold the unpadded (algorithmic) count of a graph layer
      g.x = V; g.ldx = w.CinPad; g.w = w.wino; g.ldw = w.CinPad; g.y = Mb; g.ldy = w.Cout;
      g.M = (int)tiles; g.K = w.CinPad; g.Cout = w.Cout;
      g.N = 1; g.H = 1; g.W = g.M; g.Ho = 1; g.Wo = g.M; g.KH = g.KW = 1; g.sh = g.sw = g.dh = g.dw = 1;
      g.Kp = w.CinPad; g.Cin = w.CinPad;
      g.batch = planes; g.nb1 = 1; g.xs2 = (long long)tiles * w.CinPad; g.ws2 = (long long)w.Cout * w.CinPad; g.ys2 = (long long)tiles * w.Cout;
      launch_conv(g, c.stream);
      if (ts == 2) launch_wino_output(Mb, y, th, tw, w.scale, w.shift, o.act == ACT_RELU, c.stream);
      else launch_wino4_output(Mb, y, th, tw, w.scale, w.shift, o.act, w.slope, o.res, o.res_after_act, c.stream);
      if (c.profile) {   // one record for the three launches, with the direct convolution's FLOPs
        FE_HIP(hipEventRecord(e1, c.stream));
        FE_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        FE_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        char nm[128];
        snprintf(nm, sizeof nm, "wino3x3 M=%d K=%d N=%d", p.M, p.K, p.Cout);
        c.timings.push_back({nm, 2.0 * p.M * (double)(9 * w.Cin) * p.Cout, 0.0, ms});
      }
      c.arena.rewind(mark);
      c.flops_accum += 2.0 * p.M * (double)(w.KH * w.KW * w.Cin) * (w.CoutAlg ? w.CoutAlg : p.Cout);   // algorithmic = the direct convolution's
      c.flops_saved += 2.0 * p.M * (double)(9 * w.Cin) * (w.CoutAlg ? w.CoutAlg : p.Cout) - 2.0 * (double)planes * (double)tiles * w.CinPad * w.Cout;
      return;
    }
  }
  // Split-K for few-row, long-K products (ArcFace's 25088 -> 512 embedding layer on a handful of faces; SAMP-Net's eight pattern layers,
  // K = 2592 .. 7524 -> 1024, at micro-batches above 32 rows: SAMP stage 3290 -> 3330 images/s): a single tile row would walk K
  // serially on Cout/64 CUs. K is cut into equal slices that run as ONE batched launch into [splits][M][Cout] partials, which
  // splitk_reduce sums in a fixed order and finishes with the epilogue.
  static const bool no_splitk = getenv("FE_NO_SPLITK") != nullptr;
  if (!no_splitk && w.KH == 1 && w.KW == 1 && o.sh == 1 && o.sw == 1 && o.ph == 0 && o.pw == 0 && !o.gate && p.variant == 0 && p.M <= 256 &&
      w.CinPad >= (getenv("FE_SPLITK_MIN") ? atoi(getenv("FE_SPLITK_MIN")) : 2048) && w.K == w.CinPad &&      // (A/B hook; 4096 until round 3) (size_t)((p.M + 127) / 128) * ((w.Cout + 63) / 64) < 64 &&
      (o.act == ACT_NONE || o.act == ACT_RELU || o.act == ACT_PRELU)) {
    int splits = 0;
    for (int s = 64; s >= 2; --s)
      if (w.CinPad % s == 0 && (w.CinPad / s) % 32 == 0 && w.CinPad / s >= 256) { splits = s; break; }
    const size_t part_bytes = (size_t)splits * p.M * w.Cout * sizeof(float);
    if (splits && c.arena.mark() + part_bytes + 1024 <= c.arena.capacity()) {
      const size_t mark = c.arena.mark();
      float* part = (float*)c.arena.alloc(part_bytes);
      const int Ks = w.CinPad / splits;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c.profile) {
        FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1));
        FE_HIP(hipEventRecord(e0, c.stream));
      }
      ConvParams g{};
      g.x = x.p; g.ldx = x.ld; g.w = w.w; g.ldw = w.Kp; g.y = part; g.ldy = w.Cout;
      g.M = p.M; g.K = Ks; g.Kp = Ks; g.Cin = Ks; g.Cout = w.Cout;
      g.N = 1; g.H = 1; g.W = g.M; g.Ho = 1; g.Wo = g.M; g.KH = g.KW = 1; g.sh = g.sw = g.dh = g.dw = 1;
      g.batch = splits; g.nb1 = 1; g.xs2 = Ks; g.ws2 = Ks; g.ys2 = (long long)p.M * w.Cout;
      launch_conv(g, c.stream);
      launch_splitk_reduce(part, splits, p.M, w.Cout, w.scale, w.shift, w.slope, o.act, o.res ? o.res->p : nullptr, o.res ? o.res->ld : 0,
                           o.res_after_act, y.p, y.ld, c.stream);
      if (c.profile) {
        FE_HIP(hipEventRecord(e1, c.stream));
        FE_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        FE_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        char nm[128];
        snprintf(nm, sizeof nm, "splitk%d 1x1 M=%d K=%d N=%d", splits, p.M, p.K, p.Cout);
        c.timings.push_back({nm, 2.0 * p.M * (double)w.Cin * (w.CoutAlg ? w.CoutAlg : p.Cout), 0.0, ms});
      }
      c.arena.rewind(mark);
      c.flops_accum += 2.0 * p.M * (double)w.Cin * (w.CoutAlg ? w.CoutAlg : p.Cout);
      return;
    }
  }
  // dedicated stem kernel (patch + weights resident in LDS) for 7x7/2 and 3x3/1|2 first layers; falls back otherwise
  static const bool no_stem = getenv("FE_NO_STEM") != nullptr;
  const bool stem = w.wstem && !no_stem && p.variant == 0 && o.sh == o.sw && o.ph == w.KH / 2 && o.pw == w.KW / 2 && o.dh == 1 && o.dw == 1 && !o.res &&
                    !o.gate && (o.act == ACT_NONE || o.act == ACT_RELU || o.act == ACT_PRELU);
  auto launch = [&]() {
    if (stem && launch_stem(x.p, x.ld, x.n, x.h, x.w, w.wstem, w.scale, w.shift, w.slope, w.Cout, w.KH, o.sh,
                            o.act == ACT_RELU ? 1 : (o.act == ACT_PRELU ? 2 : 0), y.p, y.ld, y.h, y.w, c.stream))
      return;
    launch_conv(p, c.stream);
  };
  if (c.profile) {
    hipEvent_t e0, e1;
    FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1));
    FE_HIP(hipEventRecord(e0, c.stream));
    launch();
    FE_HIP(hipEventRecord(e1, c.stream));
    FE_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FE_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    char nm[128];
    snprintf(nm, sizeof nm, "conv%dx%d s%d d%d M=%d K=%d N=%d", w.KH, w.KW, o.sh, o.dh, p.M, p.K, p.Cout);
    double bytes = 4.0 * ((double)x.pixels() * x.c + (double)y.pixels() * y.c * (o.res ? 2 : 1) + (double)w.Cout * w.K);
    c.timings.push_back({nm, 2.0 * p.M * (double)(w.KH * w.KW * w.Cin) * (w.CoutAlg ? w.CoutAlg : p.Cout), bytes, ms});
  } else {
    launch();
  }
  c.flops_accum += 2.0 * p.M * (double)(w.KH * w.KW * w.Cin) * (w.CoutAlg ? w.CoutAlg : p.Cout);
}

// ---- bf16 path ------------------------------------------------------------------------------------------------------
// Every dense layer goes to the bf16 MFMA kernel directly: no Winograd (F(4x4) needs more than 8 mantissa bits, and at the bf16
// matrix rate the 3x3 layers are bound by their HBM streams, which the transforms would multiply), no split-K.
template <class E>
static void conv_forward_half(Ctx& c, const ConvW& w, const TensorT<E>& x, const TensorT<E>& y, const ConvOptsT<E>& o) {
  if (!w.wh && w.w && PrecOf<E>::value != PREC_F32 && !o.res && !o.gate && x.ld == x.c && y.ld == y.c && x.c >= w.CinPad) {
    // A layer the 2-byte kernel has no form for (spatial kernels whose Cin rounded up to 8 is not a multiple of 16, or >= 64 taps):
    // the fp32 kernel between two conversion passes, like the 3-channel first layers (first_conv_half). No model of the hot path
    // has such a layer; a checkpoint variant that does still runs.
    const size_t mark = c.arena.mark();
    Tensor xf = c.arena.tensor(x.n, x.h, x.w, x.c), yf = c.arena.tensor(y.n, y.h, y.w, y.c);
    launch_convert(x.p, xf.p, x.numel(), c.stream);
    ConvOpts of; of.sh = o.sh; of.sw = o.sw; of.ph = o.ph; of.pw = o.pw; of.dh = o.dh; of.dw = o.dw; of.act = o.act;
    conv_forward(c, w, xf.slice(0, w.CinPad), yf, of);
    launch_convert(yf.p, y.p, y.numel(), c.stream);
    c.arena.rewind(mark);
    return;
  }
  FE_CHECK(w.wh && w.hprec == PrecOf<E>::value, "conv(2-byte): this layer has no weights of the activations' type (model committed under another "
           "precision, or a 3-channel first layer)");
  FE_CHECK(x.c == w.CinPadH, "conv(bf16): input channels %d != packed Cin %d", x.c, w.CinPadH);
  FE_CHECK(y.c == w.Cout && y.n == x.n, "conv(bf16): output view mismatch (c=%d Cout=%d)", y.c, w.Cout);
  FE_CHECK(y.p || o.y32, "conv(2-byte): no output tensor");
  FE_CHECK(y.h == conv_out_dim(x.h, w.KH, o.sh, o.ph, o.dh) && y.w == conv_out_dim(x.w, w.KW, o.sw, o.pw, o.dw),
           "conv(bf16): output dims %dx%d inconsistent with input %dx%d", y.h, y.w, x.h, x.w);
  FE_CHECK(y.pixels() < (1ull << 31), "conv: M too large");
  const double flops = 2.0 * (double)y.pixels() * (double)(w.KH * w.KW * w.Cin) * (w.CoutAlg ? w.CoutAlg : w.Cout);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c.profile) {
    FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1));
    FE_HIP(hipEventRecord(e0, c.stream));
  }
  ConvParamsT<E> p{};
  p.scale = w.scale; p.shift = w.shift; p.slope = w.slope;
  p.N = x.n; p.H = x.h; p.W = x.w; p.Cin = w.CinPadH; p.x = x.p; p.ldx = x.ld;
  if (w.wtap_h && o.act != ACT_PRELU && o.sh == 1 && o.sw == 1 && !o.res && !o.gate && !o.res32 && !o.y32 && y.h == x.h && y.w == x.w) {
    // narrow spatial conv (Cout <= 2): 1x1 conv to per-tap partials on the matrix cores + a gather-sum pass (see the fp32 path)
    const int T = w.KH * w.KW * w.Cout, T8 = (T + 7) & ~7;
    const size_t mark = c.arena.mark();
    TensorT<E> z = c.arena.tensor_t<E>(x.n, x.h, x.w, T8);
    p.w = (const E*)w.wtap_h; p.y = z.p; p.ldy = T8; p.scale = p.shift = p.slope = nullptr;
    p.Ho = x.h; p.Wo = x.w; p.Cout = T8; p.KH = p.KW = 1; p.sh = p.sw = p.dh = p.dw = 1;
    p.K = w.CinPadH; p.Kp = w.KpTH; p.M = (int)x.pixels(); p.cb = 32;
    launch_conv_bf16(p, c.stream);
    launch_tap_gather(z.p, T8, x.n, x.h, x.w, w.KH, w.KW, o.ph, o.pw, o.dh, o.dw, w.Cout, w.scale, w.shift, o.act, y.p, y.ld, y.h, y.w, c.stream);
    c.arena.rewind(mark);
  } else {
    FE_CHECK(o.act != ACT_PRELU || w.slope, "conv: PReLU without slopes");
    if (o.res) {
      FE_CHECK(o.res->c == y.c && o.res->pixels() == y.pixels(), "conv: residual shape mismatch");
      p.res = o.res->p; p.ldr = o.res->ld;
    }
    if (o.gate) {
      FE_CHECK(o.gate->pixels() == y.pixels() && (o.gate->c == 1 || o.gate->c == y.c), "conv: gate shape mismatch");
      p.gate = o.gate->p; p.ldg = o.gate->ld; p.gate_c1 = o.gate->c == 1;
    }
    if (o.res32) {
      FE_CHECK(!o.res && o.res32->c == y.c && o.res32->pixels() == y.pixels(), "conv: fp32 residual shape mismatch");
      p.res32 = o.res32->p; p.ldr32 = o.res32->ld;
    }
    if (o.y32) {
      FE_CHECK(o.y32->c == y.c && o.y32->pixels() == y.pixels(), "conv: fp32 output shape mismatch");
      p.y32 = o.y32->p; p.ldy32 = o.y32->ld;
    }
    p.w = (const E*)w.wh; p.y = y.p; p.ldy = y.ld;
    p.Ho = y.h; p.Wo = y.w; p.Cout = w.Cout;
    p.KH = w.KH; p.KW = w.KW; p.sh = o.sh; p.sw = o.sw; p.ph = o.ph; p.pw = o.pw; p.dh = o.dh; p.dw = o.dw;
    p.K = w.KH * w.KW * w.CinPadH; p.Kp = w.KpH; p.M = (int)y.pixels(); p.cb = w.cb;
    p.act = o.act; p.res_after_act = o.res_after_act;
    p.variant = c.force_variant;
    launch_conv_bf16(p, c.stream);
  }
  if (c.profile) {
    FE_HIP(hipEventRecord(e1, c.stream));
    FE_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FE_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    char nm[128];
    snprintf(nm, sizeof nm, "%s conv%dx%d s%d d%d M=%d K=%d N=%d", PrecOf<E>::value == PREC_F16 ? "f16" : "bf16", w.KH, w.KW, o.sh, o.dh, (int)y.pixels(), w.KH * w.KW * w.Cin, w.Cout);
    const double bytes = 2.0 * ((double)x.pixels() * x.c + (double)y.pixels() * y.c * (o.res ? 2 : 1) + (double)w.Cout * w.KpH);
    c.timings.push_back({nm, flops, bytes, ms});
  }
  c.flops_accum += flops;
  c.flops_half += flops;
}
void conv_forward(Ctx& c, const ConvW& w, const TensorH& x, const TensorH& y, const ConvOptsT<bf16>& o) { conv_forward_half<bf16>(c, w, x, y, o); }
void conv_forward(Ctx& c, const ConvW& w, const TensorF16& x, const TensorF16& y, const ConvOptsT<f16>& o) { conv_forward_half<f16>(c, w, x, y, o); }

template <class T>
TensorT<T> conv_new(Ctx& c, const ConvW& w, const TensorT<T>& x, const ConvOptsT<T>& o) {
  TensorT<T> y = c.arena.tensor_t<T>(x.n, conv_out_dim(x.h, w.KH, o.sh, o.ph, o.dh), conv_out_dim(x.w, w.KW, o.sw, o.pw, o.dw), w.Cout);
  conv_forward(c, w, x, y, o);
  return y;
}
template Tensor conv_new<float>(Ctx&, const ConvW&, const Tensor&, const ConvOpts&);
template TensorH conv_new<bf16>(Ctx&, const ConvW&, const TensorH&, const ConvOptsT<bf16>&);
template TensorF16 conv_new<f16>(Ctx&, const ConvW&, const TensorF16&, const ConvOptsT<f16>&);

template <>
Tensor first_conv<float>(Ctx& c, const ConvW& w, const Tensor& x, const ConvOpts& o) { return conv_new(c, w, x, o); }
template <class E>
static TensorT<E> first_conv_half(Ctx& c, const ConvW& w, const Tensor& x, const ConvOpts& o) {
  const int ho = conv_out_dim(x.h, w.KH, o.sh, o.ph, o.dh), wo = conv_out_dim(x.w, w.KW, o.sw, o.pw, o.dw);
  TensorT<E> y = c.arena.tensor_t<E>(x.n, ho, wo, w.Cout);
  static const bool no_stem = getenv("FE_NO_STEM") != nullptr;
  const bool stem = w.wstem && !no_stem && o.sh == o.sw && o.ph == w.KH / 2 && o.pw == w.KW / 2 && o.dh == 1 && o.dw == 1 && !o.res && !o.gate &&
                    (o.act == ACT_NONE || o.act == ACT_RELU || o.act == ACT_PRELU);
  if (stem && launch_stem(x.p, x.ld, x.n, x.h, x.w, w.wstem, w.scale, w.shift, w.slope, w.Cout, w.KH, o.sh,
                          o.act == ACT_RELU ? 1 : (o.act == ACT_PRELU ? 2 : 0), y.p, y.ld, ho, wo, c.stream)) {
    c.flops_accum += 2.0 * (double)y.pixels() * (double)(w.KH * w.KW * w.Cin) * w.Cout;
    if (w.KH == 7) c.flops_half += 2.0 * (double)y.pixels() * (double)(w.KH * w.KW * w.Cin) * w.Cout;      // the 7x7 stems run on the 2-byte matrix cores
    return y;
  }
  // shapes the stem kernel does not take: the fp32 kernel, then one conversion pass
  const size_t mark = c.arena.mark();
  Tensor yf = conv_new(c, w, x, o);
  launch_convert(yf.p, y.p, y.numel(), c.stream);
  (void)mark;   // yf stays allocated until the caller's next rewind: y was taken from the arena before it
  return y;
}
template <>
TensorH first_conv<bf16>(Ctx& c, const ConvW& w, const Tensor& x, const ConvOpts& o) { return first_conv_half<bf16>(c, w, x, o); }
template <>
TensorF16 first_conv<f16>(Ctx& c, const ConvW& w, const Tensor& x, const ConvOpts& o) { return first_conv_half<f16>(c, w, x, o); }

// ---------------------------------------------------------------------------------------------------
// ResNet
// ---------------------------------------------------------------------------------------------------
void build_resnet(ResNet& r, DeviceWeights& dw, const WeightStore& ws, const std::string& prefix,
                  bool bottleneck, const int blocks[4], bool seq_names) {
  r.bottleneck = bottleneck;
  auto nm = [&](const char* plain, int seq) { return prefix + (seq_names ? std::to_string(seq) : std::string(plain)); };
  r.stem = build_conv(dw, ws, nm("conv1", 0), nm("bn1", 1), false);
  r.layers.clear();
  for (int li = 0; li < 4; ++li) {
    std::vector<ResBlock> L;
    const std::string lp = seq_names ? prefix + std::to_string(4 + li) : prefix + "layer" + std::to_string(li + 1);
    for (int bi = 0; bi < blocks[li]; ++bi) {
      const std::string bp = lp + "." + std::to_string(bi);
      ResBlock b;
      b.bottleneck = bottleneck;
      b.stride = (bi == 0 && li > 0) ? 2 : 1;
      b.c1 = build_conv(dw, ws, bp + ".conv1", bp + ".bn1", false);
      b.c2 = build_conv(dw, ws, bp + ".conv2", bp + ".bn2", false);
      if (bottleneck) b.c3 = build_conv(dw, ws, bp + ".conv3", bp + ".bn3", false);
      if (bottleneck && dw.prec != PREC_F32 && b.stride == 1 && b.c2.Cin == 64 && b.c2.Cout == 64 && b.c2.KH == 3 && b.c2.KW == 3 && b.c3.Cin == 64 && b.c3.Cout == 256 &&
          b.c3.KH == 1 && !b.c2.slope && !b.c3.slope)
        build_c64_fragments(dw, ws.get(bp + ".conv2.weight").data.data(), ws.get(bp + ".conv3.weight").data.data(), &b.frag2, &b.frag3);
      b.has_down = ws.has(bp + ".downsample.0.weight");
      if (b.has_down) b.down = build_conv(dw, ws, bp + ".downsample.0", bp + ".downsample.1", false);
      L.push_back(b);
    }
    r.layers.push_back(std::move(L));
  }
}

template <class T>
TensorT<T> resnet_forward(Ctx& c, const ResNet& r, const Tensor& x, std::vector<TensorT<T>>* feats, bool res32, Tensor* last32) {
  ConvOpts so; so.sh = so.sw = 2; so.ph = so.pw = 3; so.act = ACT_RELU;
  TensorT<T> t = first_conv<T>(c, r.stem, x, so);
  if (feats) feats->push_back(t);
  TensorT<T> p = c.arena.tensor_t<T>(t.n, conv_out_dim(t.h, 3, 2, 1, 1), conv_out_dim(t.w, 3, 2, 1, 1), t.c);
  launch_maxpool(t, p, 3, 2, 1, c.stream);
  t = p;
  Tensor t32;      // fp32 skip stream (res32): the fp32 form of t
  if constexpr (sizeof(T) == 2) {
    if (res32) {
      t32 = c.arena.tensor(t.n, t.h, t.w, t.c);
      launch_convert(t.p, t32.p, t.numel(), c.stream);
    }
  } else {
    res32 = false;
  }
  for (size_t li = 0; li < r.layers.size(); ++li) {
    for (const ResBlock& b : r.layers[li]) {
      if constexpr (sizeof(T) == 2) {
        if (res32) {
          // identity in fp32: the stream itself, or the downsample branch written as fp32 only
          Tensor idt32 = t32;
          if (b.has_down) {
            idt32 = c.arena.tensor(t.n, conv_out_dim(t.h, 1, b.stride, 0, 1), conv_out_dim(t.w, 1, b.stride, 0, 1), b.down.Cout);
            TensorT<T> none = idt32.template retype<T>();
            ConvOptsT<T> d; d.sh = d.sw = b.stride; d.y32 = &idt32;
            conv_forward(c, b.down, t, none, d);
          }
          TensorT<T> bb;
          const ConvW* last;
          if (b.bottleneck) {
            ConvOptsT<T> o1; o1.act = ACT_RELU;
            TensorT<T> a = conv_new(c, b.c1, t, o1);
            ConvOptsT<T> o2; o2.sh = o2.sw = b.stride; o2.ph = o2.pw = 1; o2.act = ACT_RELU;
            bb = conv_new(c, b.c2, a, o2);
            last = &b.c3;
          } else {
            ConvOptsT<T> o1; o1.sh = o1.sw = b.stride; o1.ph = o1.pw = 1; o1.act = ACT_RELU;
            bb = conv_new(c, b.c1, t, o1);
            last = &b.c2;
          }
          // block output = relu(conv + identity): fp32 (the stream) and T (the next operand) from one epilogue
          ConvOptsT<T> o3; o3.act = ACT_RELU; o3.res32 = &idt32;
          if (!b.bottleneck) o3.ph = o3.pw = 1;
          TensorT<T> tn = c.arena.tensor_t<T>(bb.n, bb.h, bb.w, last->Cout);
          Tensor tn32 = c.arena.tensor(bb.n, bb.h, bb.w, last->Cout);
          o3.y32 = &tn32;
          conv_forward(c, *last, bb, tn, o3);
          t = tn; t32 = tn32;
          continue;
        }
      }
      TensorT<T> idt = t;
      if (b.has_down) {
        ConvOptsT<T> d; d.sh = d.sw = b.stride;
        idt = conv_new(c, b.down, t, d);
      }
      if (b.bottleneck) {
        // torchvision v1.5 / timm: stride sits on the 3x3 (conv2)
        ConvOptsT<T> o1; o1.act = ACT_RELU;
        TensorT<T> a = conv_new(c, b.c1, t, o1);
        if constexpr (sizeof(T) == 2) {
          if (b.frag2 && b.frag3 && a.ld == 64 && !getenv("FE_NO_FUSED_C64")) {      // (A/B hook, read per call)
            // conv2 + bn2 + relu + conv3 + bn3 + identity + relu in one launch: the 64-channel tensor between them stays in registers
            TensorT<T> tn = c.arena.tensor_t<T>(a.n, a.h, a.w, 256);
            const double px = (double)a.pixels(), flops = 2.0 * px * (576.0 * 64 + 64.0 * 256);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c.profile) { FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1)); FE_HIP(hipEventRecord(e0, c.stream)); }
            launch_conv3x3_c64(a, tn, &idt, b.frag2, b.frag3, b.c2.scale, b.c2.shift, b.c3.scale, b.c3.shift, ACT_RELU, c.stream);
            if (c.profile) {
              FE_HIP(hipEventRecord(e1, c.stream)); FE_HIP(hipEventSynchronize(e1));
              float ms = 0.f;
              FE_HIP(hipEventElapsedTime(&ms, e0, e1));
              (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
              char nm[128];
              snprintf(nm, sizeof nm, "%s fused conv3x3 64->64 + 1x1 64->256 + identity M=%d", PrecOf<T>::value == PREC_F16 ? "f16" : "bf16", (int)a.pixels());
              c.timings.push_back({nm, flops, 2.0 * px * (64 + 256 + 256), ms});
            }
            c.flops_accum += flops; c.flops_half += flops;
            t = tn;
            continue;
          }
        }
        ConvOptsT<T> o2; o2.sh = o2.sw = b.stride; o2.ph = o2.pw = 1; o2.act = ACT_RELU;
        TensorT<T> bb = conv_new(c, b.c2, a, o2);
        ConvOptsT<T> o3; o3.act = ACT_RELU; o3.res = &idt;
        t = conv_new(c, b.c3, bb, o3);
      } else {
        ConvOptsT<T> o1; o1.sh = o1.sw = b.stride; o1.ph = o1.pw = 1; o1.act = ACT_RELU;
        TensorT<T> a = conv_new(c, b.c1, t, o1);
        ConvOptsT<T> o2; o2.ph = o2.pw = 1; o2.act = ACT_RELU; o2.res = &idt;
        t = conv_new(c, b.c2, a, o2);
      }
    }
    if (feats) feats->push_back(t);
  }
  if (last32) *last32 = t32;
  return t;
}
template Tensor resnet_forward<float>(Ctx&, const ResNet&, const Tensor&, std::vector<Tensor>*, bool, Tensor*);
template TensorH resnet_forward<bf16>(Ctx&, const ResNet&, const Tensor&, std::vector<TensorH>*, bool, Tensor*);
template TensorF16 resnet_forward<f16>(Ctx&, const ResNet&, const Tensor&, std::vector<TensorF16>*, bool, Tensor*);

}  // namespace fe

// ---------------------------------------------------------------------------------------------------
// Linear / multi-head attention on token matrices
// ---------------------------------------------------------------------------------------------------
namespace fe {

void linear_forward(Ctx& c, const ConvW& w, const float* x, int ldx, int M, float* y, int ldy, int act,
                    const float* res, int ldr) {
  if (M <= 32 && !res && act != ACT_PRELU && c.force_variant == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)w.w) & 15) == 0) {
    // per-image vectors: stream the weight matrix once instead of idling 255 CUs behind one 128-row tile
    launch_gemm_skinny(x, ldx, w.w, w.Kp, w.scale, w.shift, y, ldy, M, w.Cout, w.K, act, c.stream);
    c.flops_accum += 2.0 * M * (double)w.Cin * (w.CoutAlg ? w.CoutAlg : w.Cout);
    return;
  }
  Tensor xt = mat_view(x, M, w.CinPad, ldx), yt = mat_view(y, M, w.Cout, ldy);
  ConvOpts o; o.act = act;
  Tensor rt;
  if (res) { rt = mat_view(res, M, w.Cout, ldr); o.res = &rt; }
  conv_forward(c, w, xt, yt, o);
}
void linear_forward_f32(Ctx& c, const ConvW& w, const float* x, int ldx, int M, float* y, int ldy, int act) {
  linear_forward(c, w, x, ldx, M, y, ldy, act);
}

template <class E>
static void linear_forward_half(Ctx& c, const ConvW& w, const E* x, int ldx, int M, E* y, int ldy, int act, const E* res, int ldr) {
  FE_CHECK(w.wh && w.hprec == PrecOf<E>::value, "linear(2-byte): layer has no weights of the activations' type");
  if (M <= 32 && !res && act != ACT_PRELU && ldx % 4 == 0 && ((uintptr_t)x & 7) == 0) {
    launch_gemm_skinny(x, ldx, (const E*)w.wh, w.KpH, w.scale, w.shift, y, ldy, M, w.Cout, w.CinPadH, act, c.stream);
    c.flops_accum += 2.0 * M * (double)w.Cin * (w.CoutAlg ? w.CoutAlg : w.Cout);
    return;
  }
  TensorT<E> xt = mat_view(x, M, w.CinPadH, ldx), yt = mat_view(y, M, w.Cout, ldy);
  ConvOptsT<E> o; o.act = act;
  TensorT<E> rt;
  if (res) { rt = mat_view(res, M, w.Cout, ldr); o.res = &rt; }
  conv_forward(c, w, xt, yt, o);
}
// 2-byte operand rows, fp32 residual rows (nullable), fp32 result rows: the projections that write the fp32 token stream
template <class E>
static void linear_forward_s32_half(Ctx& c, const ConvW& w, const E* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr) {
  FE_CHECK(w.wh && w.hprec == PrecOf<E>::value, "linear(2-byte): layer has no weights of the activations' type");
  TensorT<E> xt = mat_view(x, M, w.CinPadH, ldx), yt = mat_view((E*)nullptr, M, w.Cout, 0);
  Tensor y32 = mat_view(y, M, w.Cout, ldy), r32;
  ConvOptsT<E> o; o.act = act; o.y32 = &y32;
  if (res) { r32 = mat_view(res, M, w.Cout, ldr); o.res32 = &r32; }
  conv_forward(c, w, xt, yt, o);
}
void linear_forward_s32(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr) {
  linear_forward_s32_half<bf16>(c, w, x, ldx, M, y, ldy, act, res, ldr);
}
void linear_forward_s32(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr) {
  linear_forward_s32_half<f16>(c, w, x, ldx, M, y, ldy, act, res, ldr);
}
void linear_forward(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, bf16* y, int ldy, int act, const bf16* res, int ldr) {
  linear_forward_half<bf16>(c, w, x, ldx, M, y, ldy, act, res, ldr);
}
void linear_forward(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, f16* y, int ldy, int act, const f16* res, int ldr) {
  linear_forward_half<f16>(c, w, x, ldx, M, y, ldy, act, res, ldr);
}
// 2-byte activations, fp32 results: the weight matrix is streamed once per block of 32 rows (heads only: M is a batch size)
template <class E>
static void linear_forward_f32_half(Ctx& c, const ConvW& w, const E* x, int ldx, int M, float* y, int ldy, int act) {
  FE_CHECK(w.wh && w.hprec == PrecOf<E>::value && act != ACT_PRELU && ldx % 4 == 0 && ((uintptr_t)x & 7) == 0, "linear_f32(2-byte): unsupported layer");
  for (int m0 = 0; m0 < M; m0 += 32) {
    const int mb = std::min(32, M - m0);
    launch_gemm_skinny(x + (size_t)m0 * ldx, ldx, (const E*)w.wh, w.KpH, w.scale, w.shift, y + (size_t)m0 * ldy, ldy, mb, w.Cout, w.CinPadH, act, c.stream);
  }
  c.flops_accum += 2.0 * M * (double)w.Cin * (w.CoutAlg ? w.CoutAlg : w.Cout);
}
void linear_forward_f32(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float* y, int ldy, int act) { linear_forward_f32_half<bf16>(c, w, x, ldx, M, y, ldy, act); }
void linear_forward_f32(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, float* y, int ldy, int act) { linear_forward_f32_half<f16>(c, w, x, ldx, M, y, ldy, act); }
void linear_forward_xf32(Ctx& c, const ConvW& w, int prec, const float* x, int ldx, int M, float* y, int ldy, int act) {
  FE_CHECK(w.wh && w.hprec == prec && act != ACT_PRELU && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "linear_xf32: unsupported layer");
  for (int m0 = 0; m0 < M; m0 += 32) {
    const int mb = std::min(32, M - m0);
    if (prec == PREC_F16) launch_gemm_skinny(x + (size_t)m0 * ldx, ldx, (const f16*)w.wh, w.KpH, w.scale, w.shift, y + (size_t)m0 * ldy, ldy, mb, w.Cout, w.CinPadH, act, c.stream);
    else launch_gemm_skinny(x + (size_t)m0 * ldx, ldx, (const bf16*)w.wh, w.KpH, w.scale, w.shift, y + (size_t)m0 * ldy, ldy, mb, w.Cout, w.CinPadH, act, c.stream);
  }
  c.flops_accum += 2.0 * M * (double)w.Cin * (w.CoutAlg ? w.CoutAlg : w.Cout);
}

MHAW build_mha(DeviceWeights& dw, const WeightStore& ws, const std::string& prefix, int heads) {
  MHAW m;
  const HostTensor& W = ws.get(prefix + ".in_proj_weight");
  const HostTensor& Bv = ws.get(prefix + ".in_proj_bias");
  const int d = (int)W.shape[1];
  FE_CHECK(W.shape[0] == 3 * d && d % heads == 0 && d % 32 == 0, "mha %s: bad in_proj shape", prefix.c_str());
  m.d = d; m.heads = heads;
  m.q = build_linear_rows(dw, W, nullptr, 0, d);
  m.k = build_linear_rows(dw, W, &Bv, d, d);
  // q = (x Wq^T + bq) * 1/sqrt(hd)  (torch.nn.functional.multi_head_attention_forward scales q after the bias)
  const float sc = 1.0f / std::sqrt((float)(d / heads));
  std::vector<float> qs(d, sc), qb(d);
  for (int i = 0; i < d; ++i) qb[i] = Bv.data[i] * sc;
  m.q.scale = dw.upload(qs);
  m.q.shift = dw.upload(qb);
  std::vector<float> wv(W.data.begin() + (size_t)2 * d * d, W.data.begin() + (size_t)3 * d * d);
  std::vector<float> bv(Bv.data.begin() + 2 * d, Bv.data.begin() + 3 * d);
  m.wv = dw.upload(wv);
  m.bv = dw.upload(bv);
  if (dw.prec != PREC_F32) {
    std::vector<uint16_t> wvh(wv.size());
    for (size_t i = 0; i < wv.size(); ++i) wvh[i] = f32_to_half_bits(wv[i], dw.prec);
    m.wv_h = dw.upload_raw(wvh.data(), wvh.size() * sizeof(uint16_t));
  }
  m.out = build_linear(dw, ws, prefix + ".out_proj", true);
  return m;
}

static void launch_gemm(const ConvParams& p, hipStream_t s) { launch_conv(p, s); }
static void launch_gemm(const ConvParamsH& p, hipStream_t s) { launch_conv_bf16(p, s); }
static void launch_gemm(const ConvParamsT<f16>& p, hipStream_t s) { launch_conv_bf16(p, s); }

template <class T>
static void raw_gemm(Ctx& c, ConvParamsT<T>& p, double flops) {
  constexpr int kalign = sizeof(T) == 2 ? CONV_KALIGN_H : CONV_KALIGN;
  constexpr int valign = 16 / sizeof(T);
  p.N = 1; p.H = 1; p.W = p.M; p.Ho = 1; p.Wo = p.M;
  p.KH = p.KW = 1; p.sh = p.sw = p.dh = p.dw = 1; p.ph = p.pw = 0;
  p.Kp = (p.K + kalign - 1) / kalign * kalign;
  p.Cin = p.K;
  FE_CHECK(p.xs1 % valign == 0 && p.xs2 % valign == 0 && p.ws1 % valign == 0 && p.ws2 % valign == 0, "raw_gemm: batch strides must keep 16-B alignment");
  // K is rounded up to whole K-steps (Kp); the A side zero-fills its chunks past K, but a B operand that is a live activation
  // matrix (ldw != Kp) would be read past K into neighbouring rows - 0 x NaN from stale arena bytes is NaN
  FE_CHECK(p.ldw == 0 || p.ldw == p.Kp || p.K % kalign == 0, "raw_gemm: K = %d must be a multiple of %d when the B operand is an activation matrix", p.K, kalign);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c.profile) {
    FE_HIP(hipEventCreate(&e0)); FE_HIP(hipEventCreate(&e1));
    FE_HIP(hipEventRecord(e0, c.stream));
  }
  launch_gemm(p, c.stream);
  if (c.profile) {
    FE_HIP(hipEventRecord(e1, c.stream));
    FE_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FE_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    char nm[128];
    snprintf(nm, sizeof nm, "%sbgemm x%d M=%d K=%d N=%d", PrecOf<T>::value == PREC_F16 ? "f16 " : (PrecOf<T>::value == PREC_BF16 ? "bf16 " : ""), p.batch > 1 ? p.batch : 1, p.M, p.K, p.Cout);
    c.timings.push_back({nm, flops, 0.0, ms});
  }
  c.flops_accum += flops;
  if (sizeof(T) == 2) c.flops_half += flops;
}

static const float* mha_wv(const MHAW& m, const float*) { return m.wv; }
static const bf16* mha_wv(const MHAW& m, const bf16*) {
  FE_CHECK(m.wv_h && m.q.hprec == PREC_BF16, "mha(bf16): model was not committed under bf16 precision");
  return (const bf16*)m.wv_h;
}
static const f16* mha_wv(const MHAW& m, const f16*) {
  FE_CHECK(m.wv_h && m.q.hprec == PREC_F16, "mha(f16): model was not committed under f16 precision");
  return (const f16*)m.wv_h;
}

// y (+= res) in the stream type RT of the caller: the plain layer when RT is the operand type, the fp32-stream form otherwise
void linear_forward_res(Ctx& c, const ConvW& w, const float* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr) { linear_forward(c, w, x, ldx, M, y, ldy, act, res, ldr); }
void linear_forward_res(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, bf16* y, int ldy, int act, const bf16* res, int ldr) { linear_forward(c, w, x, ldx, M, y, ldy, act, res, ldr); }
void linear_forward_res(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, f16* y, int ldy, int act, const f16* res, int ldr) { linear_forward(c, w, x, ldx, M, y, ldy, act, res, ldr); }
void linear_forward_res(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr) { linear_forward_s32(c, w, x, ldx, M, y, ldy, act, res, ldr); }
void linear_forward_res(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr) { linear_forward_s32(c, w, x, ldx, M, y, ldy, act, res, ldr); }

template <class T, class RT>
void mha_forward(Ctx& c, const MHAW& m, const T* q_in, int ldq, const T* kv_in, int ldkv, int B, int Lq, int Lk,
                 const RT* res, int ldr, RT* y, int ldy, bool causal) {
  constexpr bool half = sizeof(T) == 2;
  const int d = m.d, H = m.heads, hd = d / H;
  const int Lp = (Lk + 31) / 32 * 32;  // padded key count: row stride of the score matrix and of V^T
  const size_t mark = c.arena.mark();
  T* Q = c.arena.array<T>((size_t)B * Lq * d);
  T* K = c.arena.array<T>((size_t)B * Lk * d);
  T* Vt = c.arena.array<T>((size_t)B * d * Lp);
  static const bool no_flash = getenv("FE_NO_FLASH") != nullptr;
  const bool flash = (hd == 64) && (!no_flash || half);
  FE_CHECK(flash || !causal, "causal attention needs the fused kernel (head_dim 64)");
  FE_CHECK(flash || !half, "attention(bf16) needs head_dim 64");
  T* S = flash ? nullptr : c.arena.array<T>((size_t)B * H * Lq * Lp);
  T* O = c.arena.array<T>((size_t)B * Lq * d);
  linear_forward(c, m.q, q_in, ldq, B * Lq, Q, d, ACT_NONE);
  linear_forward(c, m.k, kv_in, ldkv, B * Lk, K, d, ACT_NONE);
  if (Lp != Lk) FE_HIP(hipMemsetAsync(Vt, 0, (size_t)B * d * Lp * sizeof(T), c.stream));
  {  // V^T[b] [d][Lk] = Wv [d][d] . X_b^T : the token matrix plays the weight operand
    ConvParamsT<T> p{};
    p.x = mha_wv(m, q_in); p.ldx = d; p.w = kv_in; p.ldw = ldkv; p.y = Vt; p.ldy = Lp;
    p.M = d; p.K = d; p.Cout = Lk;
    p.batch = B; p.nb1 = 1; p.ws2 = (long long)Lk * ldkv; p.ys2 = (long long)d * Lp;
    p.pad_store = half ? 1 : 0;      // bf16 stores move 8 columns: the ragged last group lands in the zero padding of V^T
    raw_gemm(c, p, 2.0 * B * d * (double)d * Lk);
  }
  if (flash) {
    // fused QK^T -> online softmax -> PV (kernels_attn.hip / kernels_attn_bf16.hip); scores never touch HBM
    launch_attention(Q, d, K, d, Vt, Lp, m.bv, O, d, B, H, Lq, Lk, d, causal ? 1 : 0, c.stream);
    c.flops_accum += 4.0 * B * H * (double)Lq * Lk * hd;
    if (half) c.flops_half += 4.0 * B * H * (double)Lq * Lk * hd;
  } else if constexpr (!half) {
    {  // S[b,h] [Lq][Lk] = Q_bh K_bh^T
      ConvParams p{};
      p.x = Q; p.ldx = d; p.w = K; p.ldw = d; p.y = S; p.ldy = Lp;
      p.M = Lq; p.K = hd; p.Cout = Lk;
      p.batch = B * H; p.nb1 = H;
      p.xs1 = hd; p.xs2 = (long long)Lq * d; p.ws1 = hd; p.ws2 = (long long)Lk * d;
      p.ys1 = (long long)Lq * Lp; p.ys2 = (long long)H * Lq * Lp;
      raw_gemm(c, p, 2.0 * B * H * (double)Lq * Lk * hd);
    }
    launch_softmax_rows_pad(S, Lp, B * H * Lq, Lk, c.stream);
    {  // O[b][:, h*hd:(h+1)*hd] = P_bh V_bh + bv_h
      ConvParams p{};
      p.x = S; p.ldx = Lp; p.w = Vt; p.ldw = Lp; p.y = O; p.ldy = d;
      p.shift = m.bv; p.hs1 = hd;
      p.M = Lq; p.K = Lp; p.Cout = hd;
      p.batch = B * H; p.nb1 = H;
      p.xs1 = (long long)Lq * Lp; p.xs2 = (long long)H * Lq * Lp;
      p.ws1 = (long long)hd * Lp; p.ws2 = (long long)d * Lp;
      p.ys1 = hd; p.ys2 = (long long)Lq * d;
      raw_gemm(c, p, 2.0 * B * H * (double)Lq * Lk * hd);
    }
  }
  linear_forward_res(c, m.out, (const T*)O, d, B * Lq, y, ldy, ACT_NONE, res, ldr);
  c.arena.rewind(mark);
}
template void mha_forward<float, float>(Ctx&, const MHAW&, const float*, int, const float*, int, int, int, int, const float*, int, float*, int, bool);
template void mha_forward<bf16, bf16>(Ctx&, const MHAW&, const bf16*, int, const bf16*, int, int, int, int, const bf16*, int, bf16*, int, bool);
template void mha_forward<f16, f16>(Ctx&, const MHAW&, const f16*, int, const f16*, int, int, int, int, const f16*, int, f16*, int, bool);
template void mha_forward<bf16, float>(Ctx&, const MHAW&, const bf16*, int, const bf16*, int, int, int, int, const float*, int, float*, int, bool);
template void mha_forward<f16, float>(Ctx&, const MHAW&, const f16*, int, const f16*, int, int, int, int, const float*, int, float*, int, bool);

}  // namespace fe

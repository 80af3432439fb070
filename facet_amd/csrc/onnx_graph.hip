// ONNX-subset graph runtime (see onnx_graph.h). Two parts:
//   1. a protobuf wire-format reader for onnx.ModelProto (field numbers from the public onnx.proto3 schema),
//   2. an interpreter mapping nodes onto this library's kernels. Feature maps live in HBM as NHWC with the channel count
//      padded (zeros) to a multiple of 16 so every convolution takes the LDS-DMA implicit-GEMM path; Conv/Gemm followed by
//      BatchNormalization, an activation and a residual Add run as ONE launch (epilogue fusion); Transpose(0,2,3,1) +
//      Reshape tails of detection heads are free because NHWC already is that order.
// Replaces, for the reference, onnxruntime under insightface.app.FaceAnalysis (analyzers/face.py:30-38,99).
#include "onnx_graph.h"
#include <cmath>

namespace fe {

using onnx::Node;
using onnx::TensorData;

// ---------------------------------------------------------------------------------------------------------------------
static bool is_act_op(const std::string& op) { return op == "Relu" || op == "PRelu" || op == "Sigmoid" || op == "LeakyRelu"; }

void Graph::load(const uint8_t* data, size_t len) {
  m_ = onnx::Model();
  dw_.release();
  onnx::parse_model(data, len, m_);
  FE_CHECK(m_.inputs.size() == 1, "graph: %zu runtime inputs; exactly one image input is supported", m_.inputs.size());
  const int nn = (int)m_.nodes.size();
  cache_.assign(nn, NodeCache());
  uses_.clear(); producer_.clear();
  for (int k = 0; k < nn; ++k) {
    for (auto& s : m_.nodes[k].in) if (!s.empty()) uses_[s]++;
    for (auto& s : m_.nodes[k].out) producer_[s] = k;
  }
  for (auto& o : m_.outputs) uses_[o.name]++;
  has_sub_ = has_mul_ = false;
  for (int k = 0; k < nn && k < 8; ++k) {
    const std::string& nm = m_.nodes[k].name;
    if (nm.rfind("Sub", 0) == 0 || nm.rfind("_minus", 0) == 0) has_sub_ = true;
    if (nm.rfind("Mul", 0) == 0 || nm.rfind("_mul", 0) == 0) has_mul_ = true;
  }
  plan_fusion();
}

bool Graph::is_const(const std::string& name) const {
  if (m_.init.count(name)) return true;
  auto it = producer_.find(name);
  return it != producer_.end() && m_.nodes[it->second].op == "Constant";
}

// Chains Conv|Gemm|MatMul(const) -> [BatchNormalization] -> [act] -> [Add other] -> [Relu] whose intermediates have a single
// consumer become one group executed where its LAST node stands (so the residual operand is already computed).
void Graph::plan_fusion() {
  const int nn = (int)m_.nodes.size();
  group_end_.assign(nn, Group());
  absorbed_.assign(nn, 0);
  std::map<std::string, int> sole;   // value -> its only consumer node
  for (int k = 0; k < nn; ++k)
    for (auto& s : m_.nodes[k].in)
      if (!s.empty() && uses_[s] == 1) sole[s] = k;
  auto next_of = [&](int k) -> int {
    const Node& n = m_.nodes[k];
    if (n.out.size() != 1) return -1;
    auto it = sole.find(n.out[0]);
    return (it == sole.end() || absorbed_[it->second]) ? -1 : it->second;   // a node joins at most one group
  };
  for (int k = 0; k < nn; ++k) {
    const Node& n = m_.nodes[k];
    const bool conv = n.op == "Conv" && n.in.size() >= 2 && is_const(n.in[1]);
    const bool gemm = (n.op == "Gemm" || n.op == "MatMul") && n.in.size() >= 2 && is_const(n.in[1]);
    const bool bn = n.op == "BatchNormalization";
    if (!(conv || gemm || bn) || absorbed_[k]) continue;
    Group g;
    g.main = k;
    int cur = k, nx = next_of(cur);
    if (!bn && nx >= 0 && m_.nodes[nx].op == "BatchNormalization" && m_.nodes[nx].in[0] == m_.nodes[cur].out[0]) {
      g.bn = nx; cur = nx; nx = next_of(cur);
    }
    if (nx >= 0 && is_act_op(m_.nodes[nx].op) && m_.nodes[nx].in[0] == m_.nodes[cur].out[0] &&
        (m_.nodes[nx].op != "PRelu" || is_const(m_.nodes[nx].in[1]))) {
      g.act = nx; cur = nx; nx = next_of(cur);
    }
    if (!bn && nx >= 0 && m_.nodes[nx].op == "Add" && m_.nodes[nx].in.size() == 2) {
      const Node& a = m_.nodes[nx];
      const std::string& mine = m_.nodes[cur].out[0];
      const std::string& other = a.in[0] == mine ? a.in[1] : a.in[0];
      if (other != mine && !is_const(other)) {
        g.add = nx; cur = nx; nx = next_of(cur);
        if (g.act < 0 && nx >= 0 && m_.nodes[nx].op == "Relu") { g.act2 = nx; cur = nx; }
      }
    }
    for (int j : {g.main, g.bn, g.act, g.add, g.act2})
      if (j >= 0) absorbed_[j] = 1;
    group_end_[cur] = g;
  }
}

void Graph::set_host(const std::string& name, TensorData&& t) {
  temps_.push_back(std::move(t));
  Val v;
  v.kind = Val::HOST;
  v.host = &temps_.back();
  vals_[name] = v;
}

const Graph::Val& Graph::get(const std::string& name) {
  auto it = vals_.find(name);
  if (it != vals_.end()) return it->second;
  auto ii = m_.init.find(name);
  FE_CHECK(ii != m_.init.end(), "graph: value '%s' is used before it is produced", name.c_str());
  Val v;
  v.kind = Val::HOST;
  v.host = &ii->second;
  return vals_[name] = v;
}

const TensorData* Graph::host_of(const std::string& name) {
  if (name.empty()) return nullptr;
  const Val& v = get(name);
  return v.kind == Val::HOST ? v.host : nullptr;
}

// per-channel constant ([C], [C,1,1], [1,C,1,1] or a scalar) -> padded device-ready vector
void Graph::channel_vector(const TensorData& t, int C, int Cp, float padv, std::vector<float>& out) const {
  FE_CHECK(!t.is_int() || t.numel() >= 1, "graph: bad channel constant");
  const size_t n = t.numel();
  FE_CHECK(n == (size_t)C || n == 1, "graph: per-channel constant '%s' has %zu values for %d channels", t.name.c_str(), n, C);
  out.assign(Cp, padv);
  for (int k = 0; k < C; ++k) out[k] = (float)t.at(n == 1 ? 0 : k);
}

Graph::Val Graph::to_plain(Ctx& c, const Val& v) {
  if (v.kind == Val::PLAIN) return v;
  FE_CHECK(v.kind == Val::IMG, "graph: cannot view a host constant as a device tensor");
  Val o;
  o.kind = Val::PLAIN;
  const Tensor& t = v.t;
  if (v.rank == 2 || (t.h == 1 && t.w == 1)) {
    o.dims = v.rank == 2 ? std::vector<int64_t>{t.n, v.lc} : (v.flat ? std::vector<int64_t>{t.n, v.lc} : std::vector<int64_t>{t.n, v.lc, 1, 1});
    if (v.lc == t.c && t.ld == t.c) { o.p = t.p; return o; }
    o.p = (float*)c.arena.alloc(((size_t)t.n * v.lc + 4) * sizeof(float));
    const long long d[6] = {1, 1, 1, 1, t.n, v.lc}, s[6] = {0, 0, 0, 0, t.ld, 1};
    launch_gather_strided(t.p, o.p, d, s, c.stream);
    return o;
  }
  o.p = (float*)c.arena.alloc(((size_t)t.n * v.lc * t.h * t.w + 4) * sizeof(float));
  launch_nhwc_to_nchw(t.p, t.ld, o.p, t.n, v.lc, t.h, t.w, c.stream);
  o.dims = v.flat ? std::vector<int64_t>{t.n, (int64_t)v.lc * t.h * t.w} : std::vector<int64_t>{t.n, v.lc, t.h, t.w};
  return o;
}

Graph::Val Graph::to_img(Ctx& c, const Val& v) {
  if (v.kind == Val::IMG) {
    if (!v.flat || (v.t.h == 1 && v.t.w == 1)) { Val o = v; if (v.flat) { o.flat = false; o.rank = 2; } return o; }
    return to_img(c, to_plain(c, v));
  }
  FE_CHECK(v.kind == Val::PLAIN, "graph: cannot view a host constant as a feature map");
  FE_CHECK(v.dims.size() == 4 || v.dims.size() == 2, "graph: rank-%zu tensor cannot become a feature map", v.dims.size());
  Val o;
  o.kind = Val::IMG;
  const int n = (int)v.dims[0], ch = (int)v.dims[1];
  const int h = v.dims.size() == 4 ? (int)v.dims[2] : 1, w = v.dims.size() == 4 ? (int)v.dims[3] : 1;
  o.lc = ch;
  o.rank = (int)v.dims.size();
  o.t = c.arena.tensor(n, h, w, pad_channels(ch));
  launch_nchw_to_nhwc(v.p, o.t.p, n, ch, h, w, o.t.c, c.stream);
  return o;
}

// ---------------------------------------------------------------------------------------------------------------------
static int act_code(const Node& n) {
  if (n.op == "Relu") return ACT_RELU;
  if (n.op == "Sigmoid") return ACT_SIGMOID;
  return ACT_PRELU;   // PRelu, LeakyRelu (constant slope vector)
}

void Graph::exec_group(Ctx& c, int idx) {
  const Group g = group_end_[idx];
  const Node& n = m_.nodes[g.main];
  NodeCache& nc = cache_[g.main];
  const int last = idx;
  const std::string& out_name = m_.nodes[last].out[0];

  // ---- standalone BatchNormalization (+act) --------------------------------------------------------------------
  if (n.op == "BatchNormalization") {
    Val x = to_img(c, get(n.in[0]));
    const int C = x.lc, Cp = x.t.c;
    if (!nc.built || nc.key != Cp) {
      const TensorData *ga = host_of(n.in[1]), *be = host_of(n.in[2]), *mu = host_of(n.in[3]), *va = host_of(n.in[4]);
      FE_CHECK(ga && be && mu && va, "graph: BatchNormalization '%s' needs constant parameters", n.name.c_str());
      const float eps = n.getf("epsilon", 1e-5f);
      std::vector<float> sc(Cp, 0.f), sh(Cp, 0.f);
      FE_CHECK((int)ga->numel() == C, "graph: BatchNormalization '%s' has %zu channels, input has %d", n.name.c_str(), ga->numel(), C);
      for (int k = 0; k < C; ++k) {
        const float inv = 1.0f / std::sqrt(va->f[k] + eps);
        sc[k] = ga->f[k] * inv;
        sh[k] = be->f[k] - mu->f[k] * sc[k];
      }
      nc.scale = dw_.upload(sc); nc.shift = dw_.upload(sh);
      if (g.act >= 0 && act_code(m_.nodes[g.act]) == ACT_PRELU) {
        std::vector<float> sl;
        const Node& a = m_.nodes[g.act];
        if (a.op == "LeakyRelu") sl.assign(Cp, a.getf("alpha", 0.01f));
        else channel_vector(*host_of(a.in[1]), C, Cp, 0.f, sl);
        nc.slope = dw_.upload(sl);
      }
      nc.built = true; nc.key = Cp;
    }
    Val y = x;
    y.t = c.arena.tensor(x.t.n, x.t.h, x.t.w, Cp);
    launch_affine_act(x.t, y.t, nc.scale, nc.shift, g.act >= 0 ? act_code(m_.nodes[g.act]) : ACT_NONE, nc.slope, c.stream);
    vals_[out_name] = y;
    return;
  }

  // ---- Conv / Gemm / MatMul with epilogue fusion ------------------------------------------------------------------
  const bool is_conv = n.op == "Conv";
  Val xin = get(n.in[0]);
  FE_CHECK(xin.kind != Val::HOST, "graph: %s '%s' on a constant input is not supported", n.op.c_str(), n.name.c_str());
  Val x;
  long long key;
  if (is_conv) { x = to_img(c, xin); key = x.t.c; }
  else if (xin.kind == Val::IMG && xin.flat && !(xin.t.h == 1 && xin.t.w == 1)) {
    x = xin;   // Flatten of a feature map: consumed in place, the weight columns are permuted to NHWC order instead
    FE_CHECK(x.t.ld == x.t.c, "graph: flattened input must be dense");
    key = ((long long)x.t.h << 40) | ((long long)x.t.w << 20) | x.t.c;
  } else { x = to_img(c, xin); FE_CHECK(x.t.h == 1 && x.t.w == 1, "graph: %s '%s' expects a matrix input", n.op.c_str(), n.name.c_str()); key = x.t.c; }

  const TensorData* W = host_of(n.in[1]);
  FE_CHECK(W && !W->is_int(), "graph: %s '%s' needs a constant float weight", n.op.c_str(), n.name.c_str());
  const TensorData* B = n.in.size() > 2 ? host_of(n.in[2]) : nullptr;
  int Cout, Cin_l, KH = 1, KW = 1, group = 1;
  std::vector<int64_t> strides{1, 1}, pads{0, 0, 0, 0}, dil{1, 1};
  bool transB = false;
  if (is_conv) {
    FE_CHECK(W->dims.size() == 4, "graph: Conv '%s' weight rank %zu (only 2-D convolutions)", n.name.c_str(), W->dims.size());
    Cout = (int)W->dims[0]; KH = (int)W->dims[2]; KW = (int)W->dims[3];
    group = (int)n.geti("group", 1);
    Cin_l = (int)W->dims[1] * group;
    if (n.has("strides")) strides = n.getints("strides");
    if (n.has("pads")) pads = n.getints("pads");
    if (n.has("dilations")) dil = n.getints("dilations");
    const std::string ap = n.gets("auto_pad", "NOTSET");
    FE_CHECK(ap == "NOTSET" || ap == "VALID", "graph: Conv '%s' auto_pad=%s is not supported", n.name.c_str(), ap.c_str());
    FE_CHECK(pads.size() == 4 && pads[0] == pads[2] && pads[1] == pads[3], "graph: Conv '%s' has asymmetric padding", n.name.c_str());
    FE_CHECK(Cin_l == x.lc, "graph: Conv '%s' expects %d input channels, got %d", n.name.c_str(), Cin_l, x.lc);
    FE_CHECK(group == 1 || (group == Cin_l && Cout == Cin_l && W->dims[1] == 1),
             "graph: Conv '%s' group=%d: only dense and depthwise convolutions are supported", n.name.c_str(), group);
  } else {
    FE_CHECK(W->dims.size() == 2, "graph: %s '%s' weight rank", n.op.c_str(), n.name.c_str());
    transB = n.op == "Gemm" && n.geti("transB", 0) != 0;
    FE_CHECK(n.op != "Gemm" || (n.geti("transA", 0) == 0 && n.getf("alpha", 1.f) == 1.f && n.getf("beta", 1.f) == 1.f),
             "graph: Gemm '%s' with transA/alpha/beta is not supported", n.name.c_str());
    Cout = (int)(transB ? W->dims[0] : W->dims[1]);
    Cin_l = (int)(transB ? W->dims[1] : W->dims[0]);
    const int have = x.flat ? x.lc * x.t.h * x.t.w : x.lc;
    FE_CHECK(Cin_l == have, "graph: %s '%s' expects %d input features, got %d", n.op.c_str(), n.name.c_str(), Cin_l, have);
  }
  const int CoutP = pad_channels(Cout);
  const bool depthwise = group > 1;

  if (!nc.built || nc.key != key) {
    // epilogue vectors: y = act((acc + bias) * bn_scale + bn_shift)
    std::vector<float> sc, sh(CoutP, 0.f);
    if (B) { FE_CHECK((int)B->numel() == Cout, "graph: bias of '%s' has %zu values", n.name.c_str(), B->numel()); for (int k = 0; k < Cout; ++k) sh[k] = B->f[k]; }
    if (g.bn >= 0) {
      const Node& b = m_.nodes[g.bn];
      const TensorData *ga = host_of(b.in[1]), *be = host_of(b.in[2]), *mu = host_of(b.in[3]), *va = host_of(b.in[4]);
      FE_CHECK(ga && be && mu && va && (int)ga->numel() == Cout, "graph: BatchNormalization '%s' parameters", b.name.c_str());
      const float eps = b.getf("epsilon", 1e-5f);
      sc.assign(CoutP, 1.f);
      for (int k = 0; k < Cout; ++k) {
        const float inv = 1.0f / std::sqrt(va->f[k] + eps);
        sc[k] = ga->f[k] * inv;
        sh[k] = sh[k] * sc[k] + (be->f[k] - mu->f[k] * sc[k]);
      }
    }
    const bool any_shift = B || g.bn >= 0;
    if (depthwise) {
      const int Cp = x.t.c;
      std::vector<float> wt((size_t)KH * KW * Cp, 0.f);
      for (int ch = 0; ch < Cout; ++ch)
        for (int t = 0; t < KH * KW; ++t) wt[(size_t)t * Cp + ch] = W->f[(size_t)ch * KH * KW + t];
      nc.dwt = dw_.upload(wt);
      nc.scale = sc.empty() ? nullptr : dw_.upload(sc);
      nc.shift = any_shift ? dw_.upload(sh) : nullptr;
    } else {
      HostTensor hw;
      if (is_conv) {
        const int CinP = x.t.c;
        hw.shape = {CoutP, CinP, KH, KW};
        hw.data.assign((size_t)CoutP * CinP * KH * KW, 0.f);
        const size_t taps = (size_t)KH * KW;
        for (int co = 0; co < Cout; ++co)
          for (int ci = 0; ci < Cin_l; ++ci)
            memcpy(&hw.data[((size_t)co * CinP + ci) * taps], &W->f[((size_t)co * Cin_l + ci) * taps], taps * sizeof(float));
      } else {
        const int HWn = x.flat ? x.t.h * x.t.w : 1, Cp = x.t.c, Cl = x.lc;
        const int Kphys = HWn * Cp;
        hw.shape = {CoutP, Kphys};
        hw.data.assign((size_t)CoutP * Kphys, 0.f);
        for (int co = 0; co < Cout; ++co)
          for (int ch = 0; ch < Cl; ++ch)
            for (int s = 0; s < HWn; ++s) {
              const size_t kl = (size_t)ch * HWn + s;   // logical feature index (NCHW flatten)
              const float v = transB ? W->f[(size_t)co * Cin_l + kl] : W->f[kl * Cout + co];
              hw.data[(size_t)co * Kphys + (size_t)s * Cp + ch] = v;
            }
      }
      nc.cw = pack_conv(dw_, hw, sc.empty() ? nullptr : &sc, any_shift ? &sh : nullptr);
      nc.cw.Cin = Cin_l;          // algorithmic (unpadded) sizes for the FLOP counter
      nc.cw.CoutAlg = Cout;
    }
    if (g.act >= 0 && act_code(m_.nodes[g.act]) == ACT_PRELU) {
      std::vector<float> sl;
      const Node& a = m_.nodes[g.act];
      if (a.op == "LeakyRelu") { sl.assign(CoutP, 0.f); for (int k = 0; k < Cout; ++k) sl[k] = a.getf("alpha", 0.01f); }
      else channel_vector(*host_of(a.in[1]), Cout, CoutP, 0.f, sl);
      nc.slope = dw_.upload(sl);
      nc.cw.slope = nc.slope;
    }
    nc.built = true; nc.key = key;
  }

  int act = ACT_NONE, res_after = 0;
  if (g.act >= 0) { act = act_code(m_.nodes[g.act]); res_after = 1; }
  if (g.act2 >= 0) act = ACT_RELU;
  Val res;
  if (g.add >= 0) {
    const Node& a = m_.nodes[g.add];
    const std::string& prev = m_.nodes[g.act >= 0 ? g.act : (g.bn >= 0 ? g.bn : g.main)].out[0];
    res = to_img(c, get(a.in[0] == prev ? a.in[1] : a.in[0]));
  }

  Val y;
  y.kind = Val::IMG;
  y.lc = Cout;
  if (is_conv) {
    ConvOpts o;
    o.sh = (int)strides[0]; o.sw = (int)strides[1]; o.ph = (int)pads[0]; o.pw = (int)pads[1]; o.dh = (int)dil[0]; o.dw = (int)dil[1];
    o.act = act; o.res_after_act = res_after;
    const int Ho = conv_out_dim(x.t.h, KH, o.sh, o.ph, o.dh), Wo = conv_out_dim(x.t.w, KW, o.sw, o.pw, o.dw);
    FE_CHECK(Ho > 0 && Wo > 0, "graph: Conv '%s' yields an empty output", n.name.c_str());
    y.t = c.arena.tensor(x.t.n, Ho, Wo, CoutP);
    y.rank = 4;
    if (g.add >= 0) {
      FE_CHECK(res.t.c == CoutP && res.t.n == y.t.n && res.t.h == Ho && res.t.w == Wo, "graph: residual of '%s' has a different shape", n.name.c_str());
      o.res = &res.t;
    }
    if (depthwise) {
      FE_CHECK(o.dh == 1 && o.dw == 1, "graph: dilated depthwise Conv '%s' is not supported", n.name.c_str());
      FE_CHECK(!(g.add >= 0 && res_after), "graph: depthwise Conv '%s': residual after activation is not supported", n.name.c_str());
      launch_dwconv(x.t, y.t, nc.dwt, KH, KW, o.sh, o.sw, o.ph, o.pw, nc.scale, nc.shift, act, nc.slope, o.res, c.stream);
      c.flops_accum += 2.0 * (double)y.t.pixels() * Cout * KH * KW;
    } else {
      conv_forward(c, nc.cw, x.t, y.t, o);
    }
  } else {
    const int M = x.t.n;
    Tensor xm = x.t;
    if (x.flat) { xm.c = xm.ld = x.t.h * x.t.w * x.t.c; xm.h = xm.w = 1; }
    y.t = c.arena.tensor(M, 1, 1, CoutP);
    y.rank = 2;
    FE_CHECK(g.add < 0 || (res.t.c == CoutP && res.t.pixels() == (size_t)M), "graph: residual of '%s' has a different shape", n.name.c_str());
    if (g.add >= 0 && res_after) {
      linear_forward(c, nc.cw, xm.p, xm.ld, M, y.t.p, y.t.ld, act);
      launch_binary(y.t, res.t, y.t, 0, ACT_NONE, c.stream);
    } else {
      linear_forward(c, nc.cw, xm.p, xm.ld, M, y.t.p, y.t.ld, act, g.add >= 0 ? res.t.p : nullptr, g.add >= 0 ? res.t.ld : 0);
    }
  }
  vals_[out_name] = y;
}

// ---------------------------------------------------------------------------------------------------------------------
static Tensor flat_view(float* p, size_t numel) {
  Tensor t;
  t.p = p; t.n = 1; t.h = 1; t.w = (int)((numel + 3) / 4); t.c = 4; t.ld = 4;
  return t;
}

void Graph::exec_node(Ctx& c, int idx) {
  const Node& n = m_.nodes[idx];
  const std::string& op = n.op;
  auto out0 = [&]() -> const std::string& { return n.out[0]; };
  auto host_result = [&](std::vector<int64_t> dims, bool is_int, std::vector<double> v) {
    TensorData t;
    t.name = n.out[0]; t.dims = std::move(dims); t.dtype = is_int ? 7 : 1;
    if (is_int) { t.i.resize(v.size()); for (size_t k = 0; k < v.size(); ++k) t.i[k] = (int64_t)std::llround(v[k]); }
    else { t.f.resize(v.size()); for (size_t k = 0; k < v.size(); ++k) t.f[k] = (float)v[k]; }
    set_host(n.out[0], std::move(t));
  };

  if (op == "Constant") {
    FE_CHECK(n.has("value"), "graph: Constant '%s' without a tensor value", n.name.c_str());
    TensorData t = n.attr.at("value").t;
    set_host(out0(), std::move(t));
    return;
  }
  if (op == "Identity" || op == "Dropout") { vals_[out0()] = get(n.in[0]); return; }

  if (op == "Shape") {
    const Val& v = get(n.in[0]);
    std::vector<double> d;
    if (v.kind == Val::IMG) {
      if (v.rank == 2) d = {(double)v.t.n, (double)v.lc};
      else if (v.flat) d = {(double)v.t.n, (double)v.lc * v.t.h * v.t.w};
      else d = {(double)v.t.n, (double)v.lc, (double)v.t.h, (double)v.t.w};
    } else if (v.kind == Val::PLAIN) for (auto x : v.dims) d.push_back((double)x);
    else for (auto x : v.host->dims) d.push_back((double)x);
    host_result({(int64_t)d.size()}, true, d);
    return;
  }

  // ---- host-side arithmetic on small constants (shape computations of exporters) ------------------------------------
  const TensorData* h0 = n.in.size() > 0 ? host_of(n.in[0]) : nullptr;
  const TensorData* h1 = n.in.size() > 1 ? host_of(n.in[1]) : nullptr;
  if (h0 && (op == "Gather" || op == "Unsqueeze" || op == "Squeeze" || op == "Cast" || op == "Concat" || op == "Slice" ||
             op == "Floor" || op == "Ceil" || ((op == "Add" || op == "Sub" || op == "Mul" || op == "Div") && h1))) {
    if (op == "Concat") {
      std::vector<double> v;
      bool all_int = true;
      for (auto& s : n.in) {
        const TensorData* t = host_of(s);
        FE_CHECK(t && t->dims.size() <= 1, "graph: Concat '%s' mixes constants and tensors", n.name.c_str());
        all_int = all_int && t->is_int();
        for (size_t k = 0; k < t->numel(); ++k) v.push_back(t->at(k));
      }
      host_result({(int64_t)v.size()}, all_int, v);
      return;
    }
    std::vector<double> a(h0->numel());
    for (size_t k = 0; k < a.size(); ++k) a[k] = h0->at(k);
    if (op == "Gather") {
      FE_CHECK(h1 && n.geti("axis", 0) == 0 && h0->dims.size() <= 1, "graph: Gather '%s' on constants needs axis 0", n.name.c_str());
      std::vector<double> v;
      for (size_t k = 0; k < h1->numel(); ++k) {
        int64_t j = (int64_t)h1->at(k);
        if (j < 0) j += (int64_t)a.size();
        FE_CHECK(j >= 0 && j < (int64_t)a.size(), "graph: Gather '%s' index out of range", n.name.c_str());
        v.push_back(a[j]);
      }
      host_result(h1->dims, h0->is_int(), v);
      return;
    }
    if (op == "Unsqueeze") { host_result({(int64_t)a.size()}, h0->is_int(), a); return; }
    if (op == "Squeeze") { host_result(a.size() == 1 ? std::vector<int64_t>{} : std::vector<int64_t>{(int64_t)a.size()}, h0->is_int(), a); return; }
    if (op == "Cast") {
      const int64_t to = n.geti("to", 1);
      const bool ti = to == 6 || to == 7 || to == 2 || to == 3 || to == 9;
      if (ti && !h0->is_int()) for (auto& x : a) x = std::trunc(x);
      host_result(h0->dims, ti, a);
      return;
    }
    if (op == "Floor") { for (auto& x : a) x = std::floor(x); host_result(h0->dims, h0->is_int(), a); return; }
    if (op == "Ceil") { for (auto& x : a) x = std::ceil(x); host_result(h0->dims, h0->is_int(), a); return; }
    if (op == "Slice") {
      const TensorData *st = host_of(n.in.size() > 1 ? n.in[1] : ""), *en = host_of(n.in.size() > 2 ? n.in[2] : "");
      int64_t s0, e0;
      if (st && en) { s0 = (int64_t)st->at(0); e0 = (int64_t)en->at(0); }
      else { auto ss = n.getints("starts"), ee = n.getints("ends"); FE_CHECK(ss.size() == 1 && ee.size() == 1, "graph: Slice '%s'", n.name.c_str()); s0 = ss[0]; e0 = ee[0]; }
      const int64_t L = (int64_t)a.size();
      if (s0 < 0) s0 += L;
      if (e0 < 0) e0 += L;
      s0 = std::max<int64_t>(0, std::min(L, s0)); e0 = std::max<int64_t>(s0, std::min(L, e0));
      host_result({e0 - s0}, h0->is_int(), std::vector<double>(a.begin() + s0, a.begin() + e0));
      return;
    }
    // binary with numpy broadcasting limited to equal sizes or a scalar
    const size_t na = a.size(), nb = h1->numel(), no = std::max(na, nb);
    FE_CHECK(na == nb || na == 1 || nb == 1, "graph: %s '%s' on constants of %zu and %zu values", op.c_str(), n.name.c_str(), na, nb);
    std::vector<double> v(no);
    const bool ints = h0->is_int() && h1->is_int();
    for (size_t k = 0; k < no; ++k) {
      const double x = a[na == 1 ? 0 : k], y = h1->at(nb == 1 ? 0 : k);
      v[k] = op == "Add" ? x + y : op == "Sub" ? x - y : op == "Mul" ? x * y : (ints ? std::trunc(x / y) : x / y);
    }
    host_result(na >= nb ? h0->dims : h1->dims, ints, v);
    return;
  }

  // ---- activations on their own ------------------------------------------------------------------------------------
  if (is_act_op(op)) {
    const Val& v = get(n.in[0]);
    NodeCache& nc = cache_[idx];
    if (v.kind == Val::PLAIN) {
      FE_CHECK(op == "Relu" || op == "Sigmoid", "graph: %s '%s' on a reshaped tensor is not supported", op.c_str(), n.name.c_str());
      size_t numel = 1;
      for (auto d : v.dims) numel *= (size_t)d;
      Val y = v;
      y.p = (float*)c.arena.alloc((numel + 4) * sizeof(float));
      launch_affine_act(flat_view(v.p, numel), flat_view(y.p, numel), nullptr, nullptr, act_code(n), nullptr, c.stream);
      vals_[out0()] = y;
      return;
    }
    Val x = to_img(c, v);
    const int act = act_code(n);
    if (act == ACT_PRELU && (!nc.built || nc.key != x.t.c)) {
      std::vector<float> sl;
      if (op == "LeakyRelu") sl.assign(x.t.c, n.getf("alpha", 0.01f));
      else { const TensorData* s = host_of(n.in[1]); FE_CHECK(s, "graph: PRelu '%s' needs constant slopes", n.name.c_str()); channel_vector(*s, x.lc, x.t.c, 0.f, sl); }
      nc.slope = dw_.upload(sl);
      nc.built = true; nc.key = x.t.c;
    }
    Val y = x;
    y.t = c.arena.tensor(x.t.n, x.t.h, x.t.w, x.t.c);
    launch_affine_act(x.t, y.t, nullptr, nullptr, act, nc.slope, c.stream);
    vals_[out0()] = y;
    return;
  }

  // ---- elementwise binary ------------------------------------------------------------------------------------------
  if (op == "Add" || op == "Sub" || op == "Mul" || op == "Div") {
    const int code = op == "Add" ? 0 : op == "Sub" ? 1 : op == "Mul" ? 2 : 3;
    const TensorData* k = h1 ? h1 : h0;
    if (k) {   // tensor (op) constant: a per-channel affine map
      FE_CHECK(!(h0 && (code == 1 || code == 3)) || h1, "graph: %s '%s' with the constant on the left is not supported", op.c_str(), n.name.c_str());
      const Val& v = get(h1 ? n.in[0] : n.in[1]);
      NodeCache& nc = cache_[idx];
      if (v.kind == Val::PLAIN) {
        FE_CHECK(k->numel() == 1, "graph: %s '%s': only scalar constants apply to reshaped tensors", op.c_str(), n.name.c_str());
        if (!nc.built) {
          const float cv = (float)k->at(0);
          std::vector<float> sc(4, code == 2 ? cv : code == 3 ? 1.0f / cv : 1.0f), sh(4, code == 0 ? cv : code == 1 ? -cv : 0.0f);
          nc.scale = dw_.upload(sc); nc.shift = dw_.upload(sh); nc.built = true;
        }
        size_t numel = 1;
        for (auto d : v.dims) numel *= (size_t)d;
        Val y = v;
        y.p = (float*)c.arena.alloc((numel + 4) * sizeof(float));
        launch_affine_act(flat_view(v.p, numel), flat_view(y.p, numel), nc.scale, nc.shift, ACT_NONE, nullptr, c.stream);
        vals_[out0()] = y;
        return;
      }
      Val x = to_img(c, v);
      if (!nc.built || nc.key != x.t.c) {
        std::vector<float> cv;
        channel_vector(*k, x.lc, x.t.c, code >= 2 ? 1.f : 0.f, cv);
        std::vector<float> sc(x.t.c, 1.f), sh(x.t.c, 0.f);
        for (int q = 0; q < x.t.c; ++q) {
          if (code == 0) sh[q] = cv[q];
          else if (code == 1) sh[q] = -cv[q];
          else if (code == 2) sc[q] = cv[q];
          else sc[q] = 1.0f / cv[q];
        }
        nc.scale = dw_.upload(sc); nc.shift = dw_.upload(sh); nc.built = true; nc.key = x.t.c;
      }
      Val y = x;
      y.t = c.arena.tensor(x.t.n, x.t.h, x.t.w, x.t.c);
      launch_affine_act(x.t, y.t, nc.scale, nc.shift, ACT_NONE, nullptr, c.stream);
      vals_[out0()] = y;
      return;
    }
    const Val &va = get(n.in[0]), &vb = get(n.in[1]);
    if (va.kind == Val::PLAIN && vb.kind == Val::PLAIN) {
      FE_CHECK(va.dims == vb.dims, "graph: %s '%s' needs equal shapes", op.c_str(), n.name.c_str());
      size_t numel = 1;
      for (auto d : va.dims) numel *= (size_t)d;
      Val y = va;
      y.p = (float*)c.arena.alloc((numel + 4) * sizeof(float));
      launch_binary(flat_view(va.p, numel), flat_view(vb.p, numel), flat_view(y.p, numel), code, ACT_NONE, c.stream);
      vals_[out0()] = y;
      return;
    }
    Val a = to_img(c, va), b = to_img(c, vb);
    FE_CHECK(a.t.c == b.t.c && a.t.n == b.t.n && a.t.h == b.t.h && a.t.w == b.t.w && a.lc == b.lc,
             "graph: %s '%s' needs equal shapes (broadcasting between feature maps is not supported)", op.c_str(), n.name.c_str());
    Val y = a;
    y.t = c.arena.tensor(a.t.n, a.t.h, a.t.w, a.t.c);
    launch_binary(a.t, b.t, y.t, code, ACT_NONE, c.stream);
    vals_[out0()] = y;
    return;
  }

  // ---- pooling / resampling ---------------------------------------------------------------------------------------
  if (op == "MaxPool" || op == "AveragePool") {
    Val x = to_img(c, get(n.in[0]));
    auto ks = n.getints("kernel_shape"), st = n.getints("strides"), pd = n.getints("pads");
    if (st.empty()) st = {1, 1};
    if (pd.empty()) pd = {0, 0, 0, 0};
    FE_CHECK(ks.size() == 2 && ks[0] == ks[1] && st[0] == st[1] && pd[0] == pd[1] && pd[0] == pd[2] && pd[0] == pd[3],
             "graph: %s '%s': only square windows with uniform stride/padding", op.c_str(), n.name.c_str());
    const int k = (int)ks[0], s = (int)st[0], p = (int)pd[0];
    const bool ceil_mode = n.geti("ceil_mode", 0) != 0;
    auto odim = [&](int in) {
      int o = ceil_mode ? (in + 2 * p - k + s - 1) / s + 1 : (in + 2 * p - k) / s + 1;
      if (ceil_mode && (o - 1) * s >= in + p) --o;   // last window must start inside the image or its left padding
      return o;
    };
    Val y = x;
    y.t = c.arena.tensor(x.t.n, odim(x.t.h), odim(x.t.w), x.t.c);
    if (op == "MaxPool") launch_maxpool(x.t, y.t, k, s, p, c.stream);
    else launch_avgpool(x.t, y.t, k, s, p, (int)n.geti("count_include_pad", 0), c.stream);
    vals_[out0()] = y;
    return;
  }
  if (op == "GlobalAveragePool") {
    Val x = to_img(c, get(n.in[0]));
    Val y = x;
    y.t = c.arena.tensor(x.t.n, 1, 1, x.t.c);
    launch_adaptive_avgpool(x.t, y.t, c.stream);
    vals_[out0()] = y;
    return;
  }
  if (op == "Resize" || op == "Upsample") {
    Val x = to_img(c, get(n.in[0]));
    int Ho = 0, Wo = 0;
    const bool old = op == "Upsample" || n.in.size() == 2;   // opset <= 10 forms: (X, scales)
    const TensorData* scales = host_of(old ? n.in[1] : (n.in.size() > 2 ? n.in[2] : ""));
    const TensorData* sizes = op == "Resize" && n.in.size() > 3 ? host_of(n.in[3]) : nullptr;
    if (sizes && sizes->numel() == 4) { Ho = (int)sizes->at(2); Wo = (int)sizes->at(3); }
    else if (scales && scales->numel() == 4) {
      FE_CHECK(scales->at(0) == 1.0 && scales->at(1) == 1.0, "graph: Resize '%s' scales batch/channels", n.name.c_str());
      Ho = (int)std::floor(x.t.h * scales->at(2)); Wo = (int)std::floor(x.t.w * scales->at(3));
    } else FE_CHECK(false, "graph: Resize '%s' needs constant scales or sizes", n.name.c_str());
    const std::string mode = n.gets("mode", "nearest");
    const std::string ctm = n.gets("coordinate_transformation_mode", old ? "asymmetric" : "half_pixel");
    Val y = x;
    y.t = c.arena.tensor(x.t.n, Ho, Wo, x.t.c);
    if (mode == "nearest") {
      FE_CHECK(ctm == "asymmetric" && n.gets("nearest_mode", old ? "floor" : "round_prefer_floor") == "floor",
               "graph: Resize '%s': nearest is supported for asymmetric/floor only (what torch exports)", n.name.c_str());
      launch_nearest(x.t, y.t, c.stream);
    } else {
      FE_CHECK(mode == "linear" && (ctm == "half_pixel" || ctm == "pytorch_half_pixel"),
               "graph: Resize '%s' mode=%s/%s is not supported", n.name.c_str(), mode.c_str(), ctm.c_str());
      launch_bilinear(x.t, y.t, c.stream);
    }
    vals_[out0()] = y;
    return;
  }

  // ---- layout ops --------------------------------------------------------------------------------------------------
  if (op == "Flatten") {
    FE_CHECK(n.geti("axis", 1) == 1, "graph: Flatten '%s' axis must be 1", n.name.c_str());
    Val v = get(n.in[0]);
    if (v.kind == Val::PLAIN) {
      int64_t rest = 1;
      for (size_t k = 1; k < v.dims.size(); ++k) rest *= v.dims[k];
      v.dims = {v.dims[0], rest};
    } else {
      v = to_img(c, v);
      if (v.rank == 4) {
        if (v.t.ld != v.t.c) {   // make dense so a following Gemm can read it in place
          Val d = v;
          d.t = c.arena.tensor(v.t.n, v.t.h, v.t.w, v.t.c);
          launch_affine_act(v.t, d.t, nullptr, nullptr, ACT_NONE, nullptr, c.stream);
          v = d;
        }
        v.flat = true;
      }
    }
    vals_[out0()] = v;
    return;
  }
  if (op == "Concat") {
    const int64_t axis = n.geti("axis", 1);
    std::vector<Val> parts;
    for (auto& s : n.in) parts.push_back(to_img(c, get(s)));
    FE_CHECK(axis == 1 && !parts.empty(), "graph: Concat '%s': only channel concatenation of feature maps", n.name.c_str());
    int C = 0;
    for (auto& p : parts) {
      FE_CHECK(p.t.n == parts[0].t.n && p.t.h == parts[0].t.h && p.t.w == parts[0].t.w, "graph: Concat '%s' shapes", n.name.c_str());
      C += p.lc;
    }
    Val y = parts[0];
    y.lc = C;
    y.t = c.arena.tensor(parts[0].t.n, parts[0].t.h, parts[0].t.w, pad_channels(C));
    if (y.t.c != C) FE_HIP(hipMemsetAsync(y.t.p, 0, y.t.numel() * sizeof(float), c.stream));
    int c0 = 0;
    for (auto& p : parts) {   // one pitched device copy per part: rows = pixels, row width = that part's logical channels
      FE_HIP(hipMemcpy2DAsync(y.t.p + c0, (size_t)y.t.ld * sizeof(float), p.t.p, (size_t)p.t.ld * sizeof(float),
                              (size_t)p.lc * sizeof(float), p.t.pixels(), hipMemcpyDeviceToDevice, c.stream));
      c0 += p.lc;
    }
    vals_[out0()] = y;
    return;
  }
  if (op == "Transpose") {
    auto perm = n.getints("perm");
    const Val& v = get(n.in[0]);
    if (v.kind == Val::IMG && v.rank == 4 && !v.flat && perm == std::vector<int64_t>{0, 2, 3, 1}) {
      Val y;
      y.kind = Val::PLAIN;
      y.dims = {v.t.n, v.t.h, v.t.w, v.lc};
      if (v.lc == v.t.c && v.t.ld == v.t.c) y.p = v.t.p;   // NHWC already is this order
      else {
        y.p = (float*)c.arena.alloc((v.t.pixels() * (size_t)v.lc + 4) * sizeof(float));
        const long long d[6] = {1, 1, 1, 1, (long long)v.t.pixels(), v.lc}, s[6] = {0, 0, 0, 0, v.t.ld, 1};
        launch_gather_strided(v.t.p, y.p, d, s, c.stream);
      }
      vals_[out0()] = y;
      return;
    }
    Val x = to_plain(c, v);
    const int r = (int)x.dims.size();
    FE_CHECK(r <= 6, "graph: Transpose '%s' rank %d", n.name.c_str(), r);
    if (perm.empty()) for (int k = r - 1; k >= 0; --k) perm.push_back(k);
    FE_CHECK((int)perm.size() == r, "graph: Transpose '%s' perm size", n.name.c_str());
    std::vector<long long> istr(r, 1);
    for (int k = r - 2; k >= 0; --k) istr[k] = istr[k + 1] * x.dims[k + 1];
    long long d[6] = {1, 1, 1, 1, 1, 1}, s[6] = {0, 0, 0, 0, 0, 0};
    Val y;
    y.kind = Val::PLAIN;
    size_t numel = 1;
    for (int k = 0; k < r; ++k) {
      d[6 - r + k] = x.dims[perm[k]];
      s[6 - r + k] = istr[perm[k]];
      y.dims.push_back(x.dims[perm[k]]);
      numel *= (size_t)x.dims[perm[k]];
    }
    y.p = (float*)c.arena.alloc((numel + 4) * sizeof(float));
    if (numel) launch_gather_strided(x.p, y.p, d, s, c.stream);
    vals_[out0()] = y;
    return;
  }
  if (op == "Reshape" || op == "Squeeze" || op == "Unsqueeze") {
    Val x = to_plain(c, get(n.in[0]));
    size_t numel = 1;
    for (auto d : x.dims) numel *= (size_t)d;
    std::vector<int64_t> nd;
    if (op == "Reshape") {
      const TensorData* sh = host_of(n.in.size() > 1 ? n.in[1] : "");
      std::vector<int64_t> want;
      if (sh) for (size_t k = 0; k < sh->numel(); ++k) want.push_back((int64_t)sh->at(k));
      else want = n.getints("shape");
      FE_CHECK(!want.empty(), "graph: Reshape '%s' needs a constant shape", n.name.c_str());
      int64_t known = 1;
      int infer = -1;
      for (size_t k = 0; k < want.size(); ++k) {
        int64_t d = want[k];
        if (d == 0) { FE_CHECK(k < x.dims.size(), "graph: Reshape '%s' 0-dim", n.name.c_str()); d = x.dims[k]; }
        if (d == -1) { FE_CHECK(infer < 0, "graph: Reshape '%s' has two -1", n.name.c_str()); infer = (int)k; nd.push_back(1); continue; }
        nd.push_back(d);
        known *= d;
      }
      if (infer >= 0) { FE_CHECK(known > 0 && numel % (size_t)known == 0, "graph: Reshape '%s' cannot infer -1", n.name.c_str()); nd[infer] = (int64_t)(numel / (size_t)known); }
      size_t nn = 1;
      for (auto d : nd) nn *= (size_t)d;
      FE_CHECK(nn == numel, "graph: Reshape '%s' changes the element count", n.name.c_str());
    } else {
      std::vector<int64_t> axes = n.getints("axes");
      const TensorData* ax = host_of(n.in.size() > 1 ? n.in[1] : "");
      if (ax) { axes.clear(); for (size_t k = 0; k < ax->numel(); ++k) axes.push_back((int64_t)ax->at(k)); }
      nd = x.dims;
      if (op == "Squeeze") {
        std::vector<int64_t> keep;
        for (int k = 0; k < (int)nd.size(); ++k) {
          bool drop = axes.empty() ? nd[k] == 1 : false;
          for (auto a : axes) if ((a < 0 ? a + (int64_t)nd.size() : a) == k) drop = true;
          if (!drop) keep.push_back(nd[k]);
        }
        nd = keep;
      } else {
        const int r = (int)nd.size() + (int)axes.size();
        std::vector<int64_t> o(r, 0);
        for (auto a : axes) o[a < 0 ? a + r : a] = 1;
        int q = 0;
        for (int k = 0; k < r; ++k) if (o[k] == 0) o[k] = nd[q++];
        nd = o;
      }
    }
    x.dims = nd;
    vals_[out0()] = x;
    return;
  }
  if (op == "Softmax") {
    Val x = to_plain(c, get(n.in[0]));
    const int r = (int)x.dims.size();
    int64_t axis = n.geti("axis", m_.opset >= 13 ? -1 : 1);
    if (axis < 0) axis += r;
    FE_CHECK(r >= 1 && (axis == r - 1 || m_.opset < 13), "graph: Softmax '%s' must run over the last axis", n.name.c_str());
    size_t rows = 1, d = 1;
    for (int k = 0; k < r; ++k) (k < axis ? rows : d) *= (size_t)x.dims[k];
    Val y = x;
    y.p = (float*)c.arena.alloc((rows * d + 4) * sizeof(float));
    FE_HIP(hipMemcpyAsync(y.p, x.p, rows * d * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
    launch_softmax_rows(y.p, (int)d, (int)rows, (int)d, c.stream);
    vals_[out0()] = y;
    return;
  }
  if (op == "Clip") {
    // relu6-style clamps are not part of the face models; report clearly instead of guessing
    FE_CHECK(false, "graph: Clip '%s' is not supported", n.name.c_str());
  }
  FE_CHECK(false, "graph: operator %s ('%s') is not supported by the engine's ONNX subset", op.c_str(), n.name.c_str());
}

void Graph::run(Ctx& c, const Tensor& x, int lc, std::vector<GraphOutput>& outs) {
  FE_CHECK(!m_.nodes.empty(), "graph: no model loaded");
  FE_CHECK(x.c % 4 == 0 && x.ld % 4 == 0 && lc >= 1 && lc <= x.c, "graph: input view must have a multiple-of-4 channel count");
  const auto& in = m_.inputs[0];
  if (in.dims.size() == 4) {
    FE_CHECK(in.dims[1] < 0 || in.dims[1] == lc, "graph: model expects %lld input channels, got %d", (long long)in.dims[1], lc);
    FE_CHECK((in.dims[2] < 0 || in.dims[2] == x.h) && (in.dims[3] < 0 || in.dims[3] == x.w),
             "graph: model expects %lldx%lld input, got %dx%d", (long long)in.dims[2], (long long)in.dims[3], x.h, x.w);
  }
  vals_.clear();
  temps_.clear();
  Val vin;
  vin.kind = Val::IMG; vin.t = x; vin.lc = lc; vin.rank = 4;
  vals_[in.name] = vin;
  const int nn = (int)m_.nodes.size();
  for (int k = 0; k < nn; ++k) {
    if (group_end_[k].main >= 0) exec_group(c, k);
    else if (!absorbed_[k]) exec_node(c, k);
  }
  outs.clear();
  for (auto& o : m_.outputs) {
    Val v = get(o.name);
    FE_CHECK(v.kind != Val::HOST, "graph: output '%s' is a constant", o.name.c_str());
    Val p = to_plain(c, v);
    GraphOutput go;
    go.name = o.name;
    go.dims = p.dims;
    go.dev = p.p;
    go.numel = 1;
    for (auto d : p.dims) go.numel *= (size_t)d;
    outs.push_back(std::move(go));
  }
  vals_.clear();
}

}  // namespace fe

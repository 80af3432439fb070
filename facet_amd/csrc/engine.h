// Engine-side runtime: context, weight store, layer builders and the model graphs of the hot path.
#pragma once
#include "fe_common.h"
#include <memory>
#include <tuple>

namespace fe {

// Host copy of one named checkpoint tensor (fp32, contiguous, PyTorch layout).
struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

class WeightStore {
 public:
  void clear() { t_.clear(); }
  void set(const std::string& name, const float* data, const int64_t* shape, int ndim);
  bool has(const std::string& name) const { return t_.count(name) != 0; }
  const HostTensor& get(const std::string& name) const;
  size_t size() const { return t_.size(); }
 private:
  std::map<std::string, HostTensor> t_;
};

// Owns device copies of packed weights for one model; freed together.
class DeviceWeights {
 public:
  int prec = 0;   // Precision the owning model is built for: pack_conv also emits the 2-byte forms (bf16 / f16) when it is not PREC_F32
  bool res32 = false;   // FE_PRECISION_RES32: the model keeps its residual / skip streams in fp32 around 2-byte GEMM operands
  bool split3 = false;  // FE_PRECISION_SPLIT3 (CLIP image tower only)
  bool half_only = false;   // 2-byte models whose every layer has a 2-byte form (the VLM decoder): no fp32 / Winograd / tap copies uploaded
  ~DeviceWeights() { release(); }
  float* upload(const std::vector<float>& v);
  void* upload_raw(const void* data, size_t bytes);   // any element type (bf16 weights)
  void release();
  size_t bytes() const { return bytes_; }
 private:
  std::vector<void*> ptrs_;
  size_t bytes_ = 0;
};

struct Linear {  // y = x W^T + b, stored as a 1x1 ConvW
  ConvW w;
};

struct LayerNormW {
  float* g = nullptr; float* b = nullptr; int d = 0; float eps = 1e-5f;
};

// Precision of a model's activations / weights, chosen per context before fe_weights_commit (fe_set_precision).
enum Precision : int { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };
constexpr int PREC_RES32_FLAG = 16;   // or-ed into fe_set_precision's argument (include/facet_engine.h FE_PRECISION_RES32)
constexpr int PREC_SPLIT3_FLAG = 32;  // FE_PRECISION_SPLIT3: split-operand fp16 GEMMs (the ViT tower; implies fp32 streams)
template <class T> struct PrecOf { static constexpr int value = PREC_F32; };
template <> struct PrecOf<bf16> { static constexpr int value = PREC_BF16; };
template <> struct PrecOf<f16> { static constexpr int value = PREC_F16; };
// round-to-nearest-even float -> bf16 bits on the host (NaN stays NaN)
inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// round-to-nearest-even float -> fp16 bits on the host, saturating at +-65504 like the device stores (NaN stays NaN)
inline uint16_t f32_to_f16_bits(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7FFFFFFFu;
  if (u > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
  if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7BFFu);           // >= 65520 rounds past the largest finite value: saturate
  if (u < 0x38800000u) {                                             // below 2^-14: subnormal half (or zero)
    if (u < 0x33000000u) return (uint16_t)sign;                      // < 2^-25 rounds to zero
    const int shift = 126 - (int)(u >> 23);                          // 14 .. 24
    const uint32_t m = (u & 0x7FFFFFu) | 0x800000u;
    const uint32_t h = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    return (uint16_t)(sign | (h + ((rem > half || (rem == half && (h & 1))) ? 1 : 0)));
  }
  const uint32_t h = ((u - 0x38000000u) >> 13), rem = u & 0x1FFFu;
  return (uint16_t)(sign | (h + ((rem > 0x1000u || (rem == 0x1000u && (h & 1))) ? 1 : 0)));
}
inline uint16_t f32_to_half_bits(float f, int prec) { return prec == PREC_F16 ? f32_to_f16_bits(f) : f32_to_bf16_bits(f); }
// Adds the bf16 forms (ConvW.wh / wtap_h) of a weight to an already packed ConvW; a no-op for layers the bf16 kernel cannot take
// (3-channel first layers: those run on the fp32 stem / generic kernels and hand a bf16 tensor to the next layer).
void pack_conv_bf16(DeviceWeights& dw, const HostTensor& w, ConvW& c);

// Build helpers ------------------------------------------------------------------------------------
// `bf16` below: also build the bf16 form (models committed under PREC_BF16).
// packs an OIHW (or [out][in]) weight for the conv kernels; scale/shift are per-Cout epilogue vectors (nullable)
ConvW pack_conv(DeviceWeights& dw, const HostTensor& w, const std::vector<float>* scale, const std::vector<float>* shift);
// conv (OIHW) with optional bias and optional eval-mode BatchNorm folded into per-channel scale/shift.
ConvW build_conv(DeviceWeights& dw, const WeightStore& ws, const std::string& conv_prefix,
                 const std::string& bn_prefix, bool conv_bias, float bn_eps = 1e-5f);
// nn.Linear weight [out][in] (+bias) as a 1x1 conv
ConvW build_linear(DeviceWeights& dw, const WeightStore& ws, const std::string& prefix, bool bias);
// same from explicit tensor names / row ranges (for fused in_proj splits)
ConvW build_linear_rows(DeviceWeights& dw, const HostTensor& w, const HostTensor* b, int row0, int rows);
LayerNormW build_ln(DeviceWeights& dw, const WeightStore& ws, const std::string& prefix, float eps = 1e-5f);

struct Ctx;

// nn.MultiheadAttention (batch_first irrelevant: tokens of one image are contiguous rows).
struct MHAW {
  ConvW q, k;          // projections with bias; q carries the 1/sqrt(head_dim) scaling in scale/shift
  float* wv = nullptr; // raw [d][d] V weight: used as the A operand so the GEMM emits V^T directly
  void* wv_h = nullptr; // the same in the model's 2-byte type (bf16 / f16 models)
  float* bv = nullptr; // [d] V bias, added after P.V (softmax rows sum to 1)
  ConvW out;           // out_proj
  int d = 0, heads = 0;
};
MHAW build_mha(DeviceWeights& dw, const WeightStore& ws, const std::string& prefix, int heads);
// The layer wrappers below exist for T = float (fp32 path) and T = bf16 (models committed under PREC_BF16): same call shapes,
// overloads / explicit instantiations in engine.hip.
// y[B*Lq][d] = res + out_proj(softmax(q k^T / sqrt(hd)) v); q from q_in, k/v from kv_in (rows = tokens).
// RT = type of the residual / output rows: T, or float around 2-byte projections (fp32 token stream, FE_PRECISION_RES32)
template <class T, class RT>
void mha_forward(Ctx& c, const MHAW& m, const T* q_in, int ldq, const T* kv_in, int ldkv, int B, int Lq,
                 int Lk, const RT* res, int ldr, RT* y, int ldy, bool causal = false);
// y[M][N] = act(x[M][K] W^T + b) (+res)
void linear_forward(Ctx& c, const ConvW& w, const float* x, int ldx, int M, float* y, int ldy, int act,
                    const float* res = nullptr, int ldr = 0);
void linear_forward(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, bf16* y, int ldy, int act,
                    const bf16* res = nullptr, int ldr = 0);
void linear_forward(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, f16* y, int ldy, int act,
                    const f16* res = nullptr, int ldr = 0);
// 2-byte operand rows, fp32 residual rows (nullable) and fp32 result rows (the projections that write an fp32 token stream)
void linear_forward_s32(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float* y, int ldy, int act, const float* res = nullptr, int ldr = 0);
void linear_forward_s32(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, float* y, int ldy, int act, const float* res = nullptr, int ldr = 0);
// y = act(x W^T + b) + res with y / res of the caller's stream type: (T, T) = linear_forward, (2-byte, float) = linear_forward_s32
void linear_forward_res(Ctx& c, const ConvW& w, const float* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr);
void linear_forward_res(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, bf16* y, int ldy, int act, const bf16* res, int ldr);
void linear_forward_res(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, f16* y, int ldy, int act, const f16* res, int ldr);
void linear_forward_res(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr);
void linear_forward_res(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, float* y, int ldy, int act, const float* res, int ldr);
// same with fp32 outputs whatever the activation type (the last layer of a head: scores / features leave the engine in fp32)
void linear_forward_f32(Ctx& c, const ConvW& w, const float* x, int ldx, int M, float* y, int ldy, int act);
void linear_forward_f32(Ctx& c, const ConvW& w, const bf16* x, int ldx, int M, float* y, int ldy, int act);
void linear_forward_f32(Ctx& c, const ConvW& w, const f16* x, int ldx, int M, float* y, int ldy, int act);
// fp32 rows in, fp32 rows out, on the 2-byte copy of the weights (type `prec` = the model's): weight rows streamed once per 32 rows,
// activations never rounded (per-image head vectors of a FE_PRECISION_RES32 model)
void linear_forward_xf32(Ctx& c, const ConvW& w, int prec, const float* x, int ldx, int M, float* y, int ldy, int act);
template <class T>
inline TensorT<T> mat_view(const T* p, int rows, int cols, int ld) {
  TensorT<T> t; t.p = const_cast<T*>(p); t.n = 1; t.h = 1; t.w = rows; t.c = cols; t.ld = ld; return t;
}

// Runs conv on views; allocates nothing. Output spatial dims must already be set on y.
void conv_forward(Ctx& c, const ConvW& w, const Tensor& x, const Tensor& y, const ConvOpts& o);
void conv_forward(Ctx& c, const ConvW& w, const TensorH& x, const TensorH& y, const ConvOptsT<bf16>& o);
void conv_forward(Ctx& c, const ConvW& w, const TensorF16& x, const TensorF16& y, const ConvOptsT<f16>& o);
// Allocates the output from the arena with the standard conv output size.
template <class T>
TensorT<T> conv_new(Ctx& c, const ConvW& w, const TensorT<T>& x, const ConvOptsT<T>& o);
// First layer of a network: fp32 NHWC4 pixels in, activations of type T out (the 3-channel layers always run on the fp32 stem /
// generic kernels; with T = bf16 their epilogue - or one conversion pass - hands bf16 to the rest of the network).
template <class T>
TensorT<T> first_conv(Ctx& c, const ConvW& w, const Tensor& x, const ConvOpts& o);
inline int conv_out_dim(int in, int k, int s, int p, int d) { return (in + 2 * p - d * (k - 1) - 1) / s + 1; }

// ---- ResNet (timm resnet50 features_only / torchvision resnet18 children[:-2]) -------------------
struct ResBlock {
  ConvW c1, c2, c3, down;
  bool has_down = false, bottleneck = true;
  int stride = 1;
  // stride-1 bottlenecks with a 64-channel 3x3 of a 2-byte model: conv2 + conv3 + identity + ReLU as one launch (kernels_c64.hip)
  void* frag2 = nullptr;
  void* frag3 = nullptr;
};
// kernels_n16.hip: fp32 3x3 convolution to 16 output channels (U2-Net-P's REBNCONV mid layers)
float* build_n16_weights(DeviceWeights& dw, const float* W, int cin);
void launch_conv3x3_n16_f32(const Tensor& x, const Tensor& y, const float* wt, const float* scale, const float* shift, int act, hipStream_t s);
// fragment blobs of kernels_c64.hip (W2 [64][64][3][3]; W3 [256][64] or null: the plain 3x3)
void build_c64_fragments(DeviceWeights& dw, const float* W2, const float* W3, void** frag2, void** frag3);
template <class E>
void launch_conv3x3_c64(const TensorT<E>& x, const TensorT<E>& y, const TensorT<E>* res, const void* frag2, const void* frag3, const float* scale2, const float* shift2,
                        const float* scale3, const float* shift3, int act2, hipStream_t s);
struct ResNet {
  ConvW stem;
  std::vector<std::vector<ResBlock>> layers;
  bool bottleneck = true;
};
// keys: prefix + {conv1,bn1,layer1.0.conv1,...} (timm/torchvision naming) or the Sequential numbering
// of reference models/samp_net.py:652-662 when `seq_names` is true (0=conv1,1=bn1,4..7=layer1..4).
void build_resnet(ResNet& r, DeviceWeights& dw, const WeightStore& ws, const std::string& prefix,
                  bool bottleneck, const int blocks[4], bool seq_names);
// feats (optional) receives [stem-relu, layer1..layer4]; returns layer4 output. x is always the fp32 NHWC4 image tensor.
// res32 (2-byte T only; FE_PRECISION_RES32): the skip stream of the network is kept in fp32 - every block's output leaves its
// last convolution both as fp32 (the next block's identity / the downsample branch's output) and as T (the operand of the next
// convolutions); last32 (optional) receives the fp32 form of the returned map.
template <class T>
TensorT<T> resnet_forward(Ctx& c, const ResNet& r, const Tensor& x_nhwc4, std::vector<TensorT<T>>* feats, bool res32 = false, Tensor* last32 = nullptr);

// ---- PIL-exact uint8 resampling (kernels_resize.hip) --------------------------------------------------
enum ResizeFilter : int { FE_FILTER_LANCZOS = 1, FE_FILTER_BILINEAR = 2, FE_FILTER_BICUBIC = 3 };  // PIL's enum values
struct ResizeCoeffs { std::vector<int> kk, bounds; int ksize = 0, out = 0; };
struct ResizeCoeffsDev { int* kk = nullptr; int* bounds = nullptr; int ksize = 0; };
void build_resize_coeffs(int in_size, int out_size, int filter, ResizeCoeffs& rc);
void resize_u8(Ctx& c, const uint8_t* d_src, int n, int h, int w, int oh, int ow, int filter, int y0, int ch, int x0,
               int cw, uint8_t* d_dst);

class Graph;   // onnx_graph.h
struct GraphSlot;

// geometry the checkpoint's tensor shapes do not determine (transformers Qwen2_5_VLTextConfig): defaults = Qwen2.5-VL-7B-Instruct
struct VlmConfig {
  int n_heads = 28, n_kv_heads = 4, head_dim = 128; float rope_theta = 1e6f, rms_eps = 1e-6f; int mrope[3] = {16, 24, 24};
  int vis_heads = 16, fullatt[8] = {7, 15, 23, 31, 0, 0, 0, 0}, n_fullatt = 4;      // vision tower (Qwen2_5_VLVisionConfig)
};

struct OpTiming { std::string name; double flops; double bytes; float ms; };

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  Arena arena;
  std::string err;
  std::mutex mu;
  // per-op profiling (off by default): records events around each conv launch
  bool profile = false;
  std::vector<OpTiming> timings;
  double flops_accum = 0.0;
  double flops_saved = 0.0;   // algorithmic FLOPs NOT executed because a layer ran as Winograd (executed = flops_accum - flops_saved)
  double flops_half = 0.0;    // the part of flops_accum issued on the 2-byte matrix instructions (bf16 / f16 models): a mixed-precision
                              // run is priced per dtype (bench.py roofline: fp32 FLOPs / fp32 peak + 2-byte FLOPs / 2-byte peak)
  int force_variant = 0;   // developer hook: forwarded to ConvParams.variant
  int precision = PREC_F32;   // what the NEXT fe_weights_commit builds (fe_set_precision); each model remembers its own
  bool res32 = false;         // FE_PRECISION_RES32 of the next commit: fp32 residual streams around 2-byte GEMM operands
  bool split3 = false;        // FE_PRECISION_SPLIT3 of the next commit
  // TOPIQ GatedConv activations picked up by the next fe_weights_commit(FE_MODEL_TOPIQ) (fe_topiq_configure)
  int topiq_gate_act = ACT_GELU, topiq_wblk_act = ACT_GELU;
  size_t topiq_f32_below = 0;   // 2-byte TOPIQ: images with fewer pixels run on the model's fp32 weights (fe_topiq_f32_below; 0 = never)

  std::map<std::tuple<int, int, int>, ResizeCoeffsDev> resize_cache;  // (in, out, filter) -> device tables
  WeightStore staging[8];
  std::unique_ptr<struct TopiqModel> topiq;
  std::unique_ptr<struct U2NetPModel> u2netp;
  std::unique_ptr<struct SampModel> samp;
  std::unique_ptr<struct ClipModel> clip;
  std::unique_ptr<struct AestheticModel> aesthetic;
  std::unique_ptr<struct ClipTextModel> clip_text;
  std::unique_ptr<struct VlmModel> vlm;
  VlmConfig vlm_cfg;          // read by the next fe_weights_commit(FE_MODEL_VLM) (fe_vlm_configure)
  std::unique_ptr<GraphSlot> graphs[8];   // ONNX graphs (face detector / landmarks / recognition, ...)
  ~Ctx();
};

// pre-norm transformer layers of pyiqa's CFANet (DETR-style) [DEP-KNOWLEDGE]
struct EncLayerW { MHAW attn; ConvW lin1, lin2; LayerNormW n1, n2; };
struct DecLayerW { MHAW cross; ConvW lin1, lin2; LayerNormW n1, n2, n3; };
struct GatedConvW {
  ConvW split_x1, w0, w2, w4;   // w0 = weight_blk[0] composed with the x2 half of splitconv
  // 64-channel level of a 2-byte model: the whole gate + its 16 x 16 average pool as ONE launch (kernels_gate.hip); null otherwise
  void* fused = nullptr;
  float* fused_bias = nullptr;
  void* w2_frag = nullptr;      // 2-byte models: w2 (3x3, 64 -> 64) in the halo-tiled kernel's fragment order (kernels_c64.hip)
};
void build_gate64_fragments(DeviceWeights& dw, GatedConvW& g, const float* W0, const float* B0, const float* W2, const float* B2, const float* W4, float b4,
                            const float* Wx, const float* Bx);
template <class E>
void launch_topiq_gate64(const TensorT<E>& x, const TensorT<E>& out, const void* frag, const float* bias, int wblk_act, int gate_act, hipStream_t s);

struct TopiqModel {
  DeviceWeights dw;
  ResNet backbone;
  bool has_head = false;
  int gate_act = ACT_GELU;   // activation of the gated branch x1 (pyiqa GatedConv.act)
  int wblk_act = ACT_GELU;   // activation after weight_blk[0] and weight_blk[2]
  GatedConvW gate[5];
  ConvW dim_reduce[5];
  EncLayerW sa[5];
  DecLayerW cross[4];
  EncLayerW pool;
  LayerNormW s_ln0, s_ln3;
  ConvW s_l1, s_l4, s_l6;
  std::vector<float> h_emb, w_emb;           // host copies [128][32] each
  std::map<std::pair<int, int>, float*> pos; // (th,tw) -> device [th*tw][256]
};
// ---- U2-Net-P + SAMP-Net (reference models/samp_net.py) ---------------------------------------------
struct RSUW {
  ConvW in;
  std::vector<ConvW> enc;  // rebnconv1..L
  std::vector<ConvW> dec;  // rebnconv(L-1)d .. rebnconv1d
  int depth = 0;
  bool dilated = false;    // RSU4F
};
struct U2NetPModel {
  DeviceWeights dw;
  RSUW stage[11];          // stage1..6, stage5d..1d
  ConvW side[6], outconv;
};
struct SampModel {
  DeviceWeights dw;
  ResNet backbone;         // ResNet-18 trunk
  ConvW pattern_weight, pattern[8], att_feat, att_pred, com0, com3, com5;
};
void build_u2netp(U2NetPModel& m, const WeightStore& ws);
void build_sampnet(SampModel& m, const WeightStore& ws);
// T = activation type (float | bf16); x_nhwc4 is always the fp32 pixel tensor, the score outputs are always fp32
template <class T>
void u2netp_forward(Ctx& c, const U2NetPModel& m, const Tensor& x_nhwc4, const TensorT<T>& sal);
// TS = element type of the saliency map (U2-Net-P may be committed under another precision than SAMP-Net)
template <class T, class TS>
void sampnet_forward(Ctx& c, const SampModel& m, const Tensor& x_nhwc4, const TensorT<TS>& sal, float* pw, float* attrs,
                     float* dist);

// ---- CLIP ViT image tower + aesthetic MLP -------------------------------------------------------------
struct ClipBlockW {
  LayerNormW ln1, ln2; MHAW attn; ConvW fc, proj;
  // split-operand forms (ClipModel::split3): fused q|k|v and c_fc as [Wh | Wh | Wl] (K' = 3K, against activation rows [xh | xl] read
  // with a wrap), out_proj as [Wh | Wl] (K' = 2K against the plain fp16 attention output read twice), c_proj as [Wh | Wh | Wl]
  ConvW qkv3, out2, fc3, proj3;
};
struct ClipModel {
  DeviceWeights dw;   // dw.prec = the precision this model was committed under
  ConvW patch, proj;
  float* pos = nullptr; float* cls = nullptr;
  LayerNormW ln_pre, ln_post;
  std::vector<ClipBlockW> blocks;
  int width = 0, heads = 0, tokens = 0, patch_size = 0, out_dim = 0;
  // FE_PRECISION_F16 | FE_PRECISION_SPLIT3: every weight and every GEMM operand that LayerNorm or GELU produces is carried as an fp16
  // pair hi + lo (hi = fp16(x), lo = fp16(x - hi): ~22 significant bits) and the products xh.Wh + xl.Wh + xh.Wl are accumulated in
  // fp32 by ONE launch over the concatenated operands - three times the matrix work of plain fp16, fp32-class results
  bool split3 = false;
  float* zero_bias = nullptr;   // [width] zeros: the V bias rides in the fused projection's shift
};
struct AestheticModel { DeviceWeights dw; ConvW l0, l2; };
// CLIP text tower (open_clip TextTransformer: causal, pooled at the EOT token) — SURVEY §8(f)-3
struct ClipTextModel {
  DeviceWeights dw;
  float* tok_emb = nullptr;   // [vocab][width]
  float* pos = nullptr;       // [ctx][width]
  std::vector<ClipBlockW> blocks;
  LayerNormW ln_final;
  ConvW proj;
  int vocab = 0, width = 0, ctx = 0, heads = 0, out_dim = 0;
};
void build_clip_text(ClipTextModel& m, const WeightStore& ws);
// tokens: device int32 [B][ctx]; eot: device int32 [B] (argmax position per row); feat: device [B][out_dim]
void clip_text_forward(Ctx& c, const ClipTextModel& m, const int* tokens, const int* eot, int B, float* feat);
void build_clip(ClipModel& m, const WeightStore& ws);
void build_aesthetic(AestheticModel& m, const WeightStore& ws);
// T: type of the GEMM operands of the tower (float | bf16 | f16); RT: type of the token (residual) stream and of every LayerNorm
// input - T, or float under FE_PRECISION_RES32; input pixels and output features are fp32 either way
template <class T, class RT = T>
void clip_forward(Ctx& c, const ClipModel& m, const Tensor& x_nhwc4, float* feat);
void clip_forward_split3(Ctx& c, const ClipModel& m, const Tensor& x_nhwc4, float* feat);      // split-operand fp16 tower (ClipModel::split3)
void aesthetic_forward(Ctx& c, const AestheticModel& m, const float* feat, int B, float* raw);
void l2_normalize(Ctx& c, const float* x, float* y, int rows, int d);

// ---- VLM tagger text decoder (transformers Qwen2_5_VLForConditionalGeneration; reference models/vlm_tagger.py) - model_vlm.hip ----------
struct VlmLayerW { ConvW qkv, o, gate, up, down; bf16* ln1 = nullptr; bf16* ln2 = nullptr; };
struct VlmVisionBlockW { ConvW qkv, proj, gate, up, down; bf16* n1 = nullptr; bf16* n2 = nullptr; };
struct VlmVisionW {      // model.visual.* (model_vlm_vision.hip)
  bool present = false;
  ConvW patch, m0, m2;
  std::vector<VlmVisionBlockW> blocks;
  bf16* ln_q = nullptr; float* inv_freq = nullptr;
  int hidden = 0, heads = 0, inter = 0, out_hidden = 0, patch_dim = 0;
  std::vector<int> fullatt;
};
struct VlmModel {
  DeviceWeights dw;
  VlmConfig cfg;
  bf16* embed = nullptr; bf16* norm = nullptr; float* inv_freq = nullptr;
  ConvW lm_head;
  std::vector<VlmLayerW> layers;
  int vocab = 0, hidden = 0, inter = 0;
  VlmVisionW vis;
  bf16* img_embeds = nullptr; int img_rows = 0, img_cap = 0;      // merged image embeddings of the last fe_vlm_encode_images (device)
  // contiguous KV cache: per layer [n_seq][n_kv_heads][max_seq][128] keys (rotated) and values
  std::vector<bf16*> kcache, vcache;
  int cache_B = 0, max_seq = 0, cur_len = 0;
  void reserve_cache(int B, int max_seq);
  void release_cache();
  ~VlmModel() { release_cache(); if (img_embeds) (void)hipFree(img_embeds); }
};
void build_vlm(VlmModel& m, const WeightStore& ws, const VlmConfig& cfg);
void build_vlm_vision(VlmModel& m, const WeightStore& ws);
void vlm_vision_forward(Ctx& c, VlmModel& m, const float* pv, int N, const int* pos, const int* widx, const int* cu_win, int n_win, int max_win,
                        const int* cu_full, int n_full, int max_full, bf16* out);
// row kernels shared by the decoder and the vision tower (model_vlm.hip)
void vlm_rmsnorm(Ctx& c, const bf16* x, int ldx, const bf16* w, bf16* y, int ldy, int rows, int d, float eps);
void vlm_add(Ctx& c, bf16* x, const bf16* y, size_t n);
void vlm_silu_mul(Ctx& c, const bf16* g, const bf16* u, bf16* h, size_t n);
void vlm_put_rows(Ctx& c, bf16* x, const bf16* rows, const int* index, int n, int d);
void vlm_embed(Ctx& c, const VlmModel& m, const int* tok_dev, int rows, bf16* x);
void vlm_forward(Ctx& c, VlmModel& m, bf16* x, const int* pos, int B, int L, int* next_dev, float* logits_dev, const int* len_dev = nullptr);
void vlm_decode_steps(Ctx& c, VlmModel& m, int* tok_dev, int* pos_dev, int B, int n_steps, int* out_dev);

void build_topiq_head(TopiqModel& m, const WeightStore& ws);
// feats: the 5 pyramid levels for nb images; scores_dev: device [nb]
template <class T>
void topiq_head_forward(Ctx& c, TopiqModel& m, const std::vector<TensorT<T>>& feats, float* scores_dev);
// fp32 view of an activation vector: the pointer itself for fp32, an arena copy for bf16
inline const float* to_f32(Ctx&, const float* p, size_t) { return p; }
const float* to_f32(Ctx& c, const bf16* p, size_t n);
const float* to_f32(Ctx& c, const f16* p, size_t n);

}  // namespace fe

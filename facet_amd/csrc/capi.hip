// C ABI of libfacet_engine.so (declared in include/facet_engine.h).
#include "../../include/facet_engine.h"
#include <thread>

#include "engine.h"
#include <type_traits>
#include "lines_host.h"
#include "onnx_graph.h"
#include <algorithm>
#include <cmath>
#include <tuple>

using namespace fe;

struct fe_ctx {
  Ctx c;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  int microbatch = 8;
  // double-buffered H2D staging of uint8 micro-batches on a copy stream (host-buffer entry points): the copy of
  // micro-batch k+1 is issued right after the kernels of micro-batch k were queued, so PCIe overlaps compute
  hipStream_t copy_stream = nullptr;
  uint8_t* stage_buf[2] = {nullptr, nullptr};
  size_t stage_cap[2] = {0, 0};
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  // face path: device copies of OpenCV's interpolation tables (built once / per size)
  short* warp_wtab = nullptr;
  int* hsv_sdiv = nullptr; int* hsv_hdiv = nullptr;   // cv2 HSV division tables
  struct CvResizeTab { int* ofs; short* coef; };
  std::map<std::tuple<int, int, int>, CvResizeTab> cvresize;   // (src, dst, clamp) -> tables
  std::vector<void*> misc_allocs;
  float* clip_in = nullptr;  // preprocessed CLIP crops waiting for a full tower batch (ClipBatcher)
  size_t clip_in_cap = 0;
  float* samp_in = nullptr;  // same for the SAMP-Net / U2-Net-P crops (SampBatcher)
  size_t samp_in_cap = 0;
  float* d_out = nullptr;   // persistent device staging for per-image results
  size_t d_out_cap = 0;
  float* d_rec = nullptr;   // interleaved ensemble records of the host-output entry point
  size_t d_rec_cap = 0;
  int ensemble_mask = 7;    // models fe_ensemble_score runs when loaded: 1 topiq | 2 clip | 4 samp (fe_ensemble_select)
  float* out_buf(size_t floats) {
    if (floats > d_out_cap) {
      if (d_out) (void)hipFree(d_out);
      d_out = nullptr; d_out_cap = 0;
      FE_HIP(hipMalloc((void**)&d_out, floats * sizeof(float)));
      d_out_cap = floats;
    }
    return d_out;
  }
};

static std::string g_create_err;

// Walks a uint8 image batch micro-batch by micro-batch. Device-resident input: pointer arithmetic. Host input: ping-pong
// device buffers filled on the copy stream; get(k) makes the compute stream wait for chunk k, done(k) marks its last
// consumer and starts the copy of chunk k+1 (which then runs under the kernels just queued for chunk k).
class ImageStager {
 public:
  ImageStager(fe_ctx* ctx, const uint8_t* imgs, int n, size_t per_image, int mb, int on_device)
      : x_(ctx), imgs_(imgs), n_(n), per_(per_image), mb_(mb), dev_(on_device) {
    if (!dev_) {
      if (!x_->copy_stream) {
        FE_HIP(hipStreamCreateWithFlags(&x_->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
          FE_HIP(hipEventCreateWithFlags(&x_->ev_copied[i], hipEventDisableTiming));
          FE_HIP(hipEventCreateWithFlags(&x_->ev_consumed[i], hipEventDisableTiming));
        }
      }
      const size_t need = (size_t)std::min(mb_, n_) * per_;
      for (int i = 0; i < 2; ++i)
        if (x_->stage_cap[i] < need) {
          FE_HIP(hipStreamSynchronize(x_->c.stream));
          if (x_->stage_buf[i]) FE_HIP(hipFree(x_->stage_buf[i]));
          x_->stage_buf[i] = nullptr; x_->stage_cap[i] = 0;
          FE_HIP(hipMalloc((void**)&x_->stage_buf[i], need));
          x_->stage_cap[i] = need;
        }
      issue(0);
    }
  }
  int chunks() const { return (n_ + mb_ - 1) / mb_; }
  int count(int k) const { return std::min(mb_, n_ - k * mb_); }
  const uint8_t* get(int k) {
    if (dev_) return imgs_ + (size_t)k * mb_ * per_;
    FE_HIP(hipStreamWaitEvent(x_->c.stream, x_->ev_copied[k & 1], 0));
    return x_->stage_buf[k & 1];
  }
  void done(int k) {
    if (dev_) return;
    FE_HIP(hipEventRecord(x_->ev_consumed[k & 1], x_->c.stream));
    consumed_[k & 1] = true;
    if (k + 1 < chunks()) issue(k + 1);
  }
 private:
  void issue(int k) {
    const int b = k & 1;
    if (consumed_[b]) FE_HIP(hipStreamWaitEvent(x_->copy_stream, x_->ev_consumed[b], 0));
    FE_HIP(hipMemcpyAsync(x_->stage_buf[b], imgs_ + (size_t)k * mb_ * per_, (size_t)count(k) * per_, hipMemcpyHostToDevice,
                          x_->copy_stream));
    FE_HIP(hipEventRecord(x_->ev_copied[b], x_->copy_stream));
  }
  fe_ctx* x_; const uint8_t* imgs_; int n_; size_t per_; int mb_, dev_;
  bool consumed_[2] = {false, false};
};

// every entry point re-selects the context's device: the calling thread may share the process with torch / RCCL
#define FE_API_BEGIN(ctx)                         \
  if (!(ctx)) return FE_ERR_INVALID;              \
  try {                                           \
    (void)hipSetDevice((ctx)->c.device);
// On the error path nothing may stay in flight: queued async copies read the caller's host buffers and write into host
// vectors local to the entry point, both of which die when it returns.
static void fe_drain(fe_ctx* ctx);
#define FE_API_END(ctx)                           \
  }                                               \
  catch (const std::exception& e) {               \
    (ctx)->c.err = e.what();                      \
    fe_drain(ctx);                                \
    return FE_ERR_RUNTIME;                        \
  }                                               \
  return FE_OK;
static void fe_drain(fe_ctx* ctx) {
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  if (ctx->c.stream) (void)hipStreamSynchronize(ctx->c.stream);
}

extern "C" {

const char* fe_version(void) { return "facet_amd 0.1 (gfx950)"; }

int fe_create(int device, size_t arena_bytes, fe_ctx** out) {
  if (!out) return FE_ERR_INVALID;
  *out = nullptr;
  try {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) throw Error(std::string("no HIP device available: ") + hipGetErrorString(e));
    FE_CHECK(device >= 0 && device < ndev, "device %d out of range (%d devices)", device, ndev);
    FE_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FE_HIP(hipGetDeviceProperties(&prop, device));
    FE_CHECK(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, "device %d is %s; this engine is built for gfx950 only",
             device, prop.gcnArchName);
    auto* x = new fe_ctx();
    x->c.device = device;
    FE_HIP(hipStreamCreateWithFlags(&x->c.stream, hipStreamNonBlocking));
    if (!arena_bytes) {
      // default workspace: a quarter of the free HBM, capped at 64 GiB (8 x 1024^2 images of TOPIQ in flight need ~13 GB,
      // 128 SAMP crops ~20 GB); on a 288 GB MI355X that is 64 GiB. Pass an explicit size to override.
      size_t free_b = 0, total_b = 0;
      FE_HIP(hipMemGetInfo(&free_b, &total_b));
      arena_bytes = std::min<size_t>((size_t)64 << 30, std::max<size_t>((size_t)2 << 30, free_b / 4));
    }
    x->c.arena.init(arena_bytes);
    FE_HIP(hipEventCreate(&x->t0));
    FE_HIP(hipEventCreate(&x->t1));
    *out = x;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return FE_ERR_RUNTIME;
  }
  return FE_OK;
}

void fe_destroy(fe_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->c.device);
  (void)hipStreamSynchronize(ctx->c.stream);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->d_rec) (void)hipFree(ctx->d_rec);
  for (int i = 0; i < 2; ++i) {
    if (ctx->stage_buf[i]) (void)hipFree(ctx->stage_buf[i]);
    if (ctx->ev_copied[i]) (void)hipEventDestroy(ctx->ev_copied[i]);
    if (ctx->ev_consumed[i]) (void)hipEventDestroy(ctx->ev_consumed[i]);
  }
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  for (void* q : ctx->misc_allocs) (void)hipFree(q);
  if (ctx->clip_in) (void)hipFree(ctx->clip_in);
  if (ctx->samp_in) (void)hipFree(ctx->samp_in);
  delete ctx;
}

const char* fe_last_error(fe_ctx* ctx) { return ctx ? ctx->c.err.c_str() : g_create_err.c_str(); }

int fe_sync(fe_ctx* ctx) {
  FE_API_BEGIN(ctx)
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  FE_API_END(ctx)
}

int fe_set_microbatch(fe_ctx* ctx, int n) {
  FE_API_BEGIN(ctx)
  FE_CHECK(n >= 1 && n <= 256, "microbatch %d out of range", n);
  ctx->microbatch = n;
  FE_API_END(ctx)
}

int fe_set_conv_variant(fe_ctx* ctx, int variant) {
  FE_API_BEGIN(ctx)
  ctx->c.force_variant = variant;
  FE_API_END(ctx)
}

int fe_dev_alloc(fe_ctx* ctx, size_t bytes, void** d_out) {
  FE_API_BEGIN(ctx)
  FE_CHECK(d_out != nullptr, "null out");
  FE_HIP(hipSetDevice(ctx->c.device));
  FE_HIP(hipMalloc(d_out, bytes ? bytes : 16));
  FE_API_END(ctx)
}
int fe_dev_free(fe_ctx* ctx, void* d_ptr) {
  FE_API_BEGIN(ctx)
  FE_HIP(hipFree(d_ptr));
  FE_API_END(ctx)
}
int fe_memcpy_h2d(fe_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  FE_API_BEGIN(ctx)
  FE_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->c.stream));
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  FE_API_END(ctx)
}
int fe_memcpy_d2h(fe_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  FE_API_BEGIN(ctx)
  FE_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->c.stream));
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  FE_API_END(ctx)
}

int fe_timer_start(fe_ctx* ctx) {
  FE_API_BEGIN(ctx)
  FE_HIP(hipEventRecord(ctx->t0, ctx->c.stream));
  FE_API_END(ctx)
}
int fe_timer_stop(fe_ctx* ctx, float* ms_out) {
  FE_API_BEGIN(ctx)
  FE_HIP(hipEventRecord(ctx->t1, ctx->c.stream));
  FE_HIP(hipEventSynchronize(ctx->t1));
  FE_HIP(hipEventElapsedTime(ms_out, ctx->t0, ctx->t1));
  FE_API_END(ctx)
}
int fe_profile_enable(fe_ctx* ctx, int on) {
  FE_API_BEGIN(ctx)
  ctx->c.profile = on != 0;
  ctx->c.timings.clear();
  FE_API_END(ctx)
}
int fe_profile_count(fe_ctx* ctx) { return ctx ? (int)ctx->c.timings.size() : 0; }
int fe_profile_get(fe_ctx* ctx, int i, char* name, int name_cap, double* flops, double* bytes, float* ms) {
  FE_API_BEGIN(ctx)
  FE_CHECK(i >= 0 && i < (int)ctx->c.timings.size(), "profile index %d", i);
  const OpTiming& t = ctx->c.timings[i];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", t.name.c_str());
  if (flops) *flops = t.flops;
  if (bytes) *bytes = t.bytes;
  if (ms) *ms = t.ms;
  FE_API_END(ctx)
}
int fe_flops_reset(fe_ctx* ctx) {
  FE_API_BEGIN(ctx)
  ctx->c.flops_accum = 0.0;
  ctx->c.flops_saved = 0.0;
  ctx->c.flops_half = 0.0;
  FE_API_END(ctx)
}
int fe_flops_get_half(fe_ctx* ctx, double* flops) {
  FE_API_BEGIN(ctx)
  FE_CHECK(flops, "bad arguments");
  *flops = ctx->c.flops_half;
  FE_API_END(ctx)
}
int fe_flops_get_executed(fe_ctx* ctx, double* flops) {
  FE_API_BEGIN(ctx)
  FE_CHECK(flops, "bad arguments");
  *flops = ctx->c.flops_accum - ctx->c.flops_saved;
  FE_API_END(ctx)
}
int fe_flops_get(fe_ctx* ctx, double* flops) {
  FE_API_BEGIN(ctx)
  *flops = ctx->c.flops_accum;
  FE_API_END(ctx)
}

// ---- weights ------------------------------------------------------------------------------------
int fe_weights_begin(fe_ctx* ctx, int model) {
  FE_API_BEGIN(ctx)
  FE_CHECK(model >= 0 && model < 8, "model id %d", model);
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  ctx->c.staging[model].clear();
  FE_API_END(ctx)
}
int fe_weights_set(fe_ctx* ctx, int model, const char* name, const float* data, const int64_t* shape, int ndim) {
  FE_API_BEGIN(ctx)
  FE_CHECK(model >= 0 && model < 8 && name && data && shape && ndim >= 0 && ndim <= 6, "bad arguments");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  ctx->c.staging[model].set(name, data, shape, ndim);
  FE_API_END(ctx)
}
int fe_set_precision(fe_ctx* ctx, int precision) {
  FE_API_BEGIN(ctx)
  const int base = precision & ~(FE_PRECISION_RES32 | FE_PRECISION_SPLIT3);
  FE_CHECK(!(precision & FE_PRECISION_SPLIT3) || base == FE_PRECISION_F16, "set_precision: FE_PRECISION_SPLIT3 qualifies FE_PRECISION_F16");
  FE_CHECK(base == FE_PRECISION_F32 || base == FE_PRECISION_BF16 || base == FE_PRECISION_F16, "set_precision: %d", precision);
  FE_CHECK(base != FE_PRECISION_F32 || !(precision & FE_PRECISION_RES32), "set_precision: FE_PRECISION_RES32 qualifies a 2-byte precision");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  ctx->c.precision = base;
  ctx->c.res32 = (precision & (FE_PRECISION_RES32 | FE_PRECISION_SPLIT3)) != 0;      // split operands imply fp32 streams
  ctx->c.split3 = (precision & FE_PRECISION_SPLIT3) != 0;
  FE_API_END(ctx)
}
int fe_model_precision(fe_ctx* ctx, int model) {
  if (!ctx) return -1;
  auto code = [](const DeviceWeights& dw) { return dw.prec | (dw.res32 ? FE_PRECISION_RES32 : 0) | (dw.split3 ? FE_PRECISION_SPLIT3 : 0); };
  if (model == FE_MODEL_TOPIQ && ctx->c.topiq) return code(ctx->c.topiq->dw);
  if (model == FE_MODEL_U2NETP && ctx->c.u2netp) return code(ctx->c.u2netp->dw);
  if (model == FE_MODEL_SAMP && ctx->c.samp) return code(ctx->c.samp->dw);
  if (model == FE_MODEL_CLIP && ctx->c.clip) return code(ctx->c.clip->dw);
  if (model == FE_MODEL_AESTHETIC && ctx->c.aesthetic) return FE_PRECISION_F32;
  if (model == FE_MODEL_VLM && ctx->c.vlm) return FE_PRECISION_BF16;
  return -1;
}

int fe_topiq_f32_below(fe_ctx* ctx, long long pixels) {
  FE_API_BEGIN(ctx)
  FE_CHECK(pixels >= 0, "topiq_f32_below: negative pixel count");
  ctx->c.topiq_f32_below = (size_t)pixels;
  FE_API_END(ctx)
}

int fe_topiq_configure(fe_ctx* ctx, int gate_act, int weight_blk_act) {
  FE_API_BEGIN(ctx)
  auto ok = [](int a) { return a == FE_ACT_RELU || a == FE_ACT_GELU || a == FE_ACT_SOFTPLUS; };
  FE_CHECK(ok(gate_act) && ok(weight_blk_act), "topiq_configure: activations must be relu, gelu or softplus");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  ctx->c.topiq_gate_act = gate_act;
  ctx->c.topiq_wblk_act = weight_blk_act;
  FE_API_END(ctx)
}

int fe_weights_commit(fe_ctx* ctx, int model) {
  FE_API_BEGIN(ctx)
  FE_CHECK(model >= 0 && model < 8, "model id %d", model);
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  FE_HIP(hipSetDevice(ctx->c.device));
  WeightStore& ws = ctx->c.staging[model];
  if (model == FE_MODEL_TOPIQ) {
    auto m = std::make_unique<TopiqModel>();
    m->dw.prec = ctx->c.precision; m->dw.res32 = ctx->c.res32;
    m->gate_act = ctx->c.topiq_gate_act;
    m->wblk_act = ctx->c.topiq_wblk_act;
    const int blocks[4] = {3, 4, 6, 3};
    build_resnet(m->backbone, m->dw, ws, "semantic_model.", true, blocks, false);
    if (ws.has("weight_pool.0.splitconv.weight")) build_topiq_head(*m, ws);
    ctx->c.topiq = std::move(m);
  } else if (model == FE_MODEL_U2NETP) {
    auto m = std::make_unique<U2NetPModel>();
    m->dw.prec = ctx->c.precision; m->dw.res32 = ctx->c.res32;
    build_u2netp(*m, ws);
    ctx->c.u2netp = std::move(m);
  } else if (model == FE_MODEL_CLIP) {
    auto m = std::make_unique<ClipModel>();
    m->dw.prec = ctx->c.precision; m->dw.res32 = ctx->c.res32; m->dw.split3 = ctx->c.split3;      // the image tower; the text tower (built below, run once per vocabulary) stays fp32
    build_clip(*m, ws);
    ctx->c.clip = std::move(m);
    if (ws.has("token_embedding.weight")) {   // full CLIP checkpoint: also build the text tower
      auto t = std::make_unique<ClipTextModel>();
      build_clip_text(*t, ws);
      ctx->c.clip_text = std::move(t);
    } else {
      ctx->c.clip_text.reset();
    }
  } else if (model == FE_MODEL_AESTHETIC) {
    auto m = std::make_unique<AestheticModel>();
    build_aesthetic(*m, ws);
    ctx->c.aesthetic = std::move(m);
  } else if (model == FE_MODEL_SAMP) {
    auto m = std::make_unique<SampModel>();
    m->dw.prec = ctx->c.precision; m->dw.res32 = ctx->c.res32;
    build_sampnet(*m, ws);
    ctx->c.samp = std::move(m);
  } else if (model == FE_MODEL_VLM) {
    auto m = std::make_unique<VlmModel>();
    build_vlm(*m, ws, ctx->c.vlm_cfg);      // always bf16: the precision the reference loads it in (models/vlm_tagger.py:155-156)
    ctx->c.vlm = std::move(m);
  } else {
    throw Error("fe_weights_commit: model " + std::to_string(model) + " not implemented");
  }
  ws.clear();
  FE_API_END(ctx)
}
int fe_model_unload(fe_ctx* ctx, int model) {
  FE_API_BEGIN(ctx)
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  if (model == FE_MODEL_TOPIQ) ctx->c.topiq.reset();
  if (model == FE_MODEL_U2NETP) ctx->c.u2netp.reset();
  if (model == FE_MODEL_SAMP) ctx->c.samp.reset();
  if (model == FE_MODEL_CLIP) { ctx->c.clip.reset(); ctx->c.clip_text.reset(); }
  if (model == FE_MODEL_AESTHETIC) ctx->c.aesthetic.reset();
  if (model == FE_MODEL_VLM) ctx->c.vlm.reset();
  FE_API_END(ctx)
}
int fe_model_loaded(fe_ctx* ctx, int model) {
  if (!ctx) return 0;
  if (model == FE_MODEL_TOPIQ) return ctx->c.topiq != nullptr;
  if (model == FE_MODEL_U2NETP) return ctx->c.u2netp != nullptr;
  if (model == FE_MODEL_SAMP) return ctx->c.samp != nullptr;
  if (model == FE_MODEL_CLIP) return ctx->c.clip != nullptr;
  if (model == FE_MODEL_AESTHETIC) return ctx->c.aesthetic != nullptr;
  if (model == FE_MODEL_VLM) return ctx->c.vlm != nullptr;
  return 0;
}

// ---- ONNX graphs -------------------------------------------------------------------------------------
static GraphSlot& graph_slot(fe_ctx* ctx, int slot) {
  FE_CHECK(slot >= 0 && slot < FE_GRAPH_SLOTS, "graph slot %d out of range", slot);
  FE_CHECK(ctx->c.graphs[slot], "no graph loaded in slot %d", slot);
  return *ctx->c.graphs[slot];
}
int fe_onnx_probe(const void* onnx_bytes, size_t len, int* n_nodes, int* n_initializers, int* n_outputs, int64_t in_dims[4],
                  char* err, int err_cap) {
  try {
    onnx::Model m;
    onnx::parse_model((const uint8_t*)onnx_bytes, len, m);
    if (n_nodes) *n_nodes = (int)m.nodes.size();
    if (n_initializers) *n_initializers = (int)m.init.size();
    if (n_outputs) *n_outputs = (int)m.outputs.size();
    if (in_dims)
      for (int k = 0; k < 4; ++k) in_dims[k] = k < (int)m.inputs[0].dims.size() ? m.inputs[0].dims[k] : -1;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) snprintf(err, err_cap, "%s", e.what());
    return FE_ERR_RUNTIME;
  }
  return FE_OK;
}
int fe_graph_load(fe_ctx* ctx, int slot, const void* onnx_bytes, size_t len) {
  FE_API_BEGIN(ctx)
  FE_CHECK(slot >= 0 && slot < FE_GRAPH_SLOTS && onnx_bytes && len > 0, "bad arguments");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  auto gs = std::make_unique<GraphSlot>();
  gs->g.load((const uint8_t*)onnx_bytes, len);
  ctx->c.graphs[slot] = std::move(gs);
  FE_API_END(ctx)
}
int fe_graph_unload(fe_ctx* ctx, int slot) {
  FE_API_BEGIN(ctx)
  FE_CHECK(slot >= 0 && slot < FE_GRAPH_SLOTS, "bad arguments");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  ctx->c.graphs[slot].reset();
  FE_API_END(ctx)
}
int fe_graph_loaded(fe_ctx* ctx, int slot) {
  return ctx && slot >= 0 && slot < FE_GRAPH_SLOTS && ctx->c.graphs[slot] != nullptr;
}
int fe_graph_info(fe_ctx* ctx, int slot, int* n_nodes, int* n_outputs, int64_t in_dims[4], int* flags) {
  FE_API_BEGIN(ctx)
  GraphSlot& gs = graph_slot(ctx, slot);
  const auto& m = gs.g.model();
  if (n_nodes) *n_nodes = (int)m.nodes.size();
  if (n_outputs) *n_outputs = (int)m.outputs.size();
  if (in_dims)
    for (int k = 0; k < 4; ++k) in_dims[k] = k < (int)m.inputs[0].dims.size() ? m.inputs[0].dims[k] : -1;
  if (flags) *flags = (gs.g.head_has_sub() ? 1 : 0) | (gs.g.head_has_mul() ? 2 : 0);
  FE_API_END(ctx)
}
extern "C++" {
template <class T = float>
static TensorT<T> upload_nchw(Ctx& c, const float* x, int n, int ch, int h, int w, int cpad);
}
int fe_graph_run(fe_ctx* ctx, int slot, const float* x, int n, int c, int h, int w, int on_device) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  GraphSlot& gs = graph_slot(ctx, slot);
  FE_CHECK(x && n > 0 && c > 0 && h > 0 && w > 0, "bad arguments");
  C.arena.reset();
  const int cp = Graph::pad_channels(c);
  Tensor xt;
  if (on_device) {
    xt = C.arena.tensor(n, h, w, cp);
    launch_nchw_to_nhwc(x, xt.p, n, c, h, w, cp, C.stream);
  } else {
    xt = upload_nchw(C, x, n, c, h, w, cp);
  }
  std::vector<GraphOutput> outs;
  gs.g.run(C, xt, c, outs);
  gs.last.clear();
  for (auto& o : outs) {
    GraphSlot::Out h_out;
    h_out.name = o.name; h_out.dims = o.dims;
    h_out.data.resize(o.numel);
    if (o.numel) FE_HIP(hipMemcpyAsync(h_out.data.data(), o.dev, o.numel * sizeof(float), hipMemcpyDeviceToHost, C.stream));
    gs.last.push_back(std::move(h_out));
  }
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}
int fe_graph_output_info(fe_ctx* ctx, int slot, int i, char* name, int name_cap, int64_t dims[6], int* rank) {
  FE_API_BEGIN(ctx)
  GraphSlot& gs = graph_slot(ctx, slot);
  FE_CHECK(i >= 0 && i < (int)gs.last.size(), "output index %d out of range (%zu outputs)", i, gs.last.size());
  const auto& o = gs.last[i];
  FE_CHECK(o.dims.size() <= 6, "output rank %zu", o.dims.size());
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", o.name.c_str());
  if (rank) *rank = (int)o.dims.size();
  if (dims) for (size_t k = 0; k < o.dims.size(); ++k) dims[k] = o.dims[k];
  FE_API_END(ctx)
}
int fe_graph_output_copy(fe_ctx* ctx, int slot, int i, float* dst, size_t cap_floats) {
  FE_API_BEGIN(ctx)
  GraphSlot& gs = graph_slot(ctx, slot);
  FE_CHECK(i >= 0 && i < (int)gs.last.size() && dst, "bad arguments");
  FE_CHECK(cap_floats >= gs.last[i].data.size(), "destination holds %zu floats, output has %zu", cap_floats, gs.last[i].data.size());
  if (!gs.last[i].data.empty()) memcpy(dst, gs.last[i].data.data(), gs.last[i].data.size() * sizeof(float));
  FE_API_END(ctx)
}

// ---- single ops -----------------------------------------------------------------------------------
extern "C++" {
template <class T>
static TensorT<T> upload_nchw(Ctx& c, const float* x, int n, int ch, int h, int w, int cpad) {
  const size_t elems = (size_t)n * ch * h * w;
  float* tmp = (float*)c.arena.alloc(elems * sizeof(float));
  FE_HIP(hipMemcpyAsync(tmp, x, elems * sizeof(float), hipMemcpyHostToDevice, c.stream));
  TensorT<T> t = c.arena.tensor_t<T>(n, h, w, cpad);
  launch_nchw_to_nhwc(tmp, t.p, n, ch, h, w, cpad, c.stream);
  return t;
}
}  // extern "C++"
extern "C++" {
template <class T>
static void download_nchw(Ctx& c, const TensorT<T>& t, int ch, float* y) {
  const size_t elems = (size_t)t.n * ch * t.h * t.w;
  float* tmp = (float*)c.arena.alloc(elems * sizeof(float));
  launch_nhwc_to_nchw(t.p, t.ld, tmp, t.n, ch, t.h, t.w, c.stream);
  FE_HIP(hipMemcpyAsync(y, tmp, elems * sizeof(float), hipMemcpyDeviceToHost, c.stream));
  FE_HIP(hipStreamSynchronize(c.stream));
}
}  // extern "C++"

int fe_op_conv2d(fe_ctx* ctx, const float* x, int n, int c, int h, int w, const float* weight, int cout, int kh, int kw,
                 const float* scale, const float* shift, const float* res, int res_after_act, int stride, int pad,
                 int dil, int act, float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(x && weight && y && n > 0 && c > 0 && h > 0 && w > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0,
           "bad conv arguments");
  C.arena.reset();
  DeviceWeights dw;
  dw.prec = C.precision;      // FE_PRECISION_BF16: the same op on the bf16 kernel (inputs rounded to bf16 on upload, fp32 back)
  WeightStore ws;
  const int64_t wshape[4] = {cout, c, kh, kw};
  ws.set("w.weight", weight, wshape, 4);
  ConvW cw = build_conv(dw, ws, "w", "", false);
  std::vector<float> v;
  if (scale) { v.assign(scale, scale + cout); cw.scale = dw.upload(v); }
  if (shift) { v.assign(shift, shift + cout); cw.shift = dw.upload(v); }
  const int ho = conv_out_dim(h, kh, stride, pad, dil), wo = conv_out_dim(w, kw, stride, pad, dil);
  FE_CHECK(ho > 0 && wo > 0, "conv output is empty");
  auto half_op = [&](auto* tag) {      // the same op on the 2-byte kernel (inputs rounded on upload, fp32 back)
    typedef std::remove_pointer_t<decltype(tag)> E;
    FE_CHECK(cw.wh, "fe_op_conv2d(2-byte): Cin must be a multiple of 8 (16 for spatial kernels)");
    TensorT<E> xt = upload_nchw<E>(C, x, n, c, h, w, cw.CinPadH);
    ConvOptsT<E> o;
    o.sh = o.sw = stride; o.ph = o.pw = pad; o.dh = o.dw = dil; o.act = act; o.res_after_act = res_after_act;
    TensorT<E> rt;
    if (C.res32) {
      // FE_PRECISION_RES32: the fp32-stream form of the layer - residual read as fp32, result written both as fp32 rows (returned)
      // and as 2-byte rows, which must be the rounding of the fp32 ones (checked here: this entry point is the kernels' test hook)
      Tensor r32, y32 = C.arena.tensor(n, ho, wo, cout);
      if (res) { r32 = upload_nchw(C, res, n, cout, ho, wo, cout); o.res32 = &r32; }
      o.y32 = &y32;
      TensorT<E> yt = C.arena.tensor_t<E>(n, ho, wo, cout);
      conv_forward(C, cw, xt, yt, o);
      download_nchw(C, y32, cout, y);
      std::vector<float> y16((size_t)n * cout * ho * wo);
      download_nchw(C, yt, cout, y16.data());
      for (size_t i = 0; i < y16.size(); ++i) {
        const float a = std::fmin(std::fmax(y[i], -65504.f), 65504.f);
        FE_CHECK(std::fabs(y16[i] - a) <= std::fabs(a) * (PrecOf<E>::value == PREC_F16 ? 4.9e-4f : 3.95e-3f) + 6.2e-5f,
                 "fe_op_conv2d(res32): 2-byte output %g is not the rounding of the fp32 output %g at %zu", y16[i], y[i], i);
      }
      return;
    }
    if (res) { rt = upload_nchw<E>(C, res, n, cout, ho, wo, cout); o.res = &rt; }
    TensorT<E> yt = conv_new(C, cw, xt, o);
    download_nchw(C, yt, cout, y);
  };
  if (C.precision == PREC_BF16) {
    half_op((bf16*)nullptr);
  } else if (C.precision == PREC_F16) {
    half_op((f16*)nullptr);
  } else {
    Tensor xt = upload_nchw(C, x, n, c, h, w, cw.CinPad);
    ConvOpts o;
    o.sh = o.sw = stride; o.ph = o.pw = pad; o.dh = o.dw = dil; o.act = act; o.res_after_act = res_after_act;
    Tensor rt;
    if (res) { rt = upload_nchw(C, res, n, cout, ho, wo, cout); o.res = &rt; }
    Tensor yt = conv_new(C, cw, xt, o);
    download_nchw(C, yt, cout, y);
  }
  FE_API_END(ctx)
}

int fe_op_topiq_gate64(fe_ctx* ctx, const float* x, int n, int h, int w, const float* w0, const float* b0, const float* w2, const float* b2,
                       const float* w4, float b4, const float* wx, const float* bx, int wblk_act, int gate_act, float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(x && w0 && b0 && w2 && b2 && w4 && wx && bx && y && n > 0 && h > 0 && w > 0 && h % 16 == 0 && w % 16 == 0, "bad gate64 arguments");
  FE_CHECK(C.precision == PREC_BF16 || C.precision == PREC_F16, "fe_op_topiq_gate64: the fused gate exists for the 2-byte element types only");
  C.arena.reset();
  DeviceWeights dw;
  dw.prec = C.precision;
  GatedConvW g;
  build_gate64_fragments(dw, g, w0, b0, w2, b2, w4, b4, wx, bx);
  auto run = [&](auto* tag) {
    typedef std::remove_pointer_t<decltype(tag)> E;
    TensorT<E> xt = upload_nchw<E>(C, x, n, 64, h, w, 64);
    TensorT<E> yt = C.arena.tensor_t<E>(n, h / 16, w / 16, 64);
    launch_topiq_gate64(xt, yt, g.fused, g.fused_bias, wblk_act, gate_act, C.stream);
    download_nchw(C, yt, 64, y);
  };
  if (C.precision == PREC_BF16) run((bf16*)nullptr); else run((f16*)nullptr);
  FE_API_END(ctx)
}

int fe_op_conv3x3_c64(fe_ctx* ctx, const float* x, int n, int h, int w, const float* w2, const float* scale2, const float* shift2, int act2,
                      const float* w3, const float* scale3, const float* shift3, const float* res, float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(x && w2 && y && n > 0 && h > 0 && w > 0 && (!w3 || res), "bad conv3x3_c64 arguments");
  FE_CHECK(C.precision == PREC_BF16 || C.precision == PREC_F16, "fe_op_conv3x3_c64: the halo-tiled kernel exists for the 2-byte element types only");
  C.arena.reset();
  DeviceWeights dw;
  dw.prec = C.precision;
  void *f2 = nullptr, *f3 = nullptr;
  build_c64_fragments(dw, w2, w3, &f2, &f3);
  std::vector<float> v;
  auto up = [&](const float* a, int cnt) -> float* { if (!a) return nullptr; v.assign(a, a + cnt); return dw.upload(v); };
  float *s2 = up(scale2, 64), *h2 = up(shift2, 64), *s3 = up(scale3, 256), *h3 = up(shift3, 256);
  const int cout = w3 ? 256 : 64;
  auto run = [&](auto* tag) {
    typedef std::remove_pointer_t<decltype(tag)> E;
    TensorT<E> xt = upload_nchw<E>(C, x, n, 64, h, w, 64);
    TensorT<E> yt = C.arena.tensor_t<E>(n, h, w, cout);
    TensorT<E> rt;
    if (w3) rt = upload_nchw<E>(C, res, n, 256, h, w, 256);
    launch_conv3x3_c64(xt, yt, w3 ? &rt : nullptr, f2, f3, s2, h2, s3, h3, act2, C.stream);
    download_nchw(C, yt, cout, y);
  };
  if (C.precision == PREC_BF16) run((bf16*)nullptr); else run((f16*)nullptr);
  FE_API_END(ctx)
}

int fe_op_maxpool2d(fe_ctx* ctx, const float* x, int n, int c, int h, int w, int k, int stride, int pad, int ceil_mode,
                    float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  C.arena.reset();
  Tensor xt = upload_nchw(C, x, n, c, h, w, c);
  auto od = [&](int in) {
    int o = ceil_mode ? (in + 2 * pad - k + stride - 1) / stride + 1 : (in + 2 * pad - k) / stride + 1;
    if (ceil_mode && (o - 1) * stride >= in + pad) --o;
    return o;
  };
  Tensor yt = C.arena.tensor(n, od(h), od(w), c);
  launch_maxpool(xt, yt, k, stride, pad, C.stream);
  download_nchw(C, yt, c, y);
  FE_API_END(ctx)
}

int fe_op_bilinear(fe_ctx* ctx, const float* x, int n, int c, int h, int w, int ho, int wo, float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  C.arena.reset();
  Tensor xt = upload_nchw(C, x, n, c, h, w, c);
  Tensor yt = C.arena.tensor(n, ho, wo, c);
  launch_bilinear(xt, yt, C.stream);
  download_nchw(C, yt, c, y);
  FE_API_END(ctx)
}

int fe_op_adaptive_avgpool(fe_ctx* ctx, const float* x, int n, int c, int h, int w, int ho, int wo, float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  C.arena.reset();
  Tensor xt = upload_nchw(C, x, n, c, h, w, c);
  Tensor yt = C.arena.tensor(n, ho, wo, c);
  launch_adaptive_avgpool(xt, yt, C.stream);
  download_nchw(C, yt, c, y);
  FE_API_END(ctx)
}

int fe_op_layernorm(fe_ctx* ctx, const float* x, int rows, int d, const float* g, const float* b, float eps, float* y) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  C.arena.reset();
  const size_t bytes = (size_t)rows * d * sizeof(float);
  float* dx = (float*)C.arena.alloc(bytes);
  float* dy = (float*)C.arena.alloc(bytes);
  float* dg = (float*)C.arena.alloc(d * sizeof(float));
  float* db = (float*)C.arena.alloc(d * sizeof(float));
  FE_HIP(hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(dg, g, d * sizeof(float), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(db, b, d * sizeof(float), hipMemcpyHostToDevice, C.stream));
  launch_layernorm(dx, d, dy, d, dg, db, rows, d, eps, C.stream);
  FE_HIP(hipMemcpyAsync(y, dy, bytes, hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// Developer hook: time one conv shape on random data (device-resident), `iters` launches, forced tile variant.
int fe_bench_conv(fe_ctx* ctx, int n, int h, int w, int cin, int cout, int k, int stride, int pad, int with_res, int act,
                  int variant, int iters, float* ms_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  C.arena.reset();
  DeviceWeights dw;
  WeightStore ws;
  std::vector<float> hw((size_t)cout * cin * k * k);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hw) v = rnd() * 0.1f;
  const int64_t wshape[4] = {cout, cin, k, k};
  ws.set("w.weight", hw.data(), wshape, 4);
  ConvW cw = build_conv(dw, ws, "w", "", false);
  std::vector<float> sc(cout, 1.01f), sh(cout, 0.1f);
  cw.scale = dw.upload(sc); cw.shift = dw.upload(sh);
  Tensor x = C.arena.tensor(n, h, w, cw.CinPad);
  {
    std::vector<float> hx((size_t)1 << 20);
    for (auto& v : hx) v = rnd();
    for (size_t off = 0; off < x.numel(); off += hx.size())
      FE_HIP(hipMemcpyAsync(x.p + off, hx.data(), std::min(hx.size(), x.numel() - off) * sizeof(float), hipMemcpyHostToDevice, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
  }
  ConvOpts o; o.sh = o.sw = stride; o.ph = o.pw = pad; o.act = act;
  Tensor y = C.arena.tensor(n, conv_out_dim(h, k, stride, pad, 1), conv_out_dim(w, k, stride, pad, 1), cout);
  Tensor r;
  if (with_res) { r = C.arena.tensor(y.n, y.h, y.w, y.c); FE_HIP(hipMemsetAsync(r.p, 0, r.numel() * sizeof(float), C.stream)); o.res = &r; }
  C.force_variant = variant;
  conv_forward(C, cw, x, y, o);  // warm
  FE_HIP(hipEventRecord(ctx->t0, C.stream));
  for (int i = 0; i < iters; ++i) conv_forward(C, cw, x, y, o);
  FE_HIP(hipEventRecord(ctx->t1, C.stream));
  FE_HIP(hipEventSynchronize(ctx->t1));
  C.force_variant = 0;
  float ms = 0.f;
  FE_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
  *ms_out = ms / iters;
  FE_API_END(ctx)
}

// ---- TOPIQ ------------------------------------------------------------------------------------------
static const float kImagenetMean[3] = {0.485f, 0.456f, 0.406f};
static const float kImagenetStd[3] = {0.229f, 0.224f, 0.225f};

// Runs the backbone on images [i0, i0+nb) of a device-resident u8 batch.
// Long edge > 1024 is first reduced with PIL-exact LANCZOS to (int(w*s), int(h*s)), s = 1024/long_edge, exactly as
// PyIQAScorer._preprocess_image does on the host (reference models/pyiqa_scorer.py:131-153).
extern "C++" {
template <class T>
static void topiq_backbone_chunk(fe_ctx* ctx, const uint8_t* d_rgb, int nb, int h, int w, std::vector<TensorT<T>>& feats) {
  Ctx& C = ctx->c;
  const int long_edge = h > w ? h : w;
  if (long_edge > 1024) {
    const double sc = 1024.0 / long_edge;
    const int nw = (int)(w * sc), nh = (int)(h * sc);
    uint8_t* small = (uint8_t*)C.arena.alloc((size_t)nb * nh * nw * 3);
    resize_u8(C, d_rgb, nb, h, w, nh, nw, FE_FILTER_LANCZOS, 0, nh, 0, nw, small);
    d_rgb = small; h = nh; w = nw;
  }
  Tensor x = C.arena.tensor(nb, h, w, 4);
  launch_u8_to_nhwc4_norm(d_rgb, x.p, (size_t)nb * h * w, kImagenetMean, kImagenetStd, 0, C.stream);
  resnet_forward<T>(C, ctx->c.topiq->backbone, x, &feats, ctx->c.topiq->dw.res32);      // RES32: fp32 skip stream in the backbone
}
// backbone + head of one micro-batch in the precision the model was committed under; scores are fp32 either way
// fe_topiq_f32_below: images with fewer pixels run on the model's fp32 weights even when it was committed under a 2-byte precision.
// With a few dozen tokens per pyramid level the rounding noise of a 2-byte pass is not averaged down (fp16 TOPIQ on 33 x 500 and
// 97 x 131 inputs: 4e-4 .. 1.2e-3 from the oracle, against <= 6e-4 from 512 x 512 up, tests/test_precision_policy_gpu.py), and such an
// image costs under a sixteenth of a 1024 x 1024 one. The PARITY policy (facet_amd/precision.py) sets 256 x 256; the default is 0.
// (Every 2-byte model keeps its fp32 weights: pack_conv only drops them for half_only models, which TOPIQ is not.)
static void topiq_chunk_score(fe_ctx* ctx, const uint8_t* d_in, int nb, int h, int w, float* d_scores) {
  Ctx& C = ctx->c;
  const bool small = (size_t)h * (size_t)w < C.topiq_f32_below && !C.topiq->dw.half_only && !C.topiq->dw.res32;
  if (C.topiq->dw.prec != PREC_F32 && small) {
    std::vector<Tensor> feats;
    topiq_backbone_chunk<float>(ctx, d_in, nb, h, w, feats);
    topiq_head_forward<float>(C, *C.topiq, feats, d_scores);
  } else if (C.topiq->dw.prec == PREC_BF16) {
    std::vector<TensorH> feats;
    topiq_backbone_chunk<bf16>(ctx, d_in, nb, h, w, feats);
    topiq_head_forward<bf16>(C, *C.topiq, feats, d_scores);
  } else if (C.topiq->dw.prec == PREC_F16) {
    std::vector<TensorF16> feats;
    topiq_backbone_chunk<f16>(ctx, d_in, nb, h, w, feats);
    topiq_head_forward<f16>(C, *C.topiq, feats, d_scores);
  } else {
    std::vector<Tensor> feats;
    topiq_backbone_chunk<float>(ctx, d_in, nb, h, w, feats);
    topiq_head_forward<float>(C, *C.topiq, feats, d_scores);
  }
}
}  // extern "C++"

int fe_topiq_feature_shape(int h, int w, int level, int dims[3]) {
  if (!dims || h < 32 || w < 32 || level < 0 || level > 4) return FE_ERR_INVALID;
  const int long_edge = h > w ? h : w;
  if (long_edge > 1024) {   // the LANCZOS cap of PyIQAScorer._preprocess_image, as topiq_backbone_chunk applies it
    const double sc = 1024.0 / long_edge;
    w = (int)(w * sc); h = (int)(h * sc);
  }
  static const int ch[5] = {64, 256, 512, 1024, 2048};
  int fh = conv_out_dim(h, 7, 2, 3, 1), fw = conv_out_dim(w, 7, 2, 3, 1);              // stem 7x7 / 2
  if (level >= 1) { fh = conv_out_dim(fh, 3, 2, 1, 1); fw = conv_out_dim(fw, 3, 2, 1, 1); }   // max pool 3x3 / 2
  for (int l = 2; l <= level; ++l) { fh = conv_out_dim(fh, 3, 2, 1, 1); fw = conv_out_dim(fw, 3, 2, 1, 1); }   // stride-2 3x3 of layer l
  dims[0] = ch[level]; dims[1] = fh; dims[2] = fw;
  return FE_OK;
}

int fe_topiq_features(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, int level, float* out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.topiq) { C.err = "topiq weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(rgb && out && n > 0 && h >= 32 && w >= 32 && level >= 0 && level <= 4, "bad arguments");
  const size_t img_bytes = (size_t)h * w * 3;
  size_t out_per_img = 0;
  for (int i0 = 0; i0 < n; i0 += ctx->microbatch) {
    const int nb = std::min(ctx->microbatch, n - i0);
    C.arena.reset();
    const uint8_t* d_in;
    if (on_device) {
      d_in = rgb + (size_t)i0 * img_bytes;
    } else {
      uint8_t* d = (uint8_t*)C.arena.alloc(nb * img_bytes);
      FE_HIP(hipMemcpyAsync(d, rgb + (size_t)i0 * img_bytes, nb * img_bytes, hipMemcpyHostToDevice, C.stream));
      d_in = d;
    }
    if (C.topiq->dw.prec == PREC_BF16) {
      std::vector<TensorH> feats;
      topiq_backbone_chunk<bf16>(ctx, d_in, nb, h, w, feats);
      const TensorH& f = feats[level];
      out_per_img = (size_t)f.c * f.h * f.w;
      download_nchw(C, f, f.c, out + (size_t)i0 * out_per_img);
    } else if (C.topiq->dw.prec == PREC_F16) {
      std::vector<TensorF16> feats;
      topiq_backbone_chunk<f16>(ctx, d_in, nb, h, w, feats);
      const TensorF16& f = feats[level];
      out_per_img = (size_t)f.c * f.h * f.w;
      download_nchw(C, f, f.c, out + (size_t)i0 * out_per_img);
    } else {
      std::vector<Tensor> feats;
      topiq_backbone_chunk<float>(ctx, d_in, nb, h, w, feats);
      const Tensor& f = feats[level];
      out_per_img = (size_t)f.c * f.h * f.w;
      download_nchw(C, f, f.c, out + (size_t)i0 * out_per_img);
    }
  }
  FE_API_END(ctx)
}

int fe_topiq_score(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* scores) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.topiq || !C.topiq->has_head) { C.err = "topiq weights (backbone + head) not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(rgb && scores && n > 0 && h >= 32 && w >= 32, "bad arguments");
  const size_t img_bytes = (size_t)h * w * 3;
  // scores of all micro-batches accumulate in a small device buffer outside the arena; one D2H at the end
  float* d_scores = ctx->out_buf((size_t)n);
  {
    ImageStager st(ctx, rgb, n, img_bytes, ctx->microbatch, on_device);
    for (int k = 0; k < st.chunks(); ++k) {
      const int i0 = k * ctx->microbatch, nb = st.count(k);
      C.arena.reset();
      const uint8_t* d_in = st.get(k);
      topiq_chunk_score(ctx, d_in, nb, h, w, d_scores + i0);
      st.done(k);
    }
    FE_HIP(hipMemcpyAsync(scores, d_scores, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
  }
  FE_API_END(ctx)
}

// ---- U2-Net-P + SAMP-Net ---------------------------------------------------------------------------
extern "C++" {
// saliency (and optionally the SAMP-Net scores) of one chunk of normalised fp32 NHWC4 crops, in the precision the models were
// committed under. d_sal (nullable): fp32 device [n][h][w] copy of the saliency map.
static void samp_chunk(fe_ctx* ctx, const Tensor& x, bool with_samp, float* pw, float* at, float* sd, float* d_sal) {
  Ctx& C = ctx->c;
  // the two networks may be committed under different precisions (the saliency map crosses in U2-Net-P's type)
  auto samp_on = [&](auto sal) {
    typedef decltype(sal.p) SP;
    typedef std::remove_pointer_t<SP> TS;
    if (!with_samp) return;
    if (C.samp->dw.prec == PREC_BF16) sampnet_forward<bf16, TS>(C, *C.samp, x, sal, pw, at, sd);
    else if (C.samp->dw.prec == PREC_F16) sampnet_forward<f16, TS>(C, *C.samp, x, sal, pw, at, sd);
    else sampnet_forward<float, TS>(C, *C.samp, x, sal, pw, at, sd);
  };
  const int prec = C.u2netp->dw.prec;
  if (prec == PREC_BF16) {
    TensorH sal = C.arena.tensor_t<bf16>(x.n, x.h, x.w, 1);
    u2netp_forward<bf16>(C, *C.u2netp, x, sal);
    samp_on(sal);
    if (d_sal) launch_convert(sal.p, d_sal, sal.numel(), C.stream);
  } else if (prec == PREC_F16) {
    TensorF16 sal = C.arena.tensor_t<f16>(x.n, x.h, x.w, 1);
    u2netp_forward<f16>(C, *C.u2netp, x, sal);
    samp_on(sal);
    if (d_sal) launch_convert(sal.p, d_sal, sal.numel(), C.stream);
  } else {
    Tensor sal = C.arena.tensor(x.n, x.h, x.w, 1);
    u2netp_forward<float>(C, *C.u2netp, x, sal);
    samp_on(sal);
    if (d_sal) FE_HIP(hipMemcpyAsync(d_sal, sal.p, sal.numel() * sizeof(float), hipMemcpyDeviceToDevice, C.stream));
  }
}
}  // extern "C++"
// x: host fp32 NCHW [n,3,h,w], already ImageNet-normalised (what SAMPNetScorer.preprocess yields, samp_net.py:904-928)
int fe_u2netp_saliency(fe_ctx* ctx, const float* x, int n, int h, int w, float* sal_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.u2netp) { C.err = "u2netp weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(x && sal_out && n > 0 && h >= 32 && w >= 32, "bad arguments");
  const size_t per = (size_t)3 * h * w;
  for (int i0 = 0; i0 < n; i0 += ctx->microbatch) {
    const int nb = std::min(ctx->microbatch, n - i0);
    C.arena.reset();
    Tensor xt = upload_nchw(C, x + (size_t)i0 * per, nb, 3, h, w, 4);
    float* d_sal = C.arena.array<float>((size_t)nb * h * w);
    samp_chunk(ctx, xt, false, nullptr, nullptr, nullptr, d_sal);
    FE_HIP(hipMemcpyAsync(sal_out + (size_t)i0 * h * w, d_sal, (size_t)nb * h * w * sizeof(float), hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
  }
  FE_API_END(ctx)
}

// SAMPNetScorer.score_batch's model part (samp_net.py:1005-1010): saliency = U2NETP(x); SAMPNet(x, saliency).
// Outputs (host): pattern_weights [n,8] (logits), attributes [n,6], score_dist [n,5]; sal_out optional [n,224,224].
int fe_samp_forward(fe_ctx* ctx, const float* x, int n, float* pattern_weights, float* attributes, float* score_dist,
                    float* sal_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.u2netp || !C.samp) { C.err = "samp_net / u2netp weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(x && n > 0 && pattern_weights && attributes && score_dist, "bad arguments");
  const int h = 224, w = 224;
  const size_t per = (size_t)3 * h * w;
  float* d_out = ctx->out_buf((size_t)n * 19);
  for (int i0 = 0; i0 < n; i0 += ctx->microbatch) {
    const int nb = std::min(ctx->microbatch, n - i0);
    C.arena.reset();
    Tensor xt = upload_nchw(C, x + (size_t)i0 * per, nb, 3, h, w, 4);
    float* d_sal = sal_out ? C.arena.array<float>((size_t)nb * h * w) : nullptr;
    samp_chunk(ctx, xt, true, d_out + (size_t)i0 * 8, d_out + (size_t)n * 8 + (size_t)i0 * 6, d_out + (size_t)n * 14 + (size_t)i0 * 5, d_sal);
    if (sal_out) FE_HIP(hipMemcpyAsync(sal_out + (size_t)i0 * h * w, d_sal, (size_t)nb * h * w * sizeof(float), hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
  }
  FE_HIP(hipMemcpyAsync(pattern_weights, d_out, (size_t)n * 8 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipMemcpyAsync(attributes, d_out + (size_t)n * 8, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipMemcpyAsync(score_dist, d_out + (size_t)n * 14, (size_t)n * 5 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// ---- CLIP ------------------------------------------------------------------------------------------
// x: fp32 NCHW [n,3,224,224] as open_clip's eval transform yields (host, or device when on_device).
// features [n,768] un-normalised (= model.encode_image); emb_norm (nullable) = F.normalize(features);
// aesthetic_raw (nullable, needs FE_MODEL_AESTHETIC) = aesthetic_head(features) before the (x+1)*5 clamp.
// The ViT tower wants more images per launch than the 1024^2 models can hold in flight: its GEMMs have rows = images x 257
// tokens in 128-row tiles x (width / 128) column tiles over 256 CUs, and a partially filled last round of workgroups costs up
// to a third of a launch. tools/clip_mb_sweep.py: 621 img/s at 32 images per launch, ~700 at 95-127. So crops (602 KB each)
// are collected across micro-batches and the tower runs on `chunk` of them, chunk chosen for full rounds.
extern "C++" {
// Moves the `left` crops behind the `c` just consumed to the front of a batcher buffer. left can exceed c (micro-batch larger than
// the tower chunk), where one copy would have overlapping source and destination ranges: the move is cut into pieces of at most c
// crops, each with disjoint ranges, issued in ascending order on the one stream.
static void compact_crops(float* buf, size_t per, int c, int left, hipStream_t s) {
  for (int done = 0; done < left; done += c) {
    const int n = std::min(c, left - done);
    FE_HIP(hipMemcpyAsync(buf + (size_t)done * per, buf + (size_t)(c + done) * per, (size_t)n * per * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
}
static int clip_tower_chunk(const ClipModel& m, int n) {
  if (n <= 40) return n;
  const int hi = std::min(n, 128), lo = std::max(32, hi - 40), ntile = std::max(1, m.width / 128);
  int best = hi;
  double best_eff = 0.0;
  for (int c = hi; c >= lo; --c) {
    const long wgs = (((long)c * m.tokens + 127) / 128) * ntile;
    const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
  }
  return best;
}
static void clip_tower(Ctx& C, const Tensor& x, float* feat) {   // in the precision the tower was committed under
  const bool r32 = C.clip->dw.res32;      // fp32 token stream around the 2-byte GEMMs
  if (C.clip->split3) { clip_forward_split3(C, *C.clip, x, feat); return; }
  if (C.clip->dw.prec == PREC_BF16) { if (r32) clip_forward<bf16, float>(C, *C.clip, x, feat); else clip_forward<bf16>(C, *C.clip, x, feat); }
  else if (C.clip->dw.prec == PREC_F16) { if (r32) clip_forward<f16, float>(C, *C.clip, x, feat); else clip_forward<f16>(C, *C.clip, x, feat); }
  else clip_forward<float>(C, *C.clip, x, feat);
}
class ClipBatcher {
 public:
  ClipBatcher(fe_ctx* ctx, int n_total, int max_push, float* d_feat, float* d_norm, float* d_aes)
      : x_(ctx), feat_(d_feat), norm_(d_norm), aes_(d_aes) {
    const ClipModel& m = *ctx->c.clip;
    hw_ = m.patch_size * (int)std::lround(std::sqrt((double)(m.tokens - 1)));
    od_ = m.out_dim;
    per_ = (size_t)hw_ * hw_ * 4;
    chunk_ = clip_tower_chunk(m, n_total);
    const size_t need = (size_t)(chunk_ + max_push) * per_;
    if (ctx->clip_in_cap < need) {
      FE_HIP(hipStreamSynchronize(ctx->c.stream));
      if (ctx->clip_in) FE_HIP(hipFree(ctx->clip_in));
      ctx->clip_in = nullptr; ctx->clip_in_cap = 0;
      FE_HIP(hipMalloc((void**)&ctx->clip_in, need * sizeof(float)));
      ctx->clip_in_cap = need;
    }
  }
  // xt: dense NHWC4 crops of one micro-batch (arena memory; copied out before the arena is recycled)
  void push(const Tensor& xt) {
    FE_CHECK(xt.c == 4 && xt.ld == 4 && xt.h == hw_ && xt.w == hw_, "clip batcher: crop layout");
    FE_HIP(hipMemcpyAsync(x_->clip_in + (size_t)count_ * per_, xt.p, (size_t)xt.n * per_ * sizeof(float), hipMemcpyDeviceToDevice, x_->c.stream));
    count_ += xt.n;
    while (count_ >= chunk_) run(chunk_);
  }
  void finish() {
    while (count_ > 0) run(std::min(count_, chunk_));
  }
 private:
  void run(int c) {
    Ctx& C = x_->c;
    const size_t mark = C.arena.mark();
    Tensor x;
    x.p = x_->clip_in; x.n = c; x.h = hw_; x.w = hw_; x.c = 4; x.ld = 4;
    clip_tower(C, x, feat_ + (size_t)done_ * od_);
    if (norm_) l2_normalize(C, feat_ + (size_t)done_ * od_, norm_ + (size_t)done_ * od_, c, od_);
    if (aes_) aesthetic_forward(C, *C.aesthetic, feat_ + (size_t)done_ * od_, c, aes_ + done_);
    C.arena.rewind(mark);
    const int left = count_ - c;
    compact_crops(x_->clip_in, per_, c, left, C.stream);
    done_ += c;
    count_ = left;
  }
  fe_ctx* x_;
  float *feat_, *norm_, *aes_;
  int hw_ = 224, od_ = 768, chunk_ = 1, count_ = 0, done_ = 0;
  size_t per_ = 0;
};
// SAMP-Net + U2-Net-P see 224^2 crops too, and most of their ~130 convolutions run on 7x7 .. 56x56 maps: at 32 images per launch
// they are launch- and tile-quantisation-bound (tools/samp_mb_sweep.py: 2216 img/s at 32 per launch, 2903 at 128). Crops are
// collected across micro-batches like the CLIP ones; the chunk is bounded by what the arena can hold (~150 MB per image).
class SampBatcher {
 public:
  SampBatcher(fe_ctx* ctx, int n_total, int max_push, float* pw, float* at, float* sd) : x_(ctx), pw_(pw), at_(at), sd_(sd) {
    const size_t room = ctx->c.arena.capacity() > ((size_t)8 << 30) ? ctx->c.arena.capacity() - ((size_t)8 << 30) : ctx->c.arena.capacity() / 4;
    const int fit = (int)std::min<size_t>(128, std::max<size_t>(1, room / ((size_t)150 << 20)));
    chunk_ = std::max(1, std::min(n_total, std::max(fit, std::min(max_push, 32))));
    per_ = (size_t)224 * 224 * 4;
    const size_t need = (size_t)(chunk_ + max_push) * per_;
    if (ctx->samp_in_cap < need) {
      FE_HIP(hipStreamSynchronize(ctx->c.stream));
      if (ctx->samp_in) FE_HIP(hipFree(ctx->samp_in));
      ctx->samp_in = nullptr; ctx->samp_in_cap = 0;
      FE_HIP(hipMalloc((void**)&ctx->samp_in, need * sizeof(float)));
      ctx->samp_in_cap = need;
    }
  }
  void push(const Tensor& xt) {
    FE_CHECK(xt.c == 4 && xt.ld == 4 && xt.h == 224 && xt.w == 224, "samp batcher: crop layout");
    FE_HIP(hipMemcpyAsync(x_->samp_in + (size_t)count_ * per_, xt.p, (size_t)xt.n * per_ * sizeof(float), hipMemcpyDeviceToDevice, x_->c.stream));
    count_ += xt.n;
    while (count_ >= chunk_) run(chunk_);
  }
  void finish() {
    while (count_ > 0) run(std::min(count_, chunk_));
  }
 private:
  void run(int c) {
    Ctx& C = x_->c;
    const size_t mark = C.arena.mark();
    Tensor x;
    x.p = x_->samp_in; x.n = c; x.h = 224; x.w = 224; x.c = 4; x.ld = 4;
    samp_chunk(x_, x, true, pw_ + (size_t)done_ * 8, at_ + (size_t)done_ * 6, sd_ + (size_t)done_ * 5, nullptr);
    C.arena.rewind(mark);
    const int left = count_ - c;
    compact_crops(x_->samp_in, per_, c, left, C.stream);
    done_ += c;
    count_ = left;
  }
  fe_ctx* x_;
  float *pw_, *at_, *sd_;
  int chunk_ = 1, count_ = 0, done_ = 0;
  size_t per_ = 0;
};
}  // extern "C++"

/* aesthetic_head on given feature / embedding vectors (reference Facet.score_from_embedding, processing/scorer.py:619-629) */
int fe_aesthetic_score(fe_ctx* ctx, const float* feats, int n, float* aesthetic_raw) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.aesthetic) { C.err = "aesthetic head weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(feats && aesthetic_raw && n > 0, "bad arguments");
  const int d = 768;
  for (int i0 = 0; i0 < n; i0 += 65536) {
    const int nb = std::min(65536, n - i0);
    C.arena.reset();
    float* d_in = (float*)C.arena.alloc((size_t)nb * d * sizeof(float));
    float* d_out = (float*)C.arena.alloc((size_t)nb * sizeof(float));
    FE_HIP(hipMemcpyAsync(d_in, feats + (size_t)i0 * d, (size_t)nb * d * sizeof(float), hipMemcpyHostToDevice, C.stream));
    aesthetic_forward(C, *C.aesthetic, d_in, nb, d_out);
    FE_HIP(hipMemcpyAsync(aesthetic_raw + i0, d_out, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
  }
  FE_API_END(ctx)
}

int fe_clip_encode_image(fe_ctx* ctx, const float* x, int n, int on_device, float* features, float* emb_norm,
                         float* aesthetic_raw) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.clip) { C.err = "clip weights not loaded"; return FE_ERR_NOT_LOADED; }
  if (aesthetic_raw && !C.aesthetic) { C.err = "aesthetic head weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(x && n > 0 && (features || emb_norm || aesthetic_raw), "bad arguments");
  const int hw = C.clip->patch_size * (int)std::lround(std::sqrt((double)(C.clip->tokens - 1)));
  const int od = C.clip->out_dim;
  const size_t per = (size_t)3 * hw * hw;
  float* d_out = ctx->out_buf((size_t)n * (2 * od + 1));
  float* d_feat = d_out; float* d_norm = d_out + (size_t)n * od; float* d_aes = d_out + (size_t)n * 2 * od;
  const int step = clip_tower_chunk(*C.clip, n);   // inputs are already 224^2: batch the tower for full rounds of workgroups
  for (int i0 = 0; i0 < n; i0 += step) {
    const int nb = std::min(step, n - i0);
    C.arena.reset();
    Tensor xt;
    if (on_device) {
      xt = C.arena.tensor(nb, hw, hw, 4);
      launch_nchw_to_nhwc(x + (size_t)i0 * per, xt.p, nb, 3, hw, hw, 4, C.stream);
    } else {
      xt = upload_nchw(C, x + (size_t)i0 * per, nb, 3, hw, hw, 4);
    }
    clip_tower(C, xt, d_feat + (size_t)i0 * od);
    if (emb_norm) l2_normalize(C, d_feat + (size_t)i0 * od, d_norm + (size_t)i0 * od, nb, od);
    if (aesthetic_raw) aesthetic_forward(C, *C.aesthetic, d_feat + (size_t)i0 * od, nb, d_aes + i0);
  }
  if (features) FE_HIP(hipMemcpyAsync(features, d_feat, (size_t)n * od * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  if (emb_norm) FE_HIP(hipMemcpyAsync(emb_norm, d_norm, (size_t)n * od * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  if (aesthetic_raw) FE_HIP(hipMemcpyAsync(aesthetic_raw, d_aes, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// ---- PIL-exact resize + image-level entry points ---------------------------------------------------------
static const float kClipMean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
static const float kClipStd[3] = {0.26862954f, 0.26130258f, 0.27577711f};

static int py_round_half_even(double v) {
  const double f = std::floor(v);
  const double d = v - f;
  if (d > 0.5) return (int)f + 1;
  if (d < 0.5) return (int)f;
  return ((long long)f % 2 == 0) ? (int)f : (int)f + 1;
}

// device u8 batch -> device u8 batch resized like PIL (+ crop)
int fe_resize_u8(fe_ctx* ctx, const uint8_t* src, int n, int h, int w, int oh, int ow, int filter, int on_device,
                 uint8_t* dst) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(src && dst && n > 0, "bad arguments");
  C.arena.reset();
  const size_t in_b = (size_t)n * h * w * 3, out_b = (size_t)n * oh * ow * 3;
  const uint8_t* d_in = src;
  uint8_t* d_out = dst;
  if (!on_device) {
    uint8_t* t = (uint8_t*)C.arena.alloc(in_b);
    FE_HIP(hipMemcpyAsync(t, src, in_b, hipMemcpyHostToDevice, C.stream));
    d_in = t;
    d_out = (uint8_t*)C.arena.alloc(out_b);
  }
  resize_u8(C, d_in, n, h, w, oh, ow, filter, 0, oh, 0, ow, d_out);
  if (!on_device) FE_HIP(hipMemcpyAsync(dst, d_out, out_b, hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// uint8 images -> the model's normalised NHWC4 input, preprocessing exactly like the reference's PIL/torchvision path
static Tensor preprocess_square224(Ctx& C, const uint8_t* d_rgb, int nb, int h, int w, int filter, bool shorter_side_crop,
                                   const float mean[3], const float stdv[3], int bgr) {
  int oh = 224, ow = 224, y0 = 0, x0 = 0;
  if (shorter_side_crop) {  // torchvision Resize(224) + CenterCrop(224)
    if (w <= h) { ow = 224; oh = (int)(224.0 * h / w); } else { oh = 224; ow = (int)(224.0 * w / h); }
    y0 = py_round_half_even((oh - 224) / 2.0);
    x0 = py_round_half_even((ow - 224) / 2.0);
  }
  uint8_t* small = (uint8_t*)C.arena.alloc((size_t)nb * 224 * 224 * 3);
  resize_u8(C, d_rgb, nb, h, w, oh, ow, filter, y0, 224, x0, 224, small);
  Tensor x = C.arena.tensor(nb, 224, 224, 4);
  launch_u8_to_nhwc4_norm(small, x.p, (size_t)nb * 224 * 224, mean, stdv, bgr, C.stream);
  return x;
}

// CLIP from raw images: open_clip eval transform (bicubic shorter-side 224, center crop, CLIP mean/std) + tower.
int fe_clip_encode_images(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* features,
                          float* emb_norm, float* aesthetic_raw) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.clip) { C.err = "clip weights not loaded"; return FE_ERR_NOT_LOADED; }
  if (aesthetic_raw && !C.aesthetic) { C.err = "aesthetic head weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(rgb && n > 0 && h > 0 && w > 0, "bad arguments");
  const int od = C.clip->out_dim;
  const size_t per = (size_t)h * w * 3;
  float* d_out = ctx->out_buf((size_t)n * (2 * od + 1));
  float* d_feat = d_out; float* d_norm = d_out + (size_t)n * od; float* d_aes = d_out + (size_t)n * 2 * od;
  ClipBatcher tower(ctx, n, ctx->microbatch, d_feat, emb_norm ? d_norm : nullptr, aesthetic_raw ? d_aes : nullptr);
  ImageStager st(ctx, rgb, n, per, ctx->microbatch, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int i0 = k * ctx->microbatch, nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    Tensor xt = preprocess_square224(C, d_in, nb, h, w, FE_FILTER_BICUBIC, true, kClipMean, kClipStd, 0);
    st.done(k);   // the raw images are consumed by the resize kernels queued above
    (void)i0;
    tower.push(xt);
  }
  tower.finish();
  if (features) FE_HIP(hipMemcpyAsync(features, d_feat, (size_t)n * od * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  if (emb_norm) FE_HIP(hipMemcpyAsync(emb_norm, d_norm, (size_t)n * od * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  if (aesthetic_raw) FE_HIP(hipMemcpyAsync(aesthetic_raw, d_aes, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// SAMPNetScorer.score_batch from raw images (samp_net.py:904-928, 991-1010): BGR->RGB if bgr, PIL bilinear
// Resize((224,224)), ToTensor, ImageNet Normalize, U2NETP saliency, SAMPNet.
int fe_samp_score_images(fe_ctx* ctx, const uint8_t* img, int n, int h, int w, int bgr, int on_device,
                         float* pattern_weights, float* attributes, float* score_dist) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.u2netp || !C.samp) { C.err = "samp_net / u2netp weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(img && n > 0 && pattern_weights && attributes && score_dist, "bad arguments");
  const size_t per = (size_t)h * w * 3;
  float* d_out = ctx->out_buf((size_t)n * 19);
  SampBatcher batch(ctx, n, ctx->microbatch, d_out, d_out + (size_t)n * 8, d_out + (size_t)n * 14);
  ImageStager st(ctx, img, n, per, ctx->microbatch, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    Tensor xt = preprocess_square224(C, d_in, nb, h, w, FE_FILTER_BILINEAR, false, kImagenetMean, kImagenetStd, bgr);
    st.done(k);
    batch.push(xt);
  }
  batch.finish();
  FE_HIP(hipMemcpyAsync(pattern_weights, d_out, (size_t)n * 8 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipMemcpyAsync(attributes, d_out + (size_t)n * 8, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipMemcpyAsync(score_dist, d_out + (size_t)n * 14, (size_t)n * 5 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// CLIP text tower: tokens int32 [n][ctx_len] (host) -> un-normalised text features [n][768].
// Reference: `self.model.encode_text(text_tokens)` in CLIPTagger._precompute_text_embeddings (models/tagger.py:69-75).
int fe_clip_encode_text(fe_ctx* ctx, const int32_t* tokens, int n, int ctx_len, float* features) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  if (!C.clip_text) { C.err = "clip text tower not loaded (checkpoint had no token_embedding.weight)"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(tokens && features && n > 0 && ctx_len == C.clip_text->ctx, "bad arguments (context length must be %d)", C.clip_text->ctx);
  const int od = C.clip_text->out_dim;
  std::vector<int> eot(n);
  for (int b = 0; b < n; ++b) {   // text.argmax(dim=-1): first position of the largest token id (the EOT token)
    int best = 0;
    for (int t = 1; t < ctx_len; ++t)
      if (tokens[(size_t)b * ctx_len + t] > tokens[(size_t)b * ctx_len + best]) best = t;
    eot[b] = best;
  }
  float* d_out = ctx->out_buf((size_t)n * od);
  const int mb = std::max(1, ctx->microbatch * 4);
  for (int i0 = 0; i0 < n; i0 += mb) {
    const int nb = std::min(mb, n - i0);
    C.arena.reset();
    int* d_tok = (int*)C.arena.alloc((size_t)nb * ctx_len * sizeof(int));
    int* d_eot = (int*)C.arena.alloc((size_t)nb * sizeof(int));
    FE_HIP(hipMemcpyAsync(d_tok, tokens + (size_t)i0 * ctx_len, (size_t)nb * ctx_len * sizeof(int), hipMemcpyHostToDevice, C.stream));
    FE_HIP(hipMemcpyAsync(d_eot, eot.data() + i0, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, C.stream));
    clip_text_forward(C, *C.clip_text, d_tok, d_eot, nb, d_out + (size_t)i0 * od);
    FE_HIP(hipStreamSynchronize(C.stream));
  }
  FE_HIP(hipMemcpyAsync(features, d_out, (size_t)n * od * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// Batched zero-shot tag scoring: sims[n][T] = emb[n][d] . text[T][d]^T on the matrix cores (both host, row-major).
// Replaces the per-image `image_features @ text_embeddings.T` + python loop of models/tagger.py:100-106.
int fe_tag_similarities(fe_ctx* ctx, const float* emb, int n, const float* text, int T, int d, float* sims) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(emb && text && sims && n > 0 && T > 0 && d > 0 && d % 4 == 0, "bad arguments");
  C.arena.reset();
  DeviceWeights dw;
  HostTensor w; w.shape = {T, d}; w.data.assign(text, text + (size_t)T * d);
  ConvW tw = build_linear_rows(dw, w, nullptr, 0, T);
  float* d_e = (float*)C.arena.alloc((size_t)n * tw.CinPad * sizeof(float));
  float* d_s = (float*)C.arena.alloc((size_t)n * T * sizeof(float));
  FE_HIP(hipMemcpyAsync(d_e, emb, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice, C.stream));
  linear_forward(C, tw, d_e, d, n, d_s, T, ACT_NONE);
  FE_HIP(hipMemcpyAsync(sims, d_s, (size_t)n * T * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// ---- VLM tagger: text decoder of Qwen2.5-VL (models/vlm_tagger.py:163-184 load, :250-259 / :355-360 greedy generate) ----------------
int fe_vlm_vision_configure(fe_ctx* ctx, int n_heads, const int* fullatt_block_indexes, int n_fullatt) {
  FE_API_BEGIN(ctx)
  FE_CHECK(n_heads > 0 && n_fullatt >= 0 && n_fullatt <= 8 && (n_fullatt == 0 || fullatt_block_indexes), "vlm_vision_configure: bad arguments (at most 8 full-attention blocks)");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  VlmConfig& g = ctx->c.vlm_cfg;
  g.vis_heads = n_heads; g.n_fullatt = n_fullatt;
  for (int i = 0; i < n_fullatt; ++i) g.fullatt[i] = fullatt_block_indexes[i];
  FE_API_END(ctx)
}
int fe_vlm_encode_images(fe_ctx* ctx, const float* pixel_values, int n_patches, const int32_t* patch_pos_hw, const int32_t* window_index, const int32_t* cu_window_seqlens,
                         int n_windows, const int32_t* cu_seqlens, int n_images, float* embeds) {
  FE_API_BEGIN(ctx)
  if (!ctx->c.vlm || !ctx->c.vlm->vis.present) { ctx->c.err = "vlm vision tower not loaded (checkpoint had no model.visual.* tensors)"; return FE_ERR_NOT_LOADED; }
  Ctx& C = ctx->c;
  VlmModel& m = *C.vlm;
  FE_CHECK(pixel_values && patch_pos_hw && window_index && cu_window_seqlens && cu_seqlens && n_patches > 0 && n_patches % 4 == 0 && n_windows > 0 && n_images > 0,
           "bad arguments");
  FE_CHECK(cu_window_seqlens[0] == 0 && cu_window_seqlens[n_windows] == n_patches && cu_seqlens[0] == 0 && cu_seqlens[n_images] == n_patches, "segment bounds must cover the patches");
  int max_win = 0, max_full = 0;
  for (int i = 0; i < n_windows; ++i) { FE_CHECK(cu_window_seqlens[i + 1] > cu_window_seqlens[i], "empty window segment"); max_win = std::max(max_win, cu_window_seqlens[i + 1] - cu_window_seqlens[i]); }
  for (int i = 0; i < n_images; ++i) { FE_CHECK(cu_seqlens[i + 1] > cu_seqlens[i], "empty image segment"); max_full = std::max(max_full, cu_seqlens[i + 1] - cu_seqlens[i]); }
  for (int i = 0; i < n_patches / 4; ++i) FE_CHECK(window_index[i] >= 0 && window_index[i] < n_patches / 4, "window_index out of range");
  const int rows = n_patches / 4;
  if (rows > m.img_cap) {
    if (m.img_embeds) (void)hipFree(m.img_embeds);
    m.img_embeds = nullptr; m.img_cap = 0;
    FE_HIP(hipMalloc((void**)&m.img_embeds, (size_t)rows * m.hidden * sizeof(bf16)));
    m.img_cap = rows;
  }
  C.arena.reset();
  float* d_pv = (float*)C.arena.alloc((size_t)n_patches * m.vis.patch_dim * sizeof(float));
  int* d_pos = (int*)C.arena.alloc((size_t)n_patches * 2 * sizeof(int));
  int* d_widx = (int*)C.arena.alloc((size_t)rows * sizeof(int));
  int* d_cw = (int*)C.arena.alloc((size_t)(n_windows + 1) * sizeof(int));
  int* d_cf = (int*)C.arena.alloc((size_t)(n_images + 1) * sizeof(int));
  FE_HIP(hipMemcpyAsync(d_pv, pixel_values, (size_t)n_patches * m.vis.patch_dim * sizeof(float), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_pos, patch_pos_hw, (size_t)n_patches * 2 * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_widx, window_index, (size_t)rows * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_cw, cu_window_seqlens, (size_t)(n_windows + 1) * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_cf, cu_seqlens, (size_t)(n_images + 1) * sizeof(int), hipMemcpyHostToDevice, C.stream));
  vlm_vision_forward(C, m, d_pv, n_patches, d_pos, d_widx, d_cw, n_windows, max_win, d_cf, n_images, max_full, m.img_embeds);
  m.img_rows = rows;
  if (embeds) {
    float* d_f = (float*)C.arena.alloc((size_t)rows * m.hidden * sizeof(float));
    launch_convert((const bf16*)m.img_embeds, d_f, (size_t)rows * m.hidden, C.stream);
    FE_HIP(hipMemcpyAsync(embeds, d_f, (size_t)rows * m.hidden * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  }
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}
int fe_vlm_configure(fe_ctx* ctx, int n_heads, int n_kv_heads, int head_dim, float rope_theta, float rms_eps, const int* mrope_section) {
  FE_API_BEGIN(ctx)
  FE_CHECK(n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && head_dim == 128 && rope_theta > 0.f && rms_eps > 0.f && mrope_section,
           "vlm_configure: bad geometry (head_dim must be 128)");
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  VlmConfig& g = ctx->c.vlm_cfg;
  g.n_heads = n_heads; g.n_kv_heads = n_kv_heads; g.head_dim = head_dim; g.rope_theta = rope_theta; g.rms_eps = rms_eps;
  for (int i = 0; i < 3; ++i) g.mrope[i] = mrope_section[i];
  FE_API_END(ctx)
}
int fe_vlm_dims(fe_ctx* ctx, int* dims) {
  FE_API_BEGIN(ctx)
  if (!ctx->c.vlm) { ctx->c.err = "vlm weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(dims, "bad arguments");
  const VlmModel& m = *ctx->c.vlm;
  dims[0] = m.vocab; dims[1] = m.hidden; dims[2] = (int)m.layers.size(); dims[3] = m.cfg.n_heads; dims[4] = m.cfg.n_kv_heads; dims[5] = m.inter;
  dims[6] = m.max_seq; dims[7] = m.cur_len;
  FE_API_END(ctx)
}
extern "C++" {
// tokens (+ optional replacement rows for image tokens) -> embeddings -> decoder -> next tokens; shared by prefill and decode
static void vlm_step(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int len, int32_t* next_tokens, float* logits,
                     const int32_t* image_rows = nullptr, int n_image_rows = 0) {
  Ctx& C = ctx->c;
  VlmModel& m = *C.vlm;
  const int rows = n_seq * len;
  C.arena.reset();
  int* d_tok = (int*)C.arena.alloc((size_t)rows * sizeof(int));
  int* d_pos = (int*)C.arena.alloc((size_t)3 * rows * sizeof(int));
  int* d_next = (int*)C.arena.alloc((size_t)n_seq * sizeof(int));
  float* d_logits = logits ? (float*)C.arena.alloc((size_t)n_seq * m.vocab * sizeof(float)) : nullptr;
  bf16* x = C.arena.array<bf16>((size_t)rows * m.hidden);
  FE_HIP(hipMemcpyAsync(d_tok, tokens, (size_t)rows * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_pos, position_ids, (size_t)3 * rows * sizeof(int), hipMemcpyHostToDevice, C.stream));
  vlm_embed(C, m, d_tok, rows, x);
  if (n_image_rows > 0) {      // inputs_embeds.masked_scatter(image_mask, image_embeds): the merged image embeddings replace the placeholder rows, in order
    int* d_idx = (int*)C.arena.alloc((size_t)n_image_rows * sizeof(int));
    FE_HIP(hipMemcpyAsync(d_idx, image_rows, (size_t)n_image_rows * sizeof(int), hipMemcpyHostToDevice, C.stream));
    vlm_put_rows(C, x, m.img_embeds, d_idx, n_image_rows, m.hidden);
  }
  vlm_forward(C, m, x, d_pos, n_seq, len, d_next, d_logits);
  FE_HIP(hipMemcpyAsync(next_tokens, d_next, (size_t)n_seq * sizeof(int), hipMemcpyDeviceToHost, C.stream));
  if (logits) FE_HIP(hipMemcpyAsync(logits, d_logits, (size_t)n_seq * m.vocab * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
}
}  // extern "C++"
int fe_vlm_prefill(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int len, int max_seq, int32_t* next_tokens, float* logits) {
  FE_API_BEGIN(ctx)
  if (!ctx->c.vlm) { ctx->c.err = "vlm weights not loaded"; return FE_ERR_NOT_LOADED; }
  FE_CHECK(tokens && position_ids && next_tokens && n_seq > 0 && len > 0 && max_seq >= len && max_seq <= 8192, "bad arguments (max_seq <= 8192)");
  ctx->c.vlm->reserve_cache(n_seq, max_seq);
  ctx->c.vlm->cur_len = 0;
  vlm_step(ctx, tokens, position_ids, n_seq, len, next_tokens, logits);
  FE_API_END(ctx)
}
int fe_vlm_prefill_images(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int len, int max_seq, const int32_t* image_rows, int n_image_rows,
                          int32_t* next_tokens, float* logits) {
  FE_API_BEGIN(ctx)
  if (!ctx->c.vlm) { ctx->c.err = "vlm weights not loaded"; return FE_ERR_NOT_LOADED; }
  VlmModel& m = *ctx->c.vlm;
  FE_CHECK(tokens && position_ids && next_tokens && n_seq > 0 && len > 0 && max_seq >= len && max_seq <= 8192, "bad arguments (max_seq <= 8192)");
  FE_CHECK(n_image_rows == 0 || (image_rows && n_image_rows == m.img_rows), "prefill_images: %d placeholder rows but the last fe_vlm_encode_images left %d embeddings",
           n_image_rows, m.img_rows);
  for (int i = 0; i < n_image_rows; ++i) FE_CHECK(image_rows[i] >= 0 && image_rows[i] < n_seq * len, "prefill_images: row index out of range");
  m.reserve_cache(n_seq, max_seq);
  m.cur_len = 0;
  vlm_step(ctx, tokens, position_ids, n_seq, len, next_tokens, logits, image_rows, n_image_rows);
  FE_API_END(ctx)
}
int fe_vlm_generate(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int n_steps, int32_t* out_tokens) {
  FE_API_BEGIN(ctx)
  if (!ctx->c.vlm) { ctx->c.err = "vlm weights not loaded"; return FE_ERR_NOT_LOADED; }
  Ctx& C = ctx->c;
  VlmModel& m = *C.vlm;
  FE_CHECK(tokens && position_ids && out_tokens && n_steps > 0 && n_seq == m.cache_B && m.cur_len > 0, "generate: call fe_vlm_prefill for these %d sequences first", n_seq);
  C.arena.reset();
  int* d_tok = (int*)C.arena.alloc((size_t)n_seq * sizeof(int));
  int* d_pos = (int*)C.arena.alloc((size_t)3 * n_seq * sizeof(int));
  int* d_out = (int*)C.arena.alloc((size_t)n_steps * n_seq * sizeof(int));
  FE_HIP(hipMemcpyAsync(d_tok, tokens, (size_t)n_seq * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_pos, position_ids, (size_t)3 * n_seq * sizeof(int), hipMemcpyHostToDevice, C.stream));
  vlm_decode_steps(C, m, d_tok, d_pos, n_seq, n_steps, d_out);
  FE_HIP(hipMemcpyAsync(out_tokens, d_out, (size_t)n_steps * n_seq * sizeof(int), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}
int fe_vlm_decode_step(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int32_t* next_tokens, float* logits) {
  FE_API_BEGIN(ctx)
  if (!ctx->c.vlm) { ctx->c.err = "vlm weights not loaded"; return FE_ERR_NOT_LOADED; }
  VlmModel& m = *ctx->c.vlm;
  FE_CHECK(tokens && position_ids && next_tokens && n_seq == m.cache_B && m.cur_len > 0, "decode_step: call fe_vlm_prefill for these %d sequences first", n_seq);
  FE_CHECK(m.cur_len < m.max_seq, "decode_step: the KV cache is full (%d positions)", m.max_seq);
  vlm_step(ctx, tokens, position_ids, n_seq, 1, next_tokens, logits);
  FE_API_END(ctx)
}

// One call per batch for the whole ensemble (what processing/batch_processor.py:169-360 sequences per image):
// record[i] = [topiq_raw, aesthetic_raw, pattern_weights(8), attributes(6), score_dist(5), clip_emb_norm(768)] = 789 floats.
// Models that are not loaded leave their fields at 0 (mask bit i of *models_run: 1 topiq, 2 clip, 4 samp).
extern "C++" {
// SoA result planes (what the model heads write) -> [n][ld] records, one thread per record float.
__global__ void records_interleave_kernel(const float* __restrict__ planes, size_t n4, int n, float* __restrict__ rec, int ld) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * FE_RECORD_FLOATS) return;
  const int img = (int)(i / FE_RECORD_FLOATS), f = (int)(i - (size_t)img * FE_RECORD_FLOATS);
  const size_t o_aes = n4, o_pw = 2 * n4, o_at = o_pw + 8 * n4, o_sd = o_at + 6 * n4, o_emb = o_sd + 5 * n4;
  float v;
  if (f == 0) v = planes[img];
  else if (f == 1) v = planes[o_aes + img];
  else if (f < 10) v = planes[o_pw + (size_t)img * 8 + (f - 2)];
  else if (f < 16) v = planes[o_at + (size_t)img * 6 + (f - 10)];
  else if (f < 21) v = planes[o_sd + (size_t)img * 5 + (f - 16)];
  else v = planes[o_emb + (size_t)img * 768 + (f - 21)];
  rec[(size_t)img * ld + f] = v;
}

// Runs every selected model over the batch and leaves the interleaved records in device memory d_rec [n][ld].
static int ensemble_run(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* d_rec, int ld) {
  Ctx& C = ctx->c;
  const size_t per = (size_t)h * w * 3;
  // SoA planes on the device (every plane 16-B aligned), interleaved into records by a last small kernel
  const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
  const size_t o_aes = n4, o_pw = 2 * n4, o_at = o_pw + 8 * n4, o_sd = o_at + 6 * n4, o_emb = o_sd + 5 * n4,
               o_feat = o_emb + 768 * n4, total = o_feat + 768 * n4;
  float* d_pl = ctx->out_buf(total);
  FE_HIP(hipMemsetAsync(d_pl, 0, total * sizeof(float), C.stream));
  const int sel = ctx->ensemble_mask;
  const bool do_topiq = (sel & 1) && C.topiq && C.topiq->has_head, do_clip = (sel & 2) && C.clip, do_samp = (sel & 4) && C.samp && C.u2netp;
  float* p_topiq = d_pl;  float* p_aes = d_pl + o_aes;  float* p_pw = d_pl + o_pw;  float* p_at = d_pl + o_at;
  float* p_sd = d_pl + o_sd;  float* p_emb = d_pl + o_emb;  float* d_feat = d_pl + o_feat;
  std::unique_ptr<ClipBatcher> tower;
  if (do_clip) tower = std::make_unique<ClipBatcher>(ctx, n, ctx->microbatch, d_feat, p_emb, C.aesthetic ? p_aes : nullptr);
  std::unique_ptr<SampBatcher> samp;
  if (do_samp) samp = std::make_unique<SampBatcher>(ctx, n, ctx->microbatch, p_pw, p_at, p_sd);
  ImageStager st(ctx, rgb, n, per, ctx->microbatch, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int i0 = k * ctx->microbatch, nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    if (do_topiq) {
      const size_t mark = C.arena.mark();
      topiq_chunk_score(ctx, d_in, nb, h, w, p_topiq + i0);
      C.arena.rewind(mark);
    }
    if (do_clip) {
      const size_t mark = C.arena.mark();
      Tensor xt = preprocess_square224(C, d_in, nb, h, w, FE_FILTER_BICUBIC, true, kClipMean, kClipStd, 0);
      tower->push(xt);   // the ViT tower runs once enough crops have gathered for full rounds of workgroups
      C.arena.rewind(mark);
    }
    if (do_samp) {
      const size_t mark = C.arena.mark();
      Tensor xt = preprocess_square224(C, d_in, nb, h, w, FE_FILTER_BILINEAR, false, kImagenetMean, kImagenetStd, 0);
      samp->push(xt);    // U2-Net-P + SAMP-Net run once enough crops have gathered
      C.arena.rewind(mark);
    }
    st.done(k);
  }
  if (tower) tower->finish();
  if (samp) samp->finish();
  const size_t work = (size_t)n * FE_RECORD_FLOATS;
  hipLaunchKernelGGL(records_interleave_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, C.stream, d_pl, n4, n, d_rec, ld);
  FE_HIP(hipGetLastError());
  return (do_topiq ? 1 : 0) | (do_clip ? 2 : 0) | (do_samp ? 4 : 0);
}
}  // extern "C++"

int fe_ensemble_select(fe_ctx* ctx, int models) {
  FE_API_BEGIN(ctx)
  FE_CHECK(models > 0 && models <= 7, "ensemble_select: mask %d (1 topiq | 2 clip | 4 samp)", models);
  ctx->ensemble_mask = models;
  FE_API_END(ctx)
}

int fe_ensemble_score(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* records, int* models_run) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(rgb && records && n > 0 && h >= 32 && w >= 32, "bad arguments");
  const size_t floats = (size_t)n * FE_RECORD_FLOATS;
  if (floats > ctx->d_rec_cap) {
    if (ctx->d_rec) FE_HIP(hipFree(ctx->d_rec));
    ctx->d_rec = nullptr; ctx->d_rec_cap = 0;
    FE_HIP(hipMalloc((void**)&ctx->d_rec, floats * sizeof(float)));
    ctx->d_rec_cap = floats;
  }
  const int ran = ensemble_run(ctx, rgb, n, h, w, on_device, ctx->d_rec, FE_RECORD_FLOATS);
  FE_HIP(hipMemcpyAsync(records, ctx->d_rec, floats * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  if (models_run) *models_run = ran;
  FE_API_END(ctx)
}

// Same, with the records left in DEVICE memory: d_records [n][ld_records] floats (ld_records >= FE_RECORD_FLOATS; the columns past
// 789 are not touched, so a caller can keep its face slots beside them). Returns after the engine stream has drained, so the
// buffer can be handed to a collective on another stream (the multi-GPU all-gather reads it in place: facet_amd/sharding.py).
int fe_ensemble_score_dev(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* d_records, int ld_records,
                          int* models_run) {
  FE_API_BEGIN(ctx)
  FE_CHECK(rgb && d_records && n > 0 && h >= 32 && w >= 32 && ld_records >= FE_RECORD_FLOATS, "bad arguments");
  const int ran = ensemble_run(ctx, rgb, n, h, w, on_device, d_records, ld_records);
  FE_HIP(hipStreamSynchronize(ctx->c.stream));
  if (models_run) *models_run = ran;
  FE_API_END(ctx)
}

// ---- face path (InsightFace FaceAnalysis: detection -> landmark_2d_106 -> recognition) -------------------------------
static fe_ctx::CvResizeTab cv_resize_tab(fe_ctx* ctx, int src, int dst, bool clamp) {
  auto key = std::make_tuple(src, dst, clamp ? 1 : 0);
  auto it = ctx->cvresize.find(key);
  if (it != ctx->cvresize.end()) return it->second;
  std::vector<int> ofs;
  std::vector<short> coef;
  cv_resize_tables(src, dst, clamp, ofs, coef);
  fe_ctx::CvResizeTab t{};
  FE_HIP(hipMalloc((void**)&t.ofs, ofs.size() * sizeof(int)));
  ctx->misc_allocs.push_back(t.ofs);
  FE_HIP(hipMalloc((void**)&t.coef, coef.size() * sizeof(short)));
  ctx->misc_allocs.push_back(t.coef);
  FE_HIP(hipMemcpy(t.ofs, ofs.data(), ofs.size() * sizeof(int), hipMemcpyHostToDevice));
  FE_HIP(hipMemcpy(t.coef, coef.data(), coef.size() * sizeof(short), hipMemcpyHostToDevice));
  return ctx->cvresize[key] = t;
}

int fe_face_detect(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int det_h, int det_w, float thresh,
                   int max_cand, float* cand, int* counts, float* det_scale_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  GraphSlot& gs = graph_slot(ctx, FE_GRAPH_FACE_DET);
  FE_CHECK(bgr && cand && counts && n > 0 && h > 0 && w > 0 && det_h >= 32 && det_w >= 32 && det_h % 32 == 0 && det_w % 32 == 0 &&
               max_cand > 0, "bad arguments");
  // SCRFD.detect: keep the aspect ratio, fill the top-left of the det canvas (insightface model_zoo/scrfd.py [DEP-KNOWLEDGE])
  const float im_ratio = (float)h / (float)w, model_ratio = (float)det_h / (float)det_w;
  int new_h, new_w;
  if (im_ratio > model_ratio) { new_h = det_h; new_w = (int)((float)new_h / im_ratio); }
  else { new_w = det_w; new_h = (int)((float)new_w * im_ratio); }
  FE_CHECK(new_h > 0 && new_w > 0, "image aspect ratio leaves an empty detector input");
  const float det_scale = (float)new_h / (float)h;
  if (det_scale_out) *det_scale_out = det_scale;
  const bool area2 = (h == 2 * new_h && w == 2 * new_w);
  const bool copy_only = (h == new_h && w == new_w);
  auto tx = cv_resize_tab(ctx, w, new_w, true), ty = cv_resize_tab(ctx, h, new_h, false);

  const size_t no = gs.g.model().outputs.size();
  int fmc, K = 0, A;
  if (no == 6) { fmc = 3; A = 2; }
  else if (no == 9) { fmc = 3; A = 2; K = 5; }
  else if (no == 10) { fmc = 5; A = 1; }
  else if (no == 15) { fmc = 5; A = 1; K = 5; }
  else FE_CHECK(false, "detector graph has %zu outputs; SCRFD layouts have 6, 9, 10 or 15", no);
  static const int kStrides3[3] = {8, 16, 32}, kStrides5[5] = {8, 16, 32, 64, 128};
  const int* strides = fmc == 3 ? kStrides3 : kStrides5;

  const size_t per = (size_t)h * w * 3;
  float* d_cand = ctx->out_buf((size_t)n * max_cand * 16 + (size_t)n + 16);
  int* d_counts = (int*)(d_cand + (size_t)n * max_cand * 16);
  FE_HIP(hipMemsetAsync(d_counts, 0, (size_t)n * sizeof(int), C.stream));
  const int mbn = ctx->microbatch * 2;   // see fe_face_analyze
  ImageStager st(ctx, bgr, n, per, mbn, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int i0 = k * mbn, nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    uint8_t* canvas = (uint8_t*)C.arena.alloc((size_t)nb * det_h * det_w * 3);
    FE_HIP(hipMemsetAsync(canvas, 0, (size_t)nb * det_h * det_w * 3, C.stream));
    if (copy_only) {
      for (int b = 0; b < nb; ++b)
        FE_HIP(hipMemcpy2DAsync(canvas + (size_t)b * det_h * det_w * 3, (size_t)det_w * 3, d_in + (size_t)b * per, (size_t)w * 3, (size_t)w * 3, h,
                                hipMemcpyDeviceToDevice, C.stream));
    } else {
      launch_cv_resize_linear(d_in, nb, h, w, canvas, det_h, det_w, new_h, new_w, tx.ofs, tx.coef, ty.ofs, ty.coef, area2 ? 1 : 0, C.stream);
    }
    st.done(k);
    Tensor x = C.arena.tensor(nb, det_h, det_w, 4);
    launch_u8_blob(canvas, x.p, (size_t)nb * det_h * det_w, 127.5f, 1.0f / 128.0f, 1, C.stream);
    std::vector<GraphOutput> outs;
    gs.g.run(C, x, 3, outs);
    for (int l = 0; l < fmc; ++l) {
      const int s = strides[l], fh = det_h / s, fw = det_w / s;
      const size_t rows = (size_t)nb * fh * fw * A;
      FE_CHECK(outs[l].numel == rows && outs[l + fmc].numel == rows * 4 && (!K || outs[l + 2 * fmc].numel == rows * 2 * K),
               "detector output %d has %zu values, expected %zu rows for stride %d", l, outs[l].numel, rows, s);
      launch_scrfd_decode(outs[l].dev, outs[l + fmc].dev, K ? outs[l + 2 * fmc].dev : nullptr, nb, fh, fw, A, K, s, thresh, det_scale, l,
                          d_cand + (size_t)i0 * max_cand * 16, d_counts + i0, max_cand, C.stream);
    }
  }
  FE_HIP(hipMemcpyAsync(counts, d_counts, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipMemcpyAsync(cand, d_cand, (size_t)n * max_cand * 16 * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// Warps `m` crops out of device-resident images and (optionally) runs graph `gs` on them. d_out: device [m][out_dim] or null;
// crops_out: host [m][size][size][3] or null. Does not reset the arena; allocates above the caller's mark.
static void run_face_crops(fe_ctx* ctx, GraphSlot* gs, const uint8_t* d_img, int n, int h, int w, int m, const int* img_index,
                           const double* M, int size, float mean, float scale, int swap_rb, float* d_out, int out_dim, uint8_t* crops_out) {
  Ctx& C = ctx->c;
  if (m <= 0) return;
  if (!ctx->warp_wtab) {
    std::vector<short> wt;
    cv_warp_weight_table(wt);
    FE_HIP(hipMalloc((void**)&ctx->warp_wtab, wt.size() * sizeof(short)));
    ctx->misc_allocs.push_back(ctx->warp_wtab);
    FE_HIP(hipMemcpy(ctx->warp_wtab, wt.data(), wt.size() * sizeof(short), hipMemcpyHostToDevice));
  }
  // cv::warpAffine inverts the forward matrix in double before walking the destination
  std::vector<double> inv((size_t)m * 6);
  for (int f = 0; f < m; ++f) {
    FE_CHECK(img_index[f] >= 0 && img_index[f] < n, "crop %d refers to image %d of %d", f, img_index[f], n);
    const double* a = M + (size_t)f * 6;
    double D = a[0] * a[4] - a[1] * a[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = a[4] * D, A22 = a[0] * D;
    double* o = &inv[(size_t)f * 6];
    o[0] = A11; o[1] = a[1] * (-D); o[3] = a[3] * (-D); o[4] = A22;
    o[2] = -o[0] * a[2] - o[1] * a[5];
    o[5] = -o[3] * a[2] - o[4] * a[5];
  }
  double* d_inv = (double*)C.arena.alloc(inv.size() * sizeof(double));
  int* d_idx = (int*)C.arena.alloc((size_t)m * sizeof(int));
  FE_HIP(hipMemcpyAsync(d_inv, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_idx, img_index, (size_t)m * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));   // inv is a local; the copies above must finish before it goes away
  const size_t base = C.arena.mark();
  const int mb = std::max(1, ctx->microbatch * 8);   // crops are small (112^2 / 192^2): large batches fill the chip
  for (int f0 = 0; f0 < m; f0 += mb) {
    const int fb = std::min(mb, m - f0);
    C.arena.rewind(base);
    uint8_t* crops = (uint8_t*)C.arena.alloc((size_t)fb * size * size * 3);
    launch_warp_affine(d_img, h, w, d_idx + f0, d_inv + (size_t)f0 * 6, fb, size, ctx->warp_wtab, crops, C.stream);
    if (crops_out)
      FE_HIP(hipMemcpyAsync(crops_out + (size_t)f0 * size * size * 3, crops, (size_t)fb * size * size * 3, hipMemcpyDeviceToHost, C.stream));
    if (d_out) {
      Tensor x = C.arena.tensor(fb, size, size, 4);
      launch_u8_blob(crops, x.p, (size_t)fb * size * size, mean, scale, swap_rb, C.stream);
      std::vector<GraphOutput> outs;
      gs->g.run(C, x, 3, outs);
      FE_CHECK(outs[0].numel == (size_t)fb * out_dim, "graph output has %zu values for %d crops, expected %d each", outs[0].numel, fb, out_dim);
      FE_HIP(hipMemcpyAsync(d_out + (size_t)f0 * out_dim, outs[0].dev, outs[0].numel * sizeof(float), hipMemcpyDeviceToDevice, C.stream));
    }
  }
}

int fe_face_crops_run(fe_ctx* ctx, int slot, const uint8_t* bgr, int n, int h, int w, int on_device, int m, const int* img_index,
                      const double* M, int size, float mean, float scale, int swap_rb, float* out, int out_dim, uint8_t* crops_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(bgr && n > 0 && h > 0 && w > 0 && m >= 0 && size > 0 && (m == 0 || (img_index && M)), "bad arguments");
  FE_CHECK(out || crops_out, "nothing to compute: both outputs are null");
  GraphSlot* gs = out ? &graph_slot(ctx, slot) : nullptr;
  if (m == 0) return FE_OK;
  C.arena.reset();
  const uint8_t* d_img = bgr;
  if (!on_device) {
    uint8_t* d = (uint8_t*)C.arena.alloc((size_t)n * h * w * 3);
    FE_HIP(hipMemcpyAsync(d, bgr, (size_t)n * h * w * 3, hipMemcpyHostToDevice, C.stream));
    d_img = d;
  }
  float* d_out = out ? ctx->out_buf((size_t)m * out_dim) : nullptr;
  run_face_crops(ctx, gs, d_img, n, h, w, m, img_index, M, size, mean, scale, swap_rb, d_out, out_dim, crops_out);
  if (out) FE_HIP(hipMemcpyAsync(out, d_out, (size_t)m * out_dim * sizeof(float), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

// ---- FaceAnalysis.get for a whole batch, host glue in C++ ---------------------------------------------------------------
extern "C++" {
namespace {
struct Cand { float v[16]; };
// insightface SCRFD.nms on rows sorted by score (fp32 arithmetic like numpy's): returns kept indices in score order
std::vector<int> face_nms(const std::vector<Cand>& c, float thresh) {
  const int n = (int)c.size();
  std::vector<float> area(n);
  for (int i = 0; i < n; ++i) area[i] = (c[i].v[3] - c[i].v[1] + 1.f) * (c[i].v[4] - c[i].v[2] + 1.f);
  std::vector<char> dead(n, 0);
  std::vector<int> keep;
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep.push_back(i);
    for (int j = i + 1; j < n; ++j) {
      if (dead[j]) continue;
      const float xx1 = std::max(c[i].v[1], c[j].v[1]), yy1 = std::max(c[i].v[2], c[j].v[2]);
      const float xx2 = std::min(c[i].v[3], c[j].v[3]), yy2 = std::min(c[i].v[4], c[j].v[4]);
      const float ww = std::max(0.0f, xx2 - xx1 + 1.f), hh = std::max(0.0f, yy2 - yy1 + 1.f);
      const float inter = ww * hh;
      const float ovr = inter / (area[i] + area[j] - inter);
      if (!(ovr <= thresh)) dead[j] = 1;
    }
  }
  return keep;
}
// least-squares similarity src(5 pts, fp32) -> dst, closed form of Umeyama in 2-D (double); false when degenerate
bool similarity5(const float* src, const double* dst, double* M) {
  double sm[2] = {0, 0}, dm[2] = {0, 0};
  for (int k = 0; k < 5; ++k) { sm[0] += src[2 * k]; sm[1] += src[2 * k + 1]; dm[0] += dst[2 * k]; dm[1] += dst[2 * k + 1]; }
  for (int a = 0; a < 2; ++a) { sm[a] /= 5.0; dm[a] /= 5.0; }
  double A[2][2] = {{0, 0}, {0, 0}}, var = 0;
  for (int k = 0; k < 5; ++k) {
    const double sx = src[2 * k] - sm[0], sy = src[2 * k + 1] - sm[1], dx = dst[2 * k] - dm[0], dy = dst[2 * k + 1] - dm[1];
    A[0][0] += dx * sx; A[0][1] += dx * sy; A[1][0] += dy * sx; A[1][1] += dy * sy;
    var += sx * sx + sy * sy;
  }
  for (auto& r : A) for (auto& v : r) v /= 5.0;
  var /= 5.0;
  const double p = A[0][0] + A[1][1], q = A[1][0] - A[0][1], r = std::hypot(p, q);
  if (r == 0.0 || var == 0.0) return false;
  const double sc = r / var, c = p / r * sc, s = q / r * sc;
  M[0] = c; M[1] = -s; M[2] = dm[0] - (c * sm[0] - s * sm[1]);
  M[3] = s; M[4] = c;  M[5] = dm[1] - (s * sm[0] + c * sm[1]);
  return true;
}
const float kArcfaceDst[10] = {38.2946f, 51.6963f, 73.5318f, 51.5014f, 56.0252f, 71.7366f, 41.5493f, 92.3655f, 70.7299f, 92.2041f};
}  // namespace
}  // extern "C++"

int fe_face_analyze(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int det_h, int det_w, float det_thresh,
                    float nms_thresh, int max_faces, float* faces, int* counts, int* models_run) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  GraphSlot& det = graph_slot(ctx, FE_GRAPH_FACE_DET);
  GraphSlot* lmk = ctx->c.graphs[FE_GRAPH_FACE_LMK].get();
  GraphSlot* rec = ctx->c.graphs[FE_GRAPH_FACE_REC].get();
  FE_CHECK(bgr && faces && counts && n > 0 && h > 0 && w > 0 && max_faces > 0 && det_h >= 32 && det_w >= 32 && det_h % 32 == 0 &&
               det_w % 32 == 0, "bad arguments");
  if (models_run) *models_run = 1 | (lmk ? 2 : 0) | (rec ? 4 : 0);
  const float im_ratio = (float)h / (float)w, model_ratio = (float)det_h / (float)det_w;
  int new_h, new_w;
  if (im_ratio > model_ratio) { new_h = det_h; new_w = (int)((float)new_h / im_ratio); }
  else { new_w = det_w; new_h = (int)((float)new_w * im_ratio); }
  FE_CHECK(new_h > 0 && new_w > 0, "image aspect ratio leaves an empty detector input");
  const float det_scale = (float)new_h / (float)h;
  const bool area2 = (h == 2 * new_h && w == 2 * new_w), copy_only = (h == new_h && w == new_w);
  auto tx = cv_resize_tab(ctx, w, new_w, true), ty = cv_resize_tab(ctx, h, new_h, false);
  const size_t no = det.g.model().outputs.size();
  int fmc, K = 0, A;
  if (no == 6) { fmc = 3; A = 2; }
  else if (no == 9) { fmc = 3; A = 2; K = 5; }
  else if (no == 10) { fmc = 5; A = 1; }
  else if (no == 15) { fmc = 5; A = 1; K = 5; }
  else FE_CHECK(false, "detector graph has %zu outputs; SCRFD layouts have 6, 9, 10 or 15", no);
  static const int kS3[3] = {8, 16, 32}, kS5[5] = {8, 16, 32, 64, 128};
  const int* strides = fmc == 3 ? kS3 : kS5;
  auto graph_norm = [](GraphSlot* g, float dflt_std, float* mean, float* scale, int* size, int dflt_size) {
    const bool self = g->g.head_has_sub() && g->g.head_has_mul();
    *mean = self ? 0.f : 127.5f;
    *scale = 1.0f / (self ? 1.0f : dflt_std);
    const auto& d = g->g.model().inputs[0].dims;
    *size = (d.size() == 4 && d[2] > 0) ? (int)d[2] : dflt_size;
  };
  float lm_mean = 0, lm_scale = 1, rc_mean = 0, rc_scale = 1;
  int lm_size = 192, rc_size = 112;
  if (lmk) graph_norm(lmk, 128.0f, &lm_mean, &lm_scale, &lm_size, 192);
  if (rec) graph_norm(rec, 127.5f, &rc_mean, &rc_scale, &rc_size, 112);
  const auto& lo = lmk ? lmk->g.model().outputs[0].dims : std::vector<int64_t>();
  const int lm_dim = (lmk && !lo.empty() && lo.back() > 0) ? (int)lo.back() : 212;
  FE_CHECK(!lmk || lm_dim == 212, "landmark graph yields %d values per face; the record layout holds 106 x 2", lm_dim);
  const auto& ro = rec ? rec->g.model().outputs[0].dims : std::vector<int64_t>();
  const int rc_dim = (rec && !ro.empty() && ro.back() > 0) ? (int)ro.back() : 512;
  FE_CHECK(!rec || rc_dim == 512, "recognition graph yields %d values per face; the record layout holds 512", rc_dim);

  const int max_cand = 4096;
  const size_t per = (size_t)h * w * 3;
  const int mbn = ctx->microbatch * 2;   // the detector's footprint at 640^2 is ~10x below TOPIQ's at 1024^2: larger chunks fill the chip better
  memset(faces, 0, (size_t)n * max_faces * FE_FACE_FLOATS * sizeof(float));
  std::vector<float> h_cand((size_t)mbn * max_cand * 16), h_lmk, h_emb;
  std::vector<int> h_counts(mbn);
  // device image access for the crop stage: resident input is used in place; host input is staged per micro-batch
  ImageStager st(ctx, bgr, n, per, mbn, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int i0 = k * mbn, nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    float* d_cand = (float*)C.arena.alloc((size_t)nb * max_cand * 16 * sizeof(float));
    int* d_counts = (int*)C.arena.alloc((size_t)nb * sizeof(int));
    FE_HIP(hipMemsetAsync(d_counts, 0, (size_t)nb * sizeof(int), C.stream));
    const size_t keep_mark = C.arena.mark();
    uint8_t* canvas = (uint8_t*)C.arena.alloc((size_t)nb * det_h * det_w * 3);
    FE_HIP(hipMemsetAsync(canvas, 0, (size_t)nb * det_h * det_w * 3, C.stream));
    if (copy_only) {
      for (int b = 0; b < nb; ++b)
        FE_HIP(hipMemcpy2DAsync(canvas + (size_t)b * det_h * det_w * 3, (size_t)det_w * 3, d_in + (size_t)b * per, (size_t)w * 3, (size_t)w * 3, h,
                                hipMemcpyDeviceToDevice, C.stream));
    } else {
      launch_cv_resize_linear(d_in, nb, h, w, canvas, det_h, det_w, new_h, new_w, tx.ofs, tx.coef, ty.ofs, ty.coef, area2 ? 1 : 0, C.stream);
    }
    Tensor x = C.arena.tensor(nb, det_h, det_w, 4);
    launch_u8_blob(canvas, x.p, (size_t)nb * det_h * det_w, 127.5f, 1.0f / 128.0f, 1, C.stream);
    std::vector<GraphOutput> outs;
    det.g.run(C, x, 3, outs);
    for (int l = 0; l < fmc; ++l) {
      const int s = strides[l], fh = det_h / s, fw = det_w / s;
      const size_t rows = (size_t)nb * fh * fw * A;
      FE_CHECK(outs[l].numel == rows && outs[l + fmc].numel == rows * 4 && (!K || outs[l + 2 * fmc].numel == rows * 2 * K),
               "detector output %d has %zu values, expected %zu rows for stride %d", l, outs[l].numel, rows, s);
      launch_scrfd_decode(outs[l].dev, outs[l + fmc].dev, K ? outs[l + 2 * fmc].dev : nullptr, nb, fh, fw, A, K, s, det_thresh, det_scale, l,
                          d_cand, d_counts, max_cand, C.stream);
    }
    FE_HIP(hipMemcpyAsync(h_counts.data(), d_counts, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
    int maxc = 0;
    for (int b = 0; b < nb; ++b) { h_counts[b] = std::min(h_counts[b], max_cand); maxc = std::max(maxc, h_counts[b]); }
    if (maxc > 0) {
      FE_HIP(hipMemcpy2DAsync(h_cand.data(), (size_t)maxc * 16 * sizeof(float), d_cand, (size_t)max_cand * 16 * sizeof(float),
                              (size_t)maxc * 16 * sizeof(float), nb, hipMemcpyDeviceToHost, C.stream));
      FE_HIP(hipStreamSynchronize(C.stream));
    }
    // host: sort by score (ties: level, x1, y1 - deterministic), NMS, keep the best max_faces
    std::vector<int> f_img;
    std::vector<Cand> f_c;
    for (int b = 0; b < nb; ++b) {
      std::vector<Cand> c(h_counts[b]);
      for (int q = 0; q < h_counts[b]; ++q) memcpy(c[q].v, &h_cand[((size_t)b * maxc + q) * 16], 16 * sizeof(float));
      std::sort(c.begin(), c.end(), [](const Cand& a, const Cand& b2) {
        if (a.v[0] != b2.v[0]) return a.v[0] > b2.v[0];
        if (a.v[15] != b2.v[15]) return a.v[15] < b2.v[15];
        if (a.v[1] != b2.v[1]) return a.v[1] < b2.v[1];
        return a.v[2] < b2.v[2];
      });
      std::vector<int> keep = face_nms(c, nms_thresh);
      counts[i0 + b] = (int)keep.size();
      for (int q = 0; q < (int)keep.size() && q < max_faces; ++q) { f_img.push_back(b); f_c.push_back(c[keep[q]]); }
    }
    const int m = (int)f_c.size();
    C.arena.rewind(keep_mark);   // detector activations are no longer needed; candidates were copied out
    float *d_lmk = nullptr, *d_emb = nullptr;
    std::vector<double> Ml((size_t)m * 6), Mr((size_t)m * 6);
    if (m > 0 && lmk) {
      for (int f = 0; f < m; ++f) {   // Landmark.get: face_align.transform(img, center, size, size / (max(w,h) * 1.5), 0)
        const double x1 = f_c[f].v[1], y1 = f_c[f].v[2], x2 = f_c[f].v[3], y2 = f_c[f].v[4];
        const double bw = x2 - x1, bh = y2 - y1, cx = (x2 + x1) / 2, cy = (y2 + y1) / 2;
        const double sc = lm_size / (std::max(bw, bh) * 1.5);
        double* M = &Ml[(size_t)f * 6];
        M[0] = sc; M[1] = 0; M[2] = -cx * sc + lm_size / 2.0;
        M[3] = 0; M[4] = sc; M[5] = -cy * sc + lm_size / 2.0;
      }
      d_lmk = (float*)C.arena.alloc((size_t)m * 212 * sizeof(float));
      run_face_crops(ctx, lmk, d_in, nb, h, w, m, f_img.data(), Ml.data(), lm_size, lm_mean, lm_scale, 1, d_lmk, 212, nullptr);
      h_lmk.resize((size_t)m * 212);
      FE_HIP(hipMemcpyAsync(h_lmk.data(), d_lmk, h_lmk.size() * sizeof(float), hipMemcpyDeviceToHost, C.stream));
    }
    std::vector<char> rec_ok(m, 0);
    if (m > 0 && rec && K == 5) {
      double dst[10];
      for (int q = 0; q < 10; ++q) dst[q] = (double)kArcfaceDst[q] * ((double)rc_size / 112.0);
      for (int f = 0; f < m; ++f) {
        rec_ok[f] = similarity5(&f_c[f].v[5], dst, &Mr[(size_t)f * 6]) ? 1 : 0;
        if (!rec_ok[f]) { double* M = &Mr[(size_t)f * 6]; M[0] = M[4] = 1; M[1] = M[2] = M[3] = M[5] = 0; }
      }
      const size_t mk = C.arena.mark();
      d_emb = (float*)C.arena.alloc((size_t)m * 512 * sizeof(float));
      run_face_crops(ctx, rec, d_in, nb, h, w, m, f_img.data(), Mr.data(), rc_size, rc_mean, rc_scale, 1, d_emb, 512, nullptr);
      h_emb.resize((size_t)m * 512);
      FE_HIP(hipMemcpyAsync(h_emb.data(), d_emb, h_emb.size() * sizeof(float), hipMemcpyDeviceToHost, C.stream));
      (void)mk;
    }
    FE_HIP(hipStreamSynchronize(C.stream));
    st.done(k);
    std::vector<int> slot_of(nb, 0);
    for (int f = 0; f < m; ++f) {
      const int b = f_img[f];
      float* o = faces + ((size_t)(i0 + b) * max_faces + slot_of[b]++) * FE_FACE_FLOATS;
      o[0] = f_c[f].v[1]; o[1] = f_c[f].v[2]; o[2] = f_c[f].v[3]; o[3] = f_c[f].v[4]; o[4] = f_c[f].v[0];
      memcpy(o + 5, &f_c[f].v[5], 10 * sizeof(float));
      if (d_lmk) {   // pred in [-1,1] -> crop pixels -> image through the inverse crop matrix (trans_points2d)
        const double* M = &Ml[(size_t)f * 6];
        const double D = 1.0 / (M[0] * M[4]);   // rotation 0: diagonal matrix
        const double i00 = M[4] * D, i11 = M[0] * D, i02 = -i00 * M[2], i12 = -i11 * M[5];
        for (int q = 0; q < 106; ++q) {
          const float px = (h_lmk[(size_t)f * 212 + 2 * q] + 1.f) * (float)(lm_size / 2);
          const float py = (h_lmk[(size_t)f * 212 + 2 * q + 1] + 1.f) * (float)(lm_size / 2);
          o[15 + 2 * q] = (float)(i00 * (double)px + 0.0 * (double)py + i02);
          o[16 + 2 * q] = (float)(0.0 * (double)px + i11 * (double)py + i12);
        }
      }
      if (d_emb && rec_ok[f]) memcpy(o + 15 + 212, &h_emb[(size_t)f * 512], 512 * sizeof(float));
    }
  }
  FE_API_END(ctx)
}

/* cv2.resize(img, (ow, oh)) INTER_LINEAR on uint8 HWC images (restated fixed-point path); exposed for the parity tests */
int fe_cv_resize_linear_u8(fe_ctx* ctx, const uint8_t* src, int n, int h, int w, int oh, int ow, uint8_t* dst) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(src && dst && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "bad arguments");
  C.arena.reset();
  uint8_t* d_src = (uint8_t*)C.arena.alloc((size_t)n * h * w * 3);
  uint8_t* d_dst = (uint8_t*)C.arena.alloc((size_t)n * oh * ow * 3);
  FE_HIP(hipMemcpyAsync(d_src, src, (size_t)n * h * w * 3, hipMemcpyHostToDevice, C.stream));
  auto tx = cv_resize_tab(ctx, w, ow, true), ty = cv_resize_tab(ctx, h, oh, false);
  if (h == oh && w == ow) FE_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)n * h * w * 3, hipMemcpyDeviceToDevice, C.stream));
  else launch_cv_resize_linear(d_src, n, h, w, d_dst, oh, ow, oh, ow, tx.ofs, tx.coef, ty.ofs, ty.coef, (h == 2 * oh && w == 2 * ow) ? 1 : 0, C.stream);
  FE_HIP(hipMemcpyAsync(dst, d_dst, (size_t)n * oh * ow * 3, hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

/* Per-image technical statistics of a BGR batch (reference analyzers/image_cache.py:28-33 + analyzers/technical.py) */
int fe_image_stats(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, double* stats, uint8_t* gray_out, uint8_t* hsv_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(bgr && stats && n > 0 && h > 0 && w > 0, "bad arguments");
  if (!ctx->hsv_sdiv) {
    std::vector<int> sd, hd;
    cv_hsv_tables(sd, hd);
    FE_HIP(hipMalloc((void**)&ctx->hsv_sdiv, 256 * sizeof(int)));
    ctx->misc_allocs.push_back(ctx->hsv_sdiv);
    FE_HIP(hipMalloc((void**)&ctx->hsv_hdiv, 256 * sizeof(int)));
    ctx->misc_allocs.push_back(ctx->hsv_hdiv);
    FE_HIP(hipMemcpy(ctx->hsv_sdiv, sd.data(), 256 * sizeof(int), hipMemcpyHostToDevice));
    FE_HIP(hipMemcpy(ctx->hsv_hdiv, hd.data(), 256 * sizeof(int), hipMemcpyHostToDevice));
  }
  const size_t per = (size_t)h * w * 3, npx = (size_t)h * w;
  // two blocks per image in pass 1: large chunks keep all 256 CUs busy (the footprint is only ~2-5 bytes per pixel)
  const int mb = std::max(1, std::max(ctx->microbatch, 256));
  ImageStager st(ctx, bgr, n, per, mb, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int i0 = k * mb, nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    uint8_t* d_gray = (uint8_t*)C.arena.alloc((size_t)nb * npx);
    uint8_t* d_hsv = hsv_out ? (uint8_t*)C.arena.alloc((size_t)nb * per) : nullptr;
    void* d_acc = C.arena.alloc(stats_accum_bytes(nb));
    double* d_out = (double*)C.arena.alloc((size_t)nb * FE_STATS_COUNT * sizeof(double));
    launch_image_stats(d_in, nb, h, w, d_gray, d_hsv, ctx->hsv_sdiv, ctx->hsv_hdiv, d_acc, d_out, C.stream);
    st.done(k);
    FE_HIP(hipMemcpyAsync(stats + (size_t)i0 * FE_STATS_COUNT, d_out, (size_t)nb * FE_STATS_COUNT * sizeof(double), hipMemcpyDeviceToHost, C.stream));
    if (gray_out) FE_HIP(hipMemcpyAsync(gray_out + (size_t)i0 * npx, d_gray, (size_t)nb * npx, hipMemcpyDeviceToHost, C.stream));
    if (hsv_out) FE_HIP(hipMemcpyAsync(hsv_out + (size_t)i0 * per, d_hsv, (size_t)nb * per, hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));   // the arena is recycled by the next micro-batch
  }
  FE_API_END(ctx)
}

/* Laplacian statistics of m rectangular ROIs of a BGR batch (reference analyzers/face.py:160-176, 272-279) */
int fe_roi_laplacian(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int m, const int* img_index, const int* rois,
                     double* out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(bgr && n > 0 && h > 0 && w > 0 && m >= 0 && (m == 0 || (img_index && rois && out)), "bad arguments");
  if (m == 0) return FE_OK;
  for (int f = 0; f < m; ++f) {
    FE_CHECK(img_index[f] >= 0 && img_index[f] < n, "roi %d refers to image %d of %d", f, img_index[f], n);
    const int* r = rois + 4 * f;
    FE_CHECK(r[0] >= 0 && r[1] >= 0 && r[2] <= w && r[3] <= h, "roi %d = [%d,%d,%d,%d] leaves the %dx%d image", f, r[0], r[1], r[2], r[3], w, h);
  }
  C.arena.reset();
  const uint8_t* d_img = bgr;
  if (!on_device) {
    uint8_t* d = (uint8_t*)C.arena.alloc((size_t)n * h * w * 3);
    FE_HIP(hipMemcpyAsync(d, bgr, (size_t)n * h * w * 3, hipMemcpyHostToDevice, C.stream));
    d_img = d;
  }
  int* d_idx = (int*)C.arena.alloc((size_t)m * sizeof(int));
  int* d_roi = (int*)C.arena.alloc((size_t)m * 4 * sizeof(int));
  double* d_out = (double*)C.arena.alloc((size_t)m * 4 * sizeof(double));
  FE_HIP(hipMemcpyAsync(d_idx, img_index, (size_t)m * sizeof(int), hipMemcpyHostToDevice, C.stream));
  FE_HIP(hipMemcpyAsync(d_roi, rois, (size_t)m * 4 * sizeof(int), hipMemcpyHostToDevice, C.stream));
  launch_roi_laplacian(d_img, h, w, d_idx, d_roi, m, d_out, C.stream);
  FE_HIP(hipMemcpyAsync(out, d_out, (size_t)m * 4 * sizeof(double), hipMemcpyDeviceToHost, C.stream));
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

/* RGB <-> BGR copy of a packed uint8 batch into device memory */
int fe_swap_rb_u8(fe_ctx* ctx, const uint8_t* src, int on_device, size_t pixels, uint8_t* dst_device) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(src && dst_device && pixels > 0, "bad arguments");
  if (on_device) {
    FE_CHECK(src != dst_device, "swap_rb: in-place is not supported");
    launch_swap_rb_u8(src, dst_device, pixels, C.stream);
  } else {                                   // stage through the destination: upload, then swap each pixel's ends in a second buffer-free pass
    C.arena.reset();
    uint8_t* tmp = (uint8_t*)C.arena.alloc(pixels * 3);
    FE_HIP(hipMemcpyAsync(tmp, src, pixels * 3, hipMemcpyHostToDevice, C.stream));
    launch_swap_rb_u8(tmp, dst_device, pixels, C.stream);
  }
  FE_HIP(hipStreamSynchronize(C.stream));
  FE_API_END(ctx)
}

/* Leading lines (reference analyzers/composition.py:191-261): blur + Canny map on the GPU, hysteresis + probabilistic Hough per image on host threads */
int fe_leading_lines(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int canny_low, int canny_high, int threshold,
                     int min_line_length, int max_line_gap, int max_lines, int* lines, int* counts, uint8_t* edges_out) {
  FE_API_BEGIN(ctx)
  Ctx& C = ctx->c;
  FE_CHECK(bgr && n > 0 && h > 0 && w > 0 && (size_t)h * w < (1ull << 30), "bad arguments");
  FE_CHECK((lines != nullptr) == (counts != nullptr) && (lines || edges_out), "pass lines AND counts, and / or edges_out");
  FE_CHECK(!lines || max_lines > 0, "max_lines must be positive");
  FE_CHECK(canny_low >= 0 && canny_high >= canny_low && threshold > 0 && min_line_length >= 0 && max_line_gap >= 0, "bad thresholds");
  const size_t npx = (size_t)h * w, per = npx * 3;
  const int mb = std::max(1, std::min(n, 64));
  const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<uint8_t> scratch;
  if (!edges_out) scratch.resize((size_t)mb * npx);
  ImageStager st(ctx, bgr, n, per, mb, on_device);
  for (int k = 0; k < st.chunks(); ++k) {
    const int i0 = k * mb, nb = st.count(k);
    C.arena.reset();
    const uint8_t* d_in = st.get(k);
    uint8_t* d_blur = (uint8_t*)C.arena.alloc((size_t)nb * npx);
    void* d_grad = C.arena.alloc((size_t)nb * npx * 4);
    void* d_mag = C.arena.alloc((size_t)nb * npx * 2);
    uint8_t* d_map = (uint8_t*)C.arena.alloc((size_t)nb * npx);
    launch_canny_map(d_in, nb, h, w, canny_low, canny_high, d_blur, d_grad, d_mag, d_map, C.stream);
    st.done(k);
    uint8_t* maps = edges_out ? edges_out + (size_t)i0 * npx : scratch.data();
    FE_HIP(hipMemcpyAsync(maps, d_map, (size_t)nb * npx, hipMemcpyDeviceToHost, C.stream));
    FE_HIP(hipStreamSynchronize(C.stream));
    lines_host_stage(maps, nb, h, w, threshold, min_line_length, max_line_gap, max_lines, lines ? lines + (size_t)i0 * max_lines * 4 : nullptr,
                     counts ? counts + i0 : nullptr, threads);
  }
  FE_API_END(ctx)
}

}  // extern "C"

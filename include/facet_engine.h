/*
 * facet_engine.h — C ABI of libfacet_engine.so, the MI355X (gfx950) image-scoring engine.
 *
 * The reference (rlorenzo/facet) has no FFI for this path: its model layer is duck-typed Python objects
 * that end in torch nn.Module.__call__ / onnxruntime sessions. Each entry point below
 * names the reference call it stands behind; the facet_amd Python package is the ctypes binding that re-exposes the
 * reference's Python signatures on top of it.
 *
 * Conventions
 *   - every function returns FE_OK (0) or a negative fe_status; fe_last_error(ctx) gives the message.
 *   - no C++ exceptions cross this boundary; no torch / numpy types appear in it.
 *   - the caller owns every input / output buffer for the duration of the call; the engine owns
 *     weights and workspace behind the opaque fe_ctx.
 *   - pointers are HOST pointers unless the parameter is named d_* or `on_device` is non-zero.
 *   - images are uint8 HWC, RGB unless a `bgr` flag says otherwise; float tensors are fp32, NCHW,
 *     contiguous (the layouts the reference hands to its models).
 *   - one thread at a time per ctx for fe_*_score / fe_*_encode / fe_op_* (the reference calls its
 *     models from one "GPU thread", processing/batch_processor.py:123); lifecycle calls are mutex-guarded.
 */
#ifndef FACET_ENGINE_H
#define FACET_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fe_ctx fe_ctx;

enum fe_status {
  FE_OK = 0,
  FE_ERR_INVALID = -1,   /* bad argument / shape */
  FE_ERR_RUNTIME = -2,   /* HIP failure or internal check */
  FE_ERR_NOT_LOADED = -3 /* model weights not committed */
};

/* Model slots (reference names: models/model_manager.py:393-437 'topiq','clip','samp_net',...). */
enum fe_model {
  FE_MODEL_TOPIQ = 0,     /* pyiqa topiq_nr: ResNet-50 + CFANet head   (models/pyiqa_scorer.py:33-39,212) */
  FE_MODEL_CLIP = 1,      /* open_clip ViT-L/14 image tower             (processing/scorer.py:508,662)     */
  FE_MODEL_SAMP = 2,      /* SAMPNet (ResNet-18 + pattern pooling)      (models/samp_net.py:665-791)       */
  FE_MODEL_U2NETP = 3,    /* U2-Net-P saliency                          (models/samp_net.py:258-342)       */
  FE_MODEL_AESTHETIC = 4, /* Linear(768,256)-ReLU-Linear(256,1)         (processing/scorer.py:579-583)     */
  FE_MODEL_VLM = 5        /* Qwen2.5-VL text decoder (VLM tagger)       (models/vlm_tagger.py:163-184)     */
};

enum fe_act { FE_ACT_NONE = 0, FE_ACT_RELU = 1, FE_ACT_GELU = 2, FE_ACT_SIGMOID = 3, FE_ACT_SOFTPLUS = 5 /* torch.nn.Softplus(beta=1, threshold=20) */ };

/* ---- lifecycle ------------------------------------------------------------------------------- */
/* Creates a context on HIP device `device` with a workspace arena of `arena_bytes`
 * (0 = a quarter of the free HBM, at most 64 GiB). Fails (no CPU fallback) when no gfx950 device is present. */
int fe_create(int device, size_t arena_bytes, fe_ctx** out);
void fe_destroy(fe_ctx* ctx);
const char* fe_last_error(fe_ctx* ctx); /* ctx may be NULL: returns the last fe_create error */
const char* fe_version(void);
int fe_sync(fe_ctx* ctx);
/* images processed per engine pass inside the batched entry points (activation footprint knob) */
int fe_set_microbatch(fe_ctx* ctx, int n);

/* Precision of the models committed AFTER this call (each model keeps the one it was committed under; FE_MODEL_AESTHETIC, the CLIP
 * text tower and the ONNX face graphs always run in fp32).
 *   FE_PRECISION_F32   default: the arithmetic of the reference's CPU path.
 *   FE_PRECISION_F16   what the reference itself runs on a GPU for CLIP (`self.model.half()`, processing/scorer.py:513-516): fp16
 *                      activations and weights in HBM on the matrix cores (v_mfma_f32_32x32x16_f16), fp32 accumulation, fp32
 *                      LayerNorm / softmax statistics, fp32 score heads and outputs; stores saturate at +-65504.
 *   FE_PRECISION_BF16  BASELINE.json configs[3]: the same with bf16 storage (v_mfma_f32_32x32x16_bf16; 8 significant bits against
 *                      fp16's 11, fp32's exponent range).
 *   | FE_PRECISION_RES32  (or-ed onto a 2-byte precision) the residual / skip streams of the network (the ViT token stream, the ResNet
 *                      skip path) and the inputs of every LayerNorm stay fp32; only the GEMM operands are 2 bytes.
 *   | FE_PRECISION_SPLIT3 (or-ed onto FE_PRECISION_F16; the CLIP image tower; implies RES32) split-operand fp16: every weight and every
 *                      GEMM operand LayerNorm / GELU produce is an fp16 pair hi + lo (~22 significant bits) and one launch over the
 *                      concatenated operands accumulates xh.Wh + xl.Wh + xh.Wl in fp32 - three times the matrix work of plain
 *                      fp16 (still a fraction of fp32's), results that hold the fp32 path's 1e-3 gate on the final scores.
 * CLIP's 14x14 patch embedding and the non-7x7 three-channel first layers stay on the fp32 kernels in every precision. */
enum fe_precision { FE_PRECISION_F32 = 0, FE_PRECISION_BF16 = 1, FE_PRECISION_F16 = 2, FE_PRECISION_RES32 = 16, FE_PRECISION_SPLIT3 = 32 };
int fe_set_precision(fe_ctx* ctx, int precision);
int fe_model_precision(fe_ctx* ctx, int model); /* enum fe_precision value of a loaded model, with its RES32 bit; -1 when it is not loaded */

/* ---- VLM tagger (BASELINE configs[4]), slice 1: the text decoder --------------------------------------------
 * Reference: models/vlm_tagger.py loads transformers' Qwen2_5_VLForConditionalGeneration in bfloat16 (:155-184) and calls
 * generate(**inputs, max_new_tokens=..., do_sample=False) (:250-259, :355-360). FE_MODEL_VLM takes that class's state dict (tensor
 * names model.language_model.* and lm_head.weight; model.visual.* is ignored in this slice) and always runs in bf16, with the
 * rounding points of the bf16 torch modules. Token ids in, token ids out: tokenizer, chat template and tag parsing stay on the host
 * (facet_amd/vlm_tagger.py). fe_vlm_configure gives the geometry the tensor shapes do not determine (defaults = Qwen2.5-VL-7B:
 * 28 heads, 4 KV heads, head_dim 128, rope_theta 1e6, rms eps 1e-6, mrope_section 16/24/24) and is read by the NEXT
 * fe_weights_commit(FE_MODEL_VLM).
 * fe_vlm_prefill: n_seq prompts of `len` tokens each (tokens [n_seq][len], position_ids [3][n_seq][len] = the temporal / height /
 * width rotary positions transformers' get_rope_index yields; all three equal for text tokens) fill a fresh KV cache of max_seq
 * positions per sequence (<= 8192) and return the greedy next token of every sequence (argmax of the bf16 logits, first index on
 * ties, as torch.argmax) and, when `logits` is not NULL, those logits [n_seq][vocab]. fe_vlm_decode_step appends one token per
 * sequence (tokens [n_seq], position_ids [3][n_seq]). dims: vocab, hidden, layers, heads, kv_heads, intermediate, max_seq, cur_len. */
int fe_vlm_configure(fe_ctx* ctx, int n_heads, int n_kv_heads, int head_dim, float rope_theta, float rms_eps, const int* mrope_section);
/* Vision tower (slice 2; built when the FE_MODEL_VLM checkpoint carries model.visual.*): geometry the tensor shapes do not determine
 * (defaults = Qwen2.5-VL-7B: 16 heads of 80, full attention in blocks 7 / 15 / 23 / 31, every other block inside 112-pixel windows), read
 * by the NEXT fe_weights_commit(FE_MODEL_VLM).
 * fe_vlm_encode_images = `model.visual(pixel_values, grid_thw).pooler_output` (what generate() runs on the processor's output,
 * models/vlm_tagger.py:245-259): pixel_values [n_patches][1176] (the processor's flattened 3 x 2 x 14 x 14 patches, fp32), and the
 * index arrays transformers.vision_utils derives from image_grid_thw (facet_amd/vlm_tagger.py restates them in numpy): patch_pos_hw
 * [n_patches][2] = (row, column) of every patch IN WINDOW ORDER, window_index [n_patches / 4] = raster index of the 2x2 merge block at
 * each window-order slot, cu_window_seqlens [n_windows + 1] and cu_seqlens [n_images + 1] = segment bounds (in patches, window order)
 * of the windowed and the full-attention blocks. The merged embeddings [n_patches / 4][hidden] (raster order) stay on the device for
 * the next fe_vlm_prefill_images and are also copied to `embeds` when it is not NULL (bf16 values widened to float).
 * fe_vlm_prefill_images = fe_vlm_prefill whose rows image_rows[i] (flat index sequence * len + position, one per <|image_pad|> token, in
 * order) take the i-th embedding instead of the token's (`inputs_embeds.masked_scatter(image_mask, image_embeds)`); position_ids are
 * get_rope_index's (temporal / height / width ids of the image tokens). */
int fe_vlm_vision_configure(fe_ctx* ctx, int n_heads, const int* fullatt_block_indexes, int n_fullatt);
int fe_vlm_encode_images(fe_ctx* ctx, const float* pixel_values, int n_patches, const int32_t* patch_pos_hw, const int32_t* window_index,
                         const int32_t* cu_window_seqlens, int n_windows, const int32_t* cu_seqlens, int n_images, float* embeds);
int fe_vlm_prefill_images(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int len, int max_seq, const int32_t* image_rows,
                          int n_image_rows, int32_t* next_tokens, float* logits);
int fe_vlm_dims(fe_ctx* ctx, int* dims8);
int fe_vlm_prefill(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int len, int max_seq, int32_t* next_tokens,
                   float* logits);
int fe_vlm_decode_step(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int32_t* next_tokens, float* logits);
/* n_steps greedy decode steps without a host round trip (the loop generate() runs, models/vlm_tagger.py:255-259): the tokens to feed
 * first (the prefill's choice, [n_seq]) and their positions ([3][n_seq]) go up once, token ids / positions / cache length then live in
 * device memory and one captured HIP graph of a decode step is replayed; out_tokens [n_steps][n_seq] = the token each step chose.
 * End-of-sequence handling is the caller's (cut the rows at the first EOS id). */
int fe_vlm_generate(fe_ctx* ctx, const int32_t* tokens, const int32_t* position_ids, int n_seq, int n_steps, int32_t* out_tokens);

/* ---- device buffers (so callers can keep batches resident in HBM without torch) ------------- */
int fe_dev_alloc(fe_ctx* ctx, size_t bytes, void** d_out);
int fe_dev_free(fe_ctx* ctx, void* d_ptr);
int fe_memcpy_h2d(fe_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int fe_memcpy_d2h(fe_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---- timing / profiling on the engine's own HIP stream --------------------------------------- */
int fe_timer_start(fe_ctx* ctx);
int fe_timer_stop(fe_ctx* ctx, float* ms_out);
/* per-launch timing of every contraction kernel (serialises; for roofline accounting only) */
int fe_profile_enable(fe_ctx* ctx, int on);
int fe_profile_count(fe_ctx* ctx);
int fe_profile_get(fe_ctx* ctx, int i, char* name, int name_cap, double* flops, double* bytes, float* ms);
/* algorithmic FLOPs issued since the last fe_flops_reset (2*MAC of every contraction launched) */
int fe_flops_reset(fe_ctx* ctx);
int fe_flops_get(fe_ctx* ctx, double* flops);
/* Same counter minus the multiply-adds saved where a layer ran as Winograd F(2x2,3x3) (16 instead of 36 per 2x2 outputs):
 * the FLOPs the matrix cores actually executed. fe_flops_get stays the algorithmic (direct-convolution) count. */
int fe_flops_get_executed(fe_ctx* ctx, double* flops);
/* the part of fe_flops_get issued on the 2-byte (bf16 / fp16) matrix instructions: lets a mixed-precision run be priced per dtype */
int fe_flops_get_half(fe_ctx* ctx, double* flops);

/* ---- weights: replaces state_dict loading inside pyiqa.create_metric / open_clip.create_model /
 *      SAMPNet.load_state_dict (models/pyiqa_scorer.py:108, model_manager.py:140, samp_net.py:895).
 *      Tensors are passed by their checkpoint key names in PyTorch layout. -------------------- */
int fe_weights_begin(fe_ctx* ctx, int model);
int fe_weights_set(fe_ctx* ctx, int model, const char* name, const float* data, const int64_t* shape, int ndim);
int fe_weights_commit(fe_ctx* ctx, int model); /* validates, folds BN, packs for MFMA, uploads */
int fe_model_unload(fe_ctx* ctx, int model);   /* reference: ModelManager.unload_model (model_manager.py:237) */
int fe_model_loaded(fe_ctx* ctx, int model);   /* 1 / 0 */

/* ---- single ops (parity tests and building blocks; host NCHW in/out) -------------------------- */
/* y = act(conv2d(x, w) * scale + shift (+ res)), torch.nn.functional.conv2d semantics.
 * scale/shift/res may be NULL. res_after_act: add the residual after the activation. */
int fe_op_conv2d(fe_ctx* ctx, const float* x, int n, int c, int h, int w, const float* weight, int cout, int kh,
                 int kw, const float* scale, const float* shift, const float* res, int res_after_act, int stride,
                 int pad, int dil, int act, float* y);
/* test hook of the fused TOPIQ gate of the 64-channel pyramid level (kernels_gate.hip; 2-byte precisions only): x [n][64][h][w] (h, w
   multiples of 16), w0 / wx [64][64], w2 [64][64][3][3], w4 [1][64][3][3] ->
   y [n][64][h/16][w/16] = mean_16x16( act_g(wx x + bx) * sigmoid(w4 * act_w(w2 * act_w(w0 x + b0) + b2) + b4) ) */
int fe_op_topiq_gate64(fe_ctx* ctx, const float* x, int n, int h, int w, const float* w0, const float* b0, const float* w2,
                       const float* b2, const float* w4, float b4, const float* wx, const float* bx, int wblk_act,
                       int gate_act, float* y);
/* test hook of the halo-tiled 3x3 convolution 64 -> 64 (stride 1, padding 1; kernels_c64.hip; 2-byte precisions only): x [n][64][h][w],
   w2 [64][64][3][3], y = act2(conv(x, w2) * scale2 + shift2) [n][64][h][w]; with w3 [256][64] (then res [n][256][h][w] too, act2 = ReLU):
   y = relu((w3 . relu(conv * scale2 + shift2)) * scale3 + shift3 + res) [n][256][h][w] - the tail of a ResNet-50 layer1 bottleneck.
   scale / shift pointers may be null (1 / 0). */
int fe_op_conv3x3_c64(fe_ctx* ctx, const float* x, int n, int h, int w, const float* w2, const float* scale2,
                      const float* shift2, int act2, const float* w3, const float* scale3, const float* shift3,
                      const float* res, float* y);
int fe_op_maxpool2d(fe_ctx* ctx, const float* x, int n, int c, int h, int w, int k, int stride, int pad,
                    int ceil_mode, float* y);
int fe_op_bilinear(fe_ctx* ctx, const float* x, int n, int c, int h, int w, int ho, int wo, float* y);
int fe_op_adaptive_avgpool(fe_ctx* ctx, const float* x, int n, int c, int h, int w, int ho, int wo, float* y);
int fe_op_layernorm(fe_ctx* ctx, const float* x, int rows, int d, const float* g, const float* b, float eps, float* y);

/* developer hook: force a tile variant of the contraction kernel for every later launch (0 = automatic choice) */
int fe_set_conv_variant(fe_ctx* ctx, int variant);
/* developer hook: average ms of one conv shape on random device-resident data with a forced tile variant (0 = auto) */
int fe_bench_conv(fe_ctx* ctx, int n, int h, int w, int cin, int cout, int k, int stride, int pad, int with_res, int act,
                  int variant, int iters, float* ms_out);

/* ---- TOPIQ (reference: PyIQAScorer.score_image -> self.model(t), models/pyiqa_scorer.py:197-231) */
/* Activations inside pyiqa's GatedConv, read by the NEXT fe_weights_commit(FE_MODEL_TOPIQ): gate_act = the activation of the gated
 * branch x1, weight_blk_act = the one after weight_blk[0] and weight_blk[2] (each FE_ACT_RELU / FE_ACT_GELU / FE_ACT_SOFTPLUS).
 * Default GELU / GELU. Activations carry no parameters, so a checkpoint cannot tell which a pyiqa release used (pyiqa is not
 * vendored in the reference, models/pyiqa_scorer.py:33-39,108-111): the choice is a load-time option [DEP-KNOWLEDGE]. */
int fe_topiq_configure(fe_ctx* ctx, int gate_act, int weight_blk_act);
/* A TOPIQ model committed under a 2-byte precision scores images of fewer than `pixels` pixels on its fp32 weights (0, the default:
 * never). Small images give the head a few dozen tokens per level, too few to average the 2-byte rounding noise below the 1e-3 gate;
 * the PARITY precision policy (facet_amd/precision.py) sets 65536. */
int fe_topiq_f32_below(fe_ctx* ctx, long long pixels);
/* dims = {channels, height, width} of pyramid level `level` for h x w inputs (after the > 1024 LANCZOS cap of
 * models/pyiqa_scorer.py:131-153): the size of one image's block in fe_topiq_features' output. No context needed. */
int fe_topiq_feature_shape(int h, int w, int level, int dims[3]);
/* images: n x h x w x 3 uint8 RGB (all the same size, h and w multiples of 32).
 * level 0..4 = ResNet-50 pyramid feature (stem-relu, layer1..4) returned NCHW to host `out`. */
int fe_topiq_features(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, int level, float* out);
/* raw MOS per image (before the reference's clamp[0,1]*10, pyiqa_scorer.py:166-195) */
int fe_topiq_score(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* scores);

/* ---- U2-Net-P + SAMP-Net (reference models/samp_net.py) ------------------------------------------- */
/* x: fp32 NCHW [n,3,h,w], already Resize(224)+ToTensor+ImageNet-normalised as SAMPNetScorer.preprocess yields
 * (samp_net.py:904-928). fe_u2netp_saliency = SaliencyDetector.detect (:407-422): sal_out [n,h,w] in (0,1). */
int fe_u2netp_saliency(fe_ctx* ctx, const float* x, int n, int h, int w, float* sal_out);
/* = the model part of SAMPNetScorer.score_batch (:1005-1010): saliency = U2NETP(x)[0]; SAMPNet(x, saliency).
 * x is [n,3,224,224]. Outputs: pattern_weights [n,8] (logits), attributes [n,6], score_dist [n,5];
 * sal_out may be NULL. Post-processing (softmax/argmax/expectation, :957-989) stays on the host. */
int fe_samp_forward(fe_ctx* ctx, const float* x, int n, float* pattern_weights, float* attributes, float* score_dist,
                    float* sal_out);

/* ---- CLIP ViT-L/14 image tower + aesthetic MLP (reference processing/scorer.py:640-673) ------------- */
/* x: fp32 NCHW [n,3,224,224] as open_clip's eval transform yields. Any of the three outputs may be NULL:
 *   features      [n,768] = model.encode_image(x)                          (scorer.py:662)
 *   emb_norm      [n,768] = F.normalize(features, dim=-1)                  (:663; stored as 3072-byte blobs, :670)
 *   aesthetic_raw [n]     = aesthetic_head(features) (Linear-ReLU-Linear)  (:664; the (x+1)*5 clamp [0,10] stays on host, :669)
 */
int fe_clip_encode_image(fe_ctx* ctx, const float* x, int n, int on_device, float* features, float* emb_norm,
                         float* aesthetic_raw);

/* aesthetic_head on vectors already at hand: feats [n,768] -> aesthetic_raw [n]. The reference recomputes scores from STORED
 * embeddings this way (Facet.score_from_embedding, processing/scorer.py:619-629: the 3072-byte blob, i.e. the L2-normalised
 * embedding, goes through the same MLP); the (x+1)*5 clamp stays on the host. */
int fe_aesthetic_score(fe_ctx* ctx, const float* feats, int n, float* aesthetic_raw);

/* ---- preprocessing: PIL-exact uint8 resampling (bit-for-bit PIL.Image.resize, RGB 8-bit) ---------------- */
enum fe_filter { FE_LANCZOS = 1, FE_BILINEAR = 2, FE_BICUBIC = 3 }; /* PIL.Image.Resampling values */
/* src [n,h,w,3] uint8 -> dst [n,oh,ow,3] uint8; replaces PIL `image.resize((ow,oh), filter)`
 * (models/pyiqa_scorer.py:153 LANCZOS; torchvision Resize inside models/samp_net.py:823-830; open_clip transform). */
int fe_resize_u8(fe_ctx* ctx, const uint8_t* src, int n, int h, int w, int oh, int ow, int filter, int on_device,
                 uint8_t* dst);

/* ---- image-level entry points (uint8 HWC images in, per-image results out) ------------------------------ */
/* CLIP from raw RGB images: open_clip eval transform on the GPU (PIL-bicubic shorter side -> 224, center crop 224,
 * /255, CLIP mean/std) + fe_clip_encode_image. Replaces batch_processor.py:95 `scorer.preprocess(pil)` +
 * scorer.py:640-673. */
int fe_clip_encode_images(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* features,
                          float* emb_norm, float* aesthetic_raw);
/* SAMPNetScorer.score_batch from raw images (samp_net.py:904-928,991-1010): optional BGR->RGB, PIL-bilinear
 * Resize((224,224)), ToTensor, ImageNet Normalize, U2NETP saliency, SAMPNet. Outputs as fe_samp_forward. */
int fe_samp_score_images(fe_ctx* ctx, const uint8_t* img, int n, int h, int w, int bgr, int on_device,
                         float* pattern_weights, float* attributes, float* score_dist);
/* CLIP text tower (built when the FE_MODEL_CLIP checkpoint carries token_embedding.weight): tokens int32 [n][77] ->
 * un-normalised text features [n][768]. Replaces `clip_model.encode_text(tokens)` (models/tagger.py:69-75); pooling is
 * at argmax(token id) = the EOT token, as open_clip does. */
int fe_clip_encode_text(fe_ctx* ctx, const int32_t* tokens, int n, int ctx_len, float* features);

/* Batched tag scoring: sims[n][T] = emb[n][d] . text[T][d]^T. Replaces the per-image matmul + loop of
 * CLIPTagger.get_tags_from_embedding (models/tagger.py:100-106); selection (max over synonyms, threshold, top-k) stays on host. */
int fe_tag_similarities(fe_ctx* ctx, const float* emb, int n, const float* text, int T, int d, float* sims);

/* The whole ensemble on one resident batch — what processing/batch_processor.py:169-360 sequences per image.
 * records [n][FE_RECORD_FLOATS]: [0] topiq raw MOS, [1] aesthetic raw, [2..9] SAMP pattern logits, [10..15] SAMP
 * attributes, [16..20] SAMP score distribution, [21..788] L2-normalised CLIP embedding. Fields of models that are
 * not loaded stay 0; *models_run (nullable) = bitmask 1 topiq | 2 clip | 4 samp. This is also the fixed-size
 * per-image record that ranks all-gather in multi-GPU runs. */
#define FE_RECORD_FLOATS 789
int fe_ensemble_score(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* records,
                      int* models_run);

/* Which of the loaded models fe_ensemble_score runs: bitmask 1 topiq | 2 clip (+ aesthetic head) | 4 samp; default 7. The reference's
 * multi-pass mode runs one model group per pass over the same images (processing/multi_pass.py:481-644). */
int fe_ensemble_select(fe_ctx* ctx, int models);
/* fe_ensemble_score with the records left in device memory: d_records [n][ld_records] floats, ld_records >= FE_RECORD_FLOATS (further
 * columns are not touched). Returns once the engine stream has drained: the buffer can go straight into the multi-GPU all-gather
 * (RCCL reads it in place; there is no device -> host -> device hop in the step). */
int fe_ensemble_score_dev(fe_ctx* ctx, const uint8_t* rgb, int n, int h, int w, int on_device, float* d_records, int ld_records,
                          int* models_run);

/* ---- ONNX-subset graph runtime --------------------------------------------------------------------------------
 * Replaces the onnxruntime InferenceSessions that insightface.app.FaceAnalysis(name='buffalo_l') opens for the reference
 * (analyzers/face.py:30-38: det_10g.onnx, 2d106det.onnx, w600k_r50.onnx; invoked through face_app.get at :99). The
 * engine parses the .onnx bytes itself (no protobuf/onnx dependency) and executes the nodes on its own HIP kernels.
 * Supported operators: Conv (dense and depthwise), Gemm, MatMul (constant B), BatchNormalization, Relu, PRelu, LeakyRelu,
 * Sigmoid, Add/Sub/Mul/Div, MaxPool, AveragePool, GlobalAveragePool, Resize/Upsample (nearest asymmetric-floor, linear
 * half-pixel), Concat (channels), Flatten, Reshape, Transpose, Squeeze, Unsqueeze, Softmax, Identity, Dropout, Constant and
 * the constant shape arithmetic exporters emit (Shape, Gather, Cast, Slice, Concat, Floor, Ceil). Anything else fails with
 * an error (fe_last_error) naming the operator. One float image input [N,C,H,W]. */
#define FE_GRAPH_SLOTS 8
/* Parses an .onnx buffer on the host only (no context, no GPU): counts and the declared input dims, or an error text.
 * Lets callers validate a model file before a device is involved. Returns FE_OK or FE_ERR_RUNTIME. */
int fe_onnx_probe(const void* onnx_bytes, size_t len, int* n_nodes, int* n_initializers, int* n_outputs, int64_t in_dims[4],
                  char* err, int err_cap);
enum fe_graph_slot { FE_GRAPH_FACE_DET = 0, FE_GRAPH_FACE_LMK = 1, FE_GRAPH_FACE_REC = 2 };
int fe_graph_load(fe_ctx* ctx, int slot, const void* onnx_bytes, size_t len);
int fe_graph_unload(fe_ctx* ctx, int slot);
int fe_graph_loaded(fe_ctx* ctx, int slot);
/* in_dims: the declared input shape (-1 = dynamic). flags: bit0 / bit1 = a node named Sub* / Mul* (or _minus* / _mul*)
 * is among the first 8 nodes, the probe insightface uses to choose input mean/std [DEP-KNOWLEDGE]. */
int fe_graph_info(fe_ctx* ctx, int slot, int* n_nodes, int* n_outputs, int64_t in_dims[4], int* flags);
/* x: fp32 NCHW (host, or device when on_device). Outputs stay inside the engine until the next run on this slot. */
int fe_graph_run(fe_ctx* ctx, int slot, const float* x, int n, int c, int h, int w, int on_device);
int fe_graph_output_info(fe_ctx* ctx, int slot, int i, char* name, int name_cap, int64_t dims[6], int* rank);
int fe_graph_output_copy(fe_ctx* ctx, int slot, int i, float* dst, size_t cap_floats);

/* ---- face path: what insightface's FaceAnalysis.get does around its three sessions (analyzers/face.py:99) -------------
 * [DEP-KNOWLEDGE: insightface model_zoo scrfd.py / landmark.py / arcface_onnx.py, utils/face_align.py; OpenCV resize/warpAffine]
 *
 * fe_face_detect = SCRFD.detect for a batch of equally sized BGR uint8 images: aspect-preserving cv2.resize (INTER_LINEAR)
 * into the top-left of a zero det_h x det_w canvas, blobFromImage((x-127.5)/128, swapRB), the graph in FE_GRAPH_FACE_DET,
 * then per stride: score >= thresh, distance2bbox / distance2kps from anchor centres, / det_scale. Candidates come back
 * unordered as 16 floats each: score, x1,y1,x2,y2, five (x,y) keypoints, stride level. counts[i] is the number found for
 * image i (only the first max_cand are stored). Sorting and NMS (tiny, data dependent) stay with the caller.
 * det_scale_out (nullable) receives new_height / h. */
int fe_face_detect(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int det_h, int det_w, float thresh,
                   int max_cand, float* cand, int* counts, float* det_scale_out);
/* Warps m square crops with cv2.warpAffine(img[img_index[f]], M[f] (2x3 forward matrix, row-major doubles), (size,size),
 * borderValue=0), applies blobFromImages((x-mean)*scale, swapRB) and runs graph `slot` on all crops in one batch; out
 * [m][out_dim] receives its first output. This is face_align.norm_crop + ArcFaceONNX.get_feat (size 112) and
 * face_align.transform + Landmark.get's forward (size 192). crops_out (nullable) receives the uint8 crops [m][size][size][3];
 * out may be null when only the crops are wanted. */
int fe_face_crops_run(fe_ctx* ctx, int slot, const uint8_t* bgr, int n, int h, int w, int on_device, int m, const int* img_index,
                      const double* M, int size, float mean, float scale, int swap_rb, float* out, int out_dim, uint8_t* crops_out);
/* FaceAnalysis.get(img) for a whole batch in ONE call (reference: face_app.get at analyzers/face.py:99, once per image):
 * fe_face_detect's pipeline, then on the host side of the engine score-sort + NMS(nms_thresh) per image, then for the best
 * max_faces faces of every image: Landmark.get (192-crop, graph FE_GRAPH_FACE_LMK, back-projection) and ArcFaceONNX.get
 * (5-point similarity crop 112, graph FE_GRAPH_FACE_REC), both batched over all faces of a micro-batch. Input normalisation
 * per graph follows insightface's Sub/Mul probe. faces [n][max_faces][FE_FACE_FLOATS]: bbox x1,y1,x2,y2, det_score,
 * kps[5][2], landmark_2d_106[106][2], embedding[512] (zeros for absent models / unused slots); counts[i] = faces that
 * survived NMS for image i (may exceed max_faces). models_run (nullable): 1 det | 2 landmarks | 4 recognition.
 * The fixed-size slots are what ranks all-gather in multi-GPU runs. */
#define FE_FACE_FLOATS 739
int fe_face_analyze(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int det_h, int det_w, float det_thresh,
                    float nms_thresh, int max_faces, float* faces, int* counts, int* models_run);
/* cv2.resize(img, (ow, oh)) with INTER_LINEAR on uint8 HWC 3-channel images, the fixed-point path OpenCV takes. */
int fe_cv_resize_linear_u8(fe_ctx* ctx, const uint8_t* src, int n, int h, int w, int oh, int ow, uint8_t* dst);

/* ---- per-image technical statistics ----------------------------------------------------------------
 * The scans reference analyzers/image_cache.py:28-33 (cv2.cvtColor BGR2GRAY / BGR2HSV, cv2.Laplacian(CV_64F).var()) and
 * analyzers/technical.py:39-342 (calcHist 256 / 180x256, saturation mean, percentiles, Immerkaer cv2.filter2D) run per image
 * on the CPU, as two HBM-bound GPU passes over a BGR uint8 batch. stats [n][FE_STATS_DOUBLES] (all exact integers except
 * [260]): [0..255] gray histogram counts; [256] sum and [257] sum of squares of the 4-neighbour Laplacian (reflect-101 border);
 * [258] sum |Immerkaer 3x3 response|; [259] sum of HSV saturation; [260] sum c*log2(c) over the 180x256 hue-saturation
 * histogram (entropy = log2(N) - [260]/N); [261..263] reserved. gray_out [n][h][w] / hsv_out [n][h][w][3] (nullable) return
 * the converted images themselves. facet_amd/image_stats.py turns the record into the reference's seven metric dicts. */
#define FE_STATS_DOUBLES 264
int fe_image_stats(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, double* stats, uint8_t* gray_out,
                   uint8_t* hsv_out);

/* Laplacian statistics of m rectangular ROIs: what analyzers/face.py:160-176 (eye regions) and :272-279
 * (face crop) compute with cv2.cvtColor(roi, BGR2GRAY) + cv2.Laplacian(gray, CV_64F).var() + np.mean(gray) per face on the CPU.
 * rois [m][4] = x1,y1,x2,y2 (exclusive, already clipped to the image like the reference's slices); borders reflect (101) at
 * the ROI edge. out [m][4]: sum of Laplacian, sum of squares, sum of gray, pixel count (exact integers). Empty ROIs give 0s. */
int fe_roi_laplacian(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int m, const int* img_index, const int* rois,
                     double* out);

/* The reference holds every image twice, as PIL RGB and as cv2 BGR (processing/batch_processor.py:200-215). With a resident batch
 * the second copy is made on the device: dst_device [pixels][3] = src [pixels][3] with the first and third byte of every pixel
 * exchanged. src: host (on_device = 0) or device memory; not in place. */
int fe_swap_rb_u8(fe_ctx* ctx, const uint8_t* src, int on_device, size_t pixels, uint8_t* dst_device);

/* Leading lines (last item): the reference's CompositionAnalyzer.detect_leading_lines (analyzers/composition.py:191-261)
 * runs cv2.GaussianBlur(gray, (5,5), 0), cv2.Canny(blurred, 50, 150) and cv2.HoughLinesP(edges, 1, pi/180, 80, minLineLength =
 * int(min(h,w)*0.15), maxLineGap = 20) per image on the CPU. Here the pixel scans (gray, 5x5 fixed-point blur, Sobel, L1 magnitude,
 * non-maximum suppression + thresholds) run on the GPU over the BGR batch; hysteresis and the progressive probabilistic Hough
 * transform (rho 1 px, theta 1 degree; a sequential, pseudo-random-order vote-and-erase loop) run on the host, one image per
 * thread. lines [n][max_lines][4] = x1,y1,x2,y2 in the order found, counts [n] = segments found per image (may exceed max_lines:
 * only the first max_lines are stored - call again with more room); both nullable together. edges_out [n][h][w] (nullable, host)
 * receives the Canny edge image (0 / 255). facet_amd/composition.py scores the segments as the reference does. */
int fe_leading_lines(fe_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int on_device, int canny_low, int canny_high, int threshold,
                     int min_line_length, int max_line_gap, int max_lines, int* lines, int* counts, uint8_t* edges_out);

#ifdef __cplusplus
}
#endif
#endif /* FACET_ENGINE_H */

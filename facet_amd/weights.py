"""Checkpoint key/shape specs of the hot-path models and seeded synthetic checkpoints.

No weight files exist offline, so benches and parity tests use synthetic checkpoints:
every tensor is drawn from numpy's PCG64 keyed by (seed, crc32(name)) with fan-in scaled magnitudes, under
the *real* checkpoint key names (timm resnet50 / pyiqa CFANet / open_clip VisionTransformer / the
reference's in-tree SAMPNet + U2NETP, models/samp_net.py). A real checkpoint loads through the same
Engine.load_weights(name -> tensor) path.
"""
import zlib

import numpy as np

# ---------------------------------------------------------------------------------------------------
# spec builders: list of (name, shape, kind)
# ---------------------------------------------------------------------------------------------------

def _conv(spec, name, cout, cin, kh, kw=None, bias=False):
    spec.append((name + ".weight", (cout, cin, kh, kw if kw is not None else kh), "conv"))
    if bias:
        spec.append((name + ".bias", (cout,), "bias"))


def _bn(spec, name, c, last=False, kind=None):
    spec.append((name + ".weight", (c,), kind or ("bn_w_last" if last else "bn_w")))
    spec.append((name + ".bias", (c,), "bn_b"))
    spec.append((name + ".running_mean", (c,), "bn_mean"))
    spec.append((name + ".running_var", (c,), "bn_var"))


def _linear(spec, name, out, inp, bias=True):
    spec.append((name + ".weight", (out, inp), "linear"))
    if bias:
        spec.append((name + ".bias", (out,), "bias"))


def _ln(spec, name, d):
    spec.append((name + ".weight", (d,), "ln_w"))
    spec.append((name + ".bias", (d,), "ln_b"))


def resnet_spec(prefix, bottleneck, blocks, seq_names=False):
    """timm/torchvision ResNet keys (conv1,bn1,layerL.B.convK,...); seq_names=True gives the
    nn.Sequential(*children[:-2]) numbering used by reference models/samp_net.py:652-662."""
    spec = []
    n = (lambda plain, seq: prefix + (str(seq) if seq_names else plain))
    _conv(spec, n("conv1", 0), 64, 3, 7)
    _bn(spec, n("bn1", 1), 64)
    inpl = 64
    exp = 4 if bottleneck else 1
    for li, nb in enumerate(blocks):
        planes = 64 * 2 ** li
        lp = prefix + (str(4 + li) if seq_names else f"layer{li + 1}")
        for bi in range(nb):
            bp = f"{lp}.{bi}"
            stride = 2 if (bi == 0 and li > 0) else 1
            if bottleneck:
                _conv(spec, bp + ".conv1", planes, inpl, 1); _bn(spec, bp + ".bn1", planes)
                _conv(spec, bp + ".conv2", planes, planes, 3); _bn(spec, bp + ".bn2", planes)
                _conv(spec, bp + ".conv3", planes * 4, planes, 1); _bn(spec, bp + ".bn3", planes * 4, last=True)
            else:
                _conv(spec, bp + ".conv1", planes, inpl, 3); _bn(spec, bp + ".bn1", planes)
                _conv(spec, bp + ".conv2", planes, planes, 3); _bn(spec, bp + ".bn2", planes, last=True)
            if stride != 1 or inpl != planes * exp:
                _conv(spec, bp + ".downsample.0", planes * exp, inpl, 1)
                _bn(spec, bp + ".downsample.1", planes * exp)
            inpl = planes * exp
    return spec


def _mha(spec, name, d):
    spec.append((name + ".in_proj_weight", (3 * d, d), "linear"))
    spec.append((name + ".in_proj_bias", (3 * d,), "bias"))
    _linear(spec, name + ".out_proj", d, d)


def _enc_layer(spec, name, d, ff):
    _mha(spec, name + ".self_attn", d)
    _linear(spec, name + ".linear1", ff, d)
    _linear(spec, name + ".linear2", d, ff)
    _ln(spec, name + ".norm1", d)
    _ln(spec, name + ".norm2", d)


def _dec_layer(spec, name, d, ff):
    # (pyiqa's decoder layer also owns an unused `self_attn`; its forward is cross-attention only)
    _mha(spec, name + ".multihead_attn", d)
    _linear(spec, name + ".linear1", ff, d)
    _linear(spec, name + ".linear2", d, ff)
    _ln(spec, name + ".norm1", d)
    _ln(spec, name + ".norm2", d)
    _ln(spec, name + ".norm3", d)


TOPIQ_DIMS = [64, 256, 512, 1024, 2048]
TOPIQ_INTER = 256
TOPIQ_FF = 1024


def topiq_spec():
    """pyiqa CFANet (topiq_nr: resnet50, use_ref=False, inter_dim=256, 4 heads, 1 attn layer). [DEP-KNOWLEDGE]"""
    spec = resnet_spec("semantic_model.", True, [3, 4, 6, 3])
    d, ff = TOPIQ_INTER, TOPIQ_FF
    for i, dim in enumerate(TOPIQ_DIMS):
        g = f"weight_pool.{i}"
        _conv(spec, g + ".splitconv", dim * 2, dim, 1, bias=True)
        _conv(spec, g + ".weight_blk.0", 64, dim, 1, bias=True)
        _conv(spec, g + ".weight_blk.2", 64, 64, 3, bias=True)
        _conv(spec, g + ".weight_blk.4", 1, 64, 3, bias=True)
        _conv(spec, f"dim_reduce.{i}.0", d, dim, 1, bias=True)
        _enc_layer(spec, f"sa_attn_blks.{i}.layers.0", d, ff)
    for i in range(len(TOPIQ_DIMS) - 1):
        _dec_layer(spec, f"attn_blks.{i}.layers.0", d, ff)
    _enc_layer(spec, "attn_pool", d, ff)
    _ln(spec, "score_linear.0", d)
    _linear(spec, "score_linear.1", d, d)
    _ln(spec, "score_linear.3", d)
    _linear(spec, "score_linear.4", d, d)
    _linear(spec, "score_linear.6", 1, d)
    spec.append(("h_emb", (1, d // 2, 32, 1), "emb"))
    spec.append(("w_emb", (1, d // 2, 1, 32), "emb"))
    return spec


def clip_vit_spec(width=1024, layers=24, patch=14, grid=16, out_dim=768):
    """open_clip VisionTransformer (ViT-L/14) keys under `visual.` [DEP-KNOWLEDGE]."""
    spec = []
    p = "visual."
    spec.append((p + "conv1.weight", (width, 3, patch, patch), "conv"))
    spec.append((p + "class_embedding", (width,), "emb"))
    spec.append((p + "positional_embedding", (grid * grid + 1, width), "emb"))
    _ln(spec, p + "ln_pre", width)
    for i in range(layers):
        b = f"{p}transformer.resblocks.{i}"
        _ln(spec, b + ".ln_1", width)
        spec.append((b + ".attn.in_proj_weight", (3 * width, width), "linear"))
        spec.append((b + ".attn.in_proj_bias", (3 * width,), "bias"))
        _linear(spec, b + ".attn.out_proj", width, width)
        _ln(spec, b + ".ln_2", width)
        _linear(spec, b + ".mlp.c_fc", width * 4, width)
        _linear(spec, b + ".mlp.c_proj", width, width * 4)
    _ln(spec, p + "ln_post", width)
    spec.append((p + "proj", (width, out_dim), "proj"))
    return spec


def clip_text_spec(width=768, layers=12, ctx=77, vocab=49408, out_dim=768):
    """open_clip CLIP text tower keys (top level of the CLIP checkpoint, next to `visual.*`) [DEP-KNOWLEDGE]."""
    spec = [("token_embedding.weight", (vocab, width), "emb"), ("positional_embedding", (ctx, width), "emb")]
    for i in range(layers):
        b = f"transformer.resblocks.{i}"
        _ln(spec, b + ".ln_1", width)
        spec.append((b + ".attn.in_proj_weight", (3 * width, width), "linear"))
        spec.append((b + ".attn.in_proj_bias", (3 * width,), "bias"))
        _linear(spec, b + ".attn.out_proj", width, width)
        _ln(spec, b + ".ln_2", width)
        _linear(spec, b + ".mlp.c_fc", width * 4, width)
        _linear(spec, b + ".mlp.c_proj", width, width * 4)
    _ln(spec, "ln_final", width)
    spec.append(("text_projection", (width, out_dim), "proj"))
    return spec


def aesthetic_spec():
    """Linear(768,256)-ReLU-Linear(256,1), reference processing/scorer.py:579-583."""
    spec = []
    _linear(spec, "0", 256, 768)
    _linear(spec, "2", 1, 256)
    return spec


_RSU = {  # name -> (depth, dilated)
    "stage1": (7, False), "stage2": (6, False), "stage3": (5, False), "stage4": (4, False),
    "stage5": (4, True), "stage6": (4, True),
    "stage5d": (4, True), "stage4d": (4, False), "stage3d": (5, False), "stage2d": (6, False), "stage1d": (7, False),
}
_RSU_IN = {"stage1": 3, "stage2": 64, "stage3": 64, "stage4": 64, "stage5": 64, "stage6": 64,
           "stage5d": 128, "stage4d": 128, "stage3d": 128, "stage2d": 128, "stage1d": 128}


def rsu_convs(depth, in_ch, mid=16, out=64):
    """(name, cin, cout) of the REBNCONV units of one RSU block (reference models/samp_net.py:62-255)."""
    convs = [("rebnconvin", in_ch, out), ("rebnconv1", out, mid)]
    for k in range(2, depth + 1):
        convs.append((f"rebnconv{k}", mid, mid))
    for k in range(depth - 1, 1, -1):
        convs.append((f"rebnconv{k}d", mid * 2, mid))
    convs.append(("rebnconv1d", mid * 2, out))
    return convs


def u2netp_spec(prefix=""):
    spec = []
    for st, (depth, _dil) in _RSU.items():
        for name, cin, cout in rsu_convs(depth, _RSU_IN[st]):
            _conv(spec, f"{prefix}{st}.{name}.conv_s1", cout, cin, 3, bias=True)
            _bn(spec, f"{prefix}{st}.{name}.bn_s1", cout, kind="bn_w_half")
    for k in range(1, 7):
        spec.append((f"{prefix}side{k}.weight", (1, 64, 3, 3), "conv_small"))
        spec.append((f"{prefix}side{k}.bias", (1,), "bias"))
    _conv(spec, f"{prefix}outconv", 1, 6, 1, bias=True)
    return spec


SAMP_PATTERN_SHAPES = [(1296, 2, 1), (1296, 1, 2), (1373, 2, 1), (1373, 2, 1), (1296, 2, 1), (1296, 2, 2),
                       (1324, 2, 2), (836, 3, 3)]


def sampnet_spec():
    """Reference models/samp_net.py:665-758 (all Linear/Conv bias-free)."""
    spec = resnet_spec("backbone.", False, [2, 2, 2, 2], seq_names=True)
    _linear(spec, "pattern_weight_layer.3", 8, 512, bias=False)
    for i, (c, kh, kw) in enumerate(SAMP_PATTERN_SHAPES):
        spec.append((f"pattern_module.conv_list.{i}.0.weight", (1024, c, kh, kw), "conv"))
    _linear(spec, "att_feature_layer.0", 512, 1024, bias=False)
    _linear(spec, "att_pred_layer.0", 6, 512, bias=False)
    _linear(spec, "com_feature_layer.0", 512, 1024, bias=False)
    _linear(spec, "alpha_predict_layer.0", 2, 1024, bias=False)
    _linear(spec, "com_pred_layer.0", 1024, 1024, bias=False)
    _linear(spec, "com_pred_layer.3", 512, 1024, bias=False)
    _linear(spec, "com_pred_layer.5", 5, 512, bias=False)
    return spec


def qwen2_5_vl_text_spec(hidden=3584, layers=28, heads=28, kv_heads=4, inter=18944, vocab=152064):
    """Text decoder of transformers' Qwen2_5_VLForConditionalGeneration (the class models/vlm_tagger.py:163-184 instantiates), its
    state-dict key names (transformers 5.x: model.language_model.*, lm_head.weight). Defaults = Qwen2.5-VL-7B-Instruct's geometry
    (head_dim 128, q/k/v with bias, untied lm_head)."""
    hd = hidden // heads
    spec = [("model.language_model.embed_tokens.weight", (vocab, hidden), "emb1")]
    for i in range(layers):
        p = f"model.language_model.layers.{i}"
        _linear(spec, p + ".self_attn.q_proj", heads * hd, hidden)
        _linear(spec, p + ".self_attn.k_proj", kv_heads * hd, hidden)
        _linear(spec, p + ".self_attn.v_proj", kv_heads * hd, hidden)
        _linear(spec, p + ".self_attn.o_proj", hidden, heads * hd, bias=False)
        _linear(spec, p + ".mlp.gate_proj", inter, hidden, bias=False)
        _linear(spec, p + ".mlp.up_proj", inter, hidden, bias=False)
        _linear(spec, p + ".mlp.down_proj", hidden, inter, bias=False)
        spec.append((p + ".input_layernorm.weight", (hidden,), "ln_w"))
        spec.append((p + ".post_attention_layernorm.weight", (hidden,), "ln_w"))
    spec.append(("model.language_model.norm.weight", (hidden,), "ln_w"))
    spec.append(("lm_head.weight", (vocab, hidden), "linear"))
    return spec


def qwen2_5_vl_vision_spec(hidden=1280, depth=32, inter=3420, out_hidden=3584, patch=14, temporal=2):
    """Vision tower of Qwen2_5_VLForConditionalGeneration (`model.visual.*`): Conv3d patch embedding (no bias), `depth` blocks of
    RMSNorm - fused qkv (+bias) - proj (+bias) - RMSNorm - SwiGLU MLP (+biases), the patch merger (RMSNorm, Linear(4 hidden, 4 hidden),
    GELU, Linear(4 hidden, out_hidden)). Defaults = Qwen2.5-VL-7B-Instruct (16 heads of 80)."""
    v = "model.visual."
    spec = [(v + "patch_embed.proj.weight", (hidden, 3, temporal, patch, patch), "linear_nd")]
    for i in range(depth):
        b = f"{v}blocks.{i}"
        spec.append((b + ".norm1.weight", (hidden,), "ln_w"))
        spec.append((b + ".norm2.weight", (hidden,), "ln_w"))
        _linear(spec, b + ".attn.qkv", 3 * hidden, hidden)
        _linear(spec, b + ".attn.proj", hidden, hidden)
        _linear(spec, b + ".mlp.gate_proj", inter, hidden)
        _linear(spec, b + ".mlp.up_proj", inter, hidden)
        _linear(spec, b + ".mlp.down_proj", hidden, inter)
    spec.append((v + "merger.ln_q.weight", (hidden,), "ln_w"))
    _linear(spec, v + "merger.mlp.0", 4 * hidden, 4 * hidden)
    _linear(spec, v + "merger.mlp.2", out_hidden, 4 * hidden)
    return spec


# reduced vision tower of the parity tests: Qwen2.5-VL's head_dim 80, window attention with one full-attention block
VLM_VISION_TINY = dict(hidden=160, depth=3, inter=320, out_hidden=512)

# the reduced-depth configuration of the VLM parity tests (tests/golden/make_vlm_golden.py): Qwen2.5-VL's head_dim 128 and 2:1 grouped
# KV heads at a size the CPU oracle generates from in seconds
VLM_TINY = dict(hidden=512, layers=4, heads=4, kv_heads=2, inter=1408, vocab=2048)

SPECS = {
    "qwen2_5_vl_text": qwen2_5_vl_text_spec,
    "qwen2_5_vl_text_tiny": lambda: qwen2_5_vl_text_spec(**VLM_TINY),
    "qwen2_5_vl_tiny": lambda: qwen2_5_vl_text_spec(**VLM_TINY) + qwen2_5_vl_vision_spec(**VLM_VISION_TINY),
    "topiq": topiq_spec,
    "resnet50": lambda: resnet_spec("semantic_model.", True, [3, 4, 6, 3]),
    "clip": clip_vit_spec,
    "clip_text": clip_text_spec,
    "clip_full": lambda: clip_vit_spec() + clip_text_spec(),
    "aesthetic": aesthetic_spec,
    "u2netp": u2netp_spec,
    "samp_net": sampnet_spec,
}


# ---------------------------------------------------------------------------------------------------
# seeded synthetic tensors
# ---------------------------------------------------------------------------------------------------

def _draw(rng, shape, kind):
    shape = tuple(int(s) for s in shape)
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
    if kind == "conv_small":  # U2-Net-P side heads: keeps the fused saliency logit out of sigmoid saturation
        fan_in = shape[1] * shape[2] * shape[3]
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.7 * np.sqrt(2.0 / fan_in))
    if kind == "bn_w_half":
        return rng.uniform(0.6, 0.9, shape).astype(np.float32)
    if kind == "linear":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(shape[1]))
    if kind == "linear_nd":      # a linear map stored with an N-d kernel (Conv3d patch embedding): fan-in = everything but the first axis
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(np.prod(shape[1:])))
    if kind == "proj":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(shape[0]))
    if kind == "bias":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
    if kind == "bn_w":
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if kind == "bn_w_last":  # keeps residual sums tame over 16 blocks
        return rng.uniform(0.2, 0.4, shape).astype(np.float32)
    if kind == "bn_b":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
    if kind == "bn_mean":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
    if kind == "bn_var":
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if kind == "ln_w":
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if kind == "ln_b":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
    if kind == "emb":
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
    if kind == "emb1":      # token embeddings of a pre-norm decoder (RMSNorm rescales them; unit scale keeps bf16 away from its subnormals)
        return rng.standard_normal(shape, dtype=np.float32)
    raise ValueError(kind)


def synthetic_state_dict(model, seed=0, spec=None):
    """name -> float32 ndarray for `model` in SPECS (or an explicit spec list)."""
    spec = spec if spec is not None else SPECS[model]()
    out = {}
    for name, shape, kind in spec:
        rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
        out[name] = _draw(rng, shape, kind)
    return out


def synthetic_images(seed, n, h, w):
    """uint8 HWC RGB, default_rng(seed).integers(0,256)."""
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def checkpoint_or_synthetic(kind, weights_path, synthetic, seed, loader):
    """The state dict the drop-in wrappers load. A real checkpoint path wins. Without one the reference would download the weights
    (pyiqa / open_clip / the aesthetic head, models/pyiqa_scorer.py:108, model_manager.py:140, processing/scorer.py:560-577) and fail
    loudly when it cannot; there is no network here, so the wrappers raise FileNotFoundError instead of scoring with made-up
    weights - unless the caller opts in with synthetic=True or FACET_AMD_SYNTHETIC=1 (tests, bench.py, tools/)."""
    import os
    if weights_path:
        return loader(weights_path)
    if synthetic or os.environ.get("FACET_AMD_SYNTHETIC") == "1":
        return synthetic_state_dict(kind, seed)
    raise FileNotFoundError(
        f"no {kind} checkpoint path given. Pass weights_path=... (a local .safetensors / .pth file); the seeded synthetic checkpoint "
        "is only used when asked for explicitly (synthetic=True or FACET_AMD_SYNTHETIC=1) because its scores are meaningless.")

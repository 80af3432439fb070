"""Per-model precision policies of the engine (measured by tools/precision_ablation.py, profiles/r03_precision_ablation.txt).

Precision is a property of a model's committed weights (include/facet_engine.h fe_set_precision), so a policy is a mapping
model name -> precision name, applied at load time. The reference itself runs everything in fp32 on the CPU and halves CLIP only
(`self.model.half()`) on a GPU (processing/scorer.py:513-516).

  FP32           every model fp32: the arithmetic of the reference's CPU path (the headline).
  REFERENCE_GPU  the reference's own GPU precisions: CLIP in fp16, everything else fp32.
  PARITY         the fastest assignment whose FINAL scores stay within SURVEY 8(d)'s 1e-3 of the fp32 oracle on every model:
                 TOPIQ and U2-Net-P in fp16 (1e-4 - 6e-4 on the MOS from 256 x 256 pixels up; smaller images, whose few dozen
                 tokens per level do not average the rounding noise down - up to 1.2e-3 at 33 x 500 - are scored on TOPIQ's fp32
                 weights, Engine.topiq_f32_below; the saliency map's 1e-3 absolute error moves comp_score by 5e-6),
                 SAMP-Net in fp32 (its ResNet-18 trunk + pattern module in fp16 moves comp_score by 1.7e-3 - 4.3e-3), CLIP in
                 split-operand fp16 ('f16x3': plain fp16 GEMM operands - weights, LayerNorm outputs, GELU outputs - move the
                 aesthetic score by 1.4e-3 even with an fp32 token stream, tools/clip_rounding_sources.py; carried as fp16 pairs
                 hi + lo they leave 3.7e-4, at three times the matrix work of plain fp16 and still ~2x faster than fp32).
  FAST16         every model in fp16 with fp32 residual streams where they help (CLIP): embedding cosine >= 1 - 1e-6, scores within
                 5e-3 - tighter than the reference's own all-fp16 CLIP, but outside the 1e-3 gate.
  BF16           BASELINE.json configs[3] taken literally (bf16 storage): 2e-2 - 3e-2 on the scores.
"""
from ._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP

MODEL_IDS = {"topiq": FE_MODEL_TOPIQ, "clip": FE_MODEL_CLIP, "aesthetic": FE_MODEL_AESTHETIC, "u2netp": FE_MODEL_U2NETP,
             "samp_net": FE_MODEL_SAMP}

FP32 = {"topiq": "f32", "clip": "f32", "u2netp": "f32", "samp_net": "f32"}
REFERENCE_GPU = {"topiq": "f32", "clip": "f16", "u2netp": "f32", "samp_net": "f32"}
PARITY = {"topiq": "f16", "clip": "f16x3", "u2netp": "f16", "samp_net": "f32"}
FAST16 = {"topiq": "f16", "clip": "f16+r32", "u2netp": "f16", "samp_net": "f16+r32"}
BF16 = {"topiq": "bf16", "clip": "bf16", "u2netp": "bf16", "samp_net": "bf16"}
POLICIES = {"f32": FP32, "reference_gpu": REFERENCE_GPU, "parity": PARITY, "fast16": FAST16, "bf16": BF16,
            "f16": {"topiq": "f16", "clip": "f16", "u2netp": "f16", "samp_net": "f16"}}


def resolve(policy):
    """A policy name, a single precision name ('f16': every model) or a dict -> dict over the four model names."""
    if isinstance(policy, dict):
        return {**FP32, **policy}
    if policy in POLICIES:
        return dict(POLICIES[policy])
    return {k: policy for k in FP32}      # a plain precision name; Engine.set_precision rejects unknown ones


TOPIQ_F32_BELOW_PIXELS = 256 * 256


def load_models(engine, policy, state_dicts):
    """Commits `state_dicts` ({model name: {tensor name: array}}) on `engine`, each model under its precision of `policy`; the
    aesthetic MLP is always fp32. The context's default precision is left at fp32."""
    pol = resolve(policy)
    for name, sd in state_dicts.items():
        engine.set_precision(pol.get(name, "f32"))
        engine.load_weights(MODEL_IDS[name], sd)
    engine.set_precision("f32")
    # policies that claim the 1e-3 gate score small images on TOPIQ's fp32 weights (Engine.topiq_f32_below; include/facet_engine.h)
    engine.topiq_f32_below(TOPIQ_F32_BELOW_PIXELS if pol in (PARITY, REFERENCE_GPU) else 0)
    return pol


def describe(policy):
    pol = resolve(policy)
    return ", ".join(f"{k} {pol[k]}" for k in ("topiq", "u2netp", "samp_net", "clip"))

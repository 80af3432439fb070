"""Golden vectors for the VLM tagger's text decoder (SURVEY 8(f)-4 / BASELINE configs[4], slice 1) from the REFERENCE's own model class.

models/vlm_tagger.py:163-184 instantiates transformers' `Qwen2_5_VLForConditionalGeneration` and calls
`generate(**inputs, max_new_tokens=..., do_sample=False)` (:250-259, :355-360) in bfloat16 (:155-156). transformers is installed in the
build container (no checkpoint is: `from_pretrained` would download), so this script builds that class from a reduced-depth config
(Qwen2.5-VL's head_dim 128, grouped KV heads, q/k/v bias, M-RoPE sections 16/24/24, untied lm_head), loads the seeded synthetic
checkpoint of facet_amd/weights.py (`qwen2_5_vl_text_tiny`, regenerated from the seed on both sides, never committed), runs the same
greedy `generate` on seeded prompts in bfloat16, and stores prompts, generated token ids, the fp32 value of every step's bf16 logits
and their top-2 margins. Run in the build container:  python tests/golden/make_vlm_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from facet_amd.weights import synthetic_state_dict, VLM_TINY  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vlm_golden.npz")
MROPE = [16, 24, 24]
ROPE_THETA, RMS_EPS = 1000000.0, 1e-6


def planted_state_dict(seed):
    """The seeded checkpoint with a PLANTED read-out: lm_head row j = embed_tokens row perm[j] / 16. A random-init decoder's bf16 logits
    tie at the top every few steps (2048 near-Gaussian logits against a bf16 ulp of 2^-6: even transformers' own eager and sdpa attention
    paths then generate different tokens - `python tests/golden/make_vlm_golden.py --survey` prints it), so greedy token ids of such a
    model pin nothing. With the planted read-out the token whose embedding dominates the final hidden state wins by a wide margin, while
    every layer still runs at full random scale and moves the logits (which the logit tolerance of the parity test sees): the greedy
    path becomes a property of the arithmetic instead of its rounding noise."""
    sd = synthetic_state_dict("qwen2_5_vl_text_tiny", seed)
    perm = np.random.default_rng([seed, 77]).permutation(VLM_TINY["vocab"])
    sd["lm_head.weight"] = (sd["model.language_model.embed_tokens.weight"][perm] / 16.0).astype(np.float32)
    return sd


def build(seed, attn="sdpa", planted=True):
    from transformers import Qwen2_5_VLForConditionalGeneration, Qwen2_5_VLConfig
    c = VLM_TINY
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                         num_key_value_heads=c["kv_heads"], intermediate_size=c["inter"], vocab_size=c["vocab"], rms_norm_eps=RMS_EPS,
                         max_position_embeddings=4096, tie_word_embeddings=False, bos_token_id=None, eos_token_id=None, pad_token_id=None,
                         rope_parameters={"rope_theta": ROPE_THETA, "rope_type": "default", "mrope_section": MROPE}),
        vision_config=dict(depth=1, hidden_size=64, intermediate_size=128, num_heads=2, out_hidden_size=c["hidden"], patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[0]),
        bos_token_id=None, eos_token_id=None, pad_token_id=None)
    cfg._attn_implementation = attn
    m = Qwen2_5_VLForConditionalGeneration(cfg).eval()
    sd = planted_state_dict(seed) if planted else synthetic_state_dict("qwen2_5_vl_text_tiny", seed)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all("visual" in k for k in missing), (missing[:4], unexpected[:4])
    return m.to(torch.bfloat16)


def run(m, prompts, new_tokens):
    ids = torch.from_numpy(prompts)
    with torch.no_grad():
        out = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=new_tokens, do_sample=False,
                         output_logits=True, return_dict_in_generate=True, pad_token_id=0, eos_token_id=None)
    toks = out.sequences[:, prompts.shape[1]:].numpy().astype(np.int32)
    logits = torch.stack(out.logits, 1).float().numpy()        # [B][new][vocab], the bf16 logits widened
    top2 = np.sort(logits, -1)[..., -2:]
    return toks, logits, (top2[..., 1] - top2[..., 0])


def main():
    B, L, NEW = 2, 24, 40
    if "--survey" in sys.argv:      # the unplanted random-init decoder: ties at the top, attention-path dependent greedy tokens
        for seed in range(11, 17):
            prompts = np.random.default_rng(1000 + seed).integers(0, VLM_TINY["vocab"], (B, L)).astype(np.int64)
            toks, logits, margin = run(build(seed, planted=False), prompts, NEW)
            toks_e, _, _ = run(build(seed, "eager", planted=False), prompts, NEW)
            print(f"seed {seed}: min top-2 margin {margin.min():.4f} (logit scale {np.abs(logits).max():.2f}), eager == sdpa tokens: "
                  f"{bool(np.array_equal(toks, toks_e))}")
        return
    seed = 16
    prompts = np.random.default_rng(1000 + seed).integers(0, VLM_TINY["vocab"], (B, L)).astype(np.int64)
    toks, logits, margin = run(build(seed), prompts, NEW)
    toks_e, logits_e, _ = run(build(seed, "eager"), prompts, NEW)
    print(f"seed {seed}: min top-2 margin {margin.min():.3f}, logit scale {np.abs(logits).max():.2f}, {len(np.unique(toks))} distinct tokens, "
          f"eager == sdpa tokens: {bool(np.array_equal(toks, toks_e))}, max |logit eager - sdpa| {np.abs(logits - logits_e).max():.3f}")
    assert np.array_equal(toks, toks_e) and margin.min() > 1.0
    # second fixture: the UNPLANTED random-init checkpoint of the same seed, for teacher-forced logit comparison (the engine is fed
    # these tokens step by step): every step's bf16 logits (stored as their 16-bit patterns), in the regime where all layers decide
    toks_r, logits_r, margin_r = run(build(seed, planted=False), prompts, NEW)
    bits_r = (logits_r.view(np.uint32) >> 16).astype(np.uint16)
    assert np.array_equal((bits_r.astype(np.uint32) << 16).view(np.float32), logits_r)      # they are bf16 values
    print(f"unplanted: min top-2 margin {margin_r.min():.4f}, median {np.median(margin_r):.3f}, logit scale {np.abs(logits_r).max():.2f}")
    np.savez_compressed(OUT, seed_w=seed, prompts=prompts.astype(np.int32), tokens=toks, margin=margin.astype(np.float32),
                        logits_step0=logits[:, 0].astype(np.float32), logits_last=logits[:, -1].astype(np.float32),
                        top_logit=logits.max(-1).astype(np.float32), attn_impl_spread=np.float32(np.abs(logits - logits_e).max()),
                        random_tokens=toks_r, random_logits_bf16=bits_r, random_margin=margin_r.astype(np.float32),
                        mrope_section=np.asarray(MROPE, np.int32), rope_theta=np.float32(ROPE_THETA), rms_eps=np.float32(RMS_EPS),
                        **{k: np.int32(v) for k, v in VLM_TINY.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

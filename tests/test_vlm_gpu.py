"""GPU: the VLM tagger's text decoder (fe_vlm_prefill / fe_vlm_decode_step, SURVEY 8(f)-4 / BASELINE configs[4], slice 1) against vectors
of the REFERENCE's own model class - transformers' Qwen2_5_VLForConditionalGeneration, which models/vlm_tagger.py:163-184 instantiates -
generated in the build container by tests/golden/make_vlm_golden.py (greedy `generate(do_sample=False)` in bfloat16).

What is compared, and why two checkpoints:
  * the PLANTED-read-out checkpoint: free-running greedy generation, 2 prompts x 40 new tokens - the token ids must be IDENTICAL to the
    reference's, and the logits of the first and the last step within 0.25 (logit scale 23: two bf16 units; transformers' own sdpa and
    eager attention paths differ by 0.125 on these vectors). A random-init decoder's top logits tie in bf16 every few steps (then even
    transformers' two attention paths generate different tokens), so token ids of an unplanted checkpoint would pin nothing.
  * the UNPLANTED random checkpoint, teacher-forced with the reference's tokens: all 40 steps' logits within 0.0625 (scale 4.5: 2-4
    bf16 units), and the argmax equal wherever the reference's top-2 margin exceeds twice that - every layer decides these logits.
When transformers is importable on the test box, the same comparison also runs LIVE at other shapes (3 prompts of 37 tokens; one layer
at Qwen2.5-VL-7B's width: 28 heads over 4 KV heads, intermediate 18944).
"""
import os

import numpy as np
import pytest

from facet_amd._lib import FE_MODEL_VLM
from facet_amd.weights import synthetic_state_dict, qwen2_5_vl_text_spec, VLM_TINY

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vlm_golden.npz"))


def _bf16_bits_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def _planted(seed):
    sd = synthetic_state_dict("qwen2_5_vl_text_tiny", seed)
    perm = np.random.default_rng([seed, 77]).permutation(VLM_TINY["vocab"])
    sd["lm_head.weight"] = (sd["model.language_model.embed_tokens.weight"][perm] / 16.0).astype(np.float32)
    return sd


@pytest.fixture()
def vlm_engine():
    from facet_amd import Engine
    e = Engine(0, arena_bytes=8 << 30)
    e.vlm_configure(n_heads=VLM_TINY["heads"], n_kv_heads=VLM_TINY["kv_heads"], head_dim=128, rope_theta=float(G["rope_theta"]),
                    rms_eps=float(G["rms_eps"]), mrope_section=[int(v) for v in G["mrope_section"]])
    yield e
    e.close()


def test_greedy_token_ids_identical_to_the_reference_class(vlm_engine):
    e = vlm_engine
    e.load_weights(FE_MODEL_VLM, _planted(int(G["seed_w"])))
    d = e.vlm_dims()
    assert (d["vocab"], d["hidden"], d["layers"], d["heads"], d["kv_heads"], d["intermediate"]) == (2048, 512, 4, 4, 2, 1408)
    toks, logits = e.vlm_generate(G["prompts"], G["tokens"].shape[1], want_logits=True)
    err0, err1 = np.abs(logits[:, 0] - G["logits_step0"]).max(), np.abs(logits[:, -1] - G["logits_last"]).max()
    print(f"[vlm planted] tokens equal: {np.array_equal(toks, G['tokens'])}; logits |diff| step 0 {err0:.4f}, last step {err1:.4f} "
          f"(scale {np.abs(G['logits_step0']).max():.1f}, reference top-2 margin >= {G['margin'].min():.2f}, sdpa-vs-eager spread {float(G['attn_impl_spread']):.3f})")
    assert np.array_equal(toks, G["tokens"])
    assert err0 <= 0.25 and err1 <= 0.25
    assert np.abs(logits.max(-1) - G["top_logit"]).max() <= 0.25          # the winning logit of every step


def test_device_resident_generate_equals_the_stepwise_path(vlm_engine):
    """fe_vlm_generate (token ids / positions / cache length in device memory, one captured HIP graph of a decode step replayed, the
    streaming GEMV kernels of the small-batch decode) against the reference's tokens (planted checkpoint) and against the stepwise
    fe_vlm_decode_step path on the random checkpoint, for 1, 2, 3 and 5 sequences (GEMV row counts 1 / 2 / 4 and the shared GEMM)."""
    e = vlm_engine
    e.load_weights(FE_MODEL_VLM, _planted(int(G["seed_w"])))
    toks = e.vlm_generate(G["prompts"], G["tokens"].shape[1])
    assert np.array_equal(toks, G["tokens"])
    assert e.vlm_dims()["cur_len"] == G["prompts"].shape[1] + G["tokens"].shape[1] - 1
    rng = np.random.default_rng(4)
    for B in (1, 2, 3, 5):
        p = rng.integers(0, 2048, (B, 19)).astype(np.int32)
        fast = e.vlm_generate(p, 12)
        slow, _ = e.vlm_generate(p, 12, want_logits=True)
        assert np.array_equal(fast, slow), (B, fast, slow)
    # EOS handling of the wrapper: the row is padded with the EOS id from its first occurrence on
    p = G["prompts"][:1]
    full = e.vlm_generate(p, 10)
    cut = e.vlm_generate(p, 10, eos_token_ids=[int(full[0, 4])])
    assert np.array_equal(cut[0, :5], full[0, :5]) and (cut[0, 4:] == full[0, 4]).all()


def test_decode_kernels_agree_across_batch_sizes(vlm_engine):
    """The three decode projection paths - streaming GEMV (<= 3 sequences), weight-streaming matrix-core GEMM with a K split whose sums
    are taken by the fused residual / norm / SwiGLU passes (4 .. 32), the shared tiled GEMM (> 32) - on the same sequences: 3 prompts
    alone, the same 3 inside batches of 4, 8, 32 and 40. Planted read-out, so the token ids must be identical; logits within one bf16 unit
    of their scale (different summation orders of the same products)."""
    e = vlm_engine
    e.load_weights(FE_MODEL_VLM, _planted(7))
    rng = np.random.default_rng(12)
    p3 = rng.integers(0, 2048, (3, 21)).astype(np.int32)
    t3, l3 = e.vlm_generate(p3, 6, want_logits=True)
    for B in (4, 8, 32, 40):
        pb = np.concatenate([p3] + [rng.integers(0, 2048, (B - 3, 21)).astype(np.int32)], 0)
        tb, lb = e.vlm_generate(pb, 6, want_logits=True)
        assert np.array_equal(tb[:3], t3), B
        assert np.abs(lb[:3] - l3).max() <= 0.125, (B, float(np.abs(lb[:3] - l3).max()))
        assert np.array_equal(e.vlm_generate(pb, 6)[:3], t3), B          # device-resident loop, same kernels


def test_teacher_forced_logits_of_the_random_checkpoint(vlm_engine):
    e = vlm_engine
    e.load_weights(FE_MODEL_VLM, synthetic_state_dict("qwen2_5_vl_text_tiny", int(G["seed_w"])))
    ref = _bf16_bits_to_f32(G["random_logits_bf16"])                      # [2][40][2048]
    toks, logits = e.vlm_generate(G["prompts"], ref.shape[1], want_logits=True, forced_tokens=G["random_tokens"])
    diff = np.abs(logits - ref)
    tol = 0.0625
    decisive = G["random_margin"] > 2 * tol
    same = toks == G["random_tokens"]
    print(f"[vlm random, teacher-forced] max |logit diff| {diff.max():.4f} (mean {diff.mean():.5f}, logit scale {np.abs(ref).max():.2f}); "
          f"{int(decisive.sum())} of {decisive.size} steps have a reference margin > {2 * tol}: argmax equal on {int((same & decisive).sum())} of them; "
          f"equal on {int(same.sum())} of all {same.size}")
    assert diff.max() <= tol
    assert (same | ~decisive).all()


def test_decode_step_equals_prefill_of_the_longer_prompt(vlm_engine):
    """KV-cache consistency: prefill(L) + one decode step sees the same keys as prefill(L + 1) - different attention kernels (matrix-core
    tiles vs the grouped-query single-query pass over key chunks + merge), same logits up to bf16 rounding of the attention output."""
    e = vlm_engine
    e.load_weights(FE_MODEL_VLM, synthetic_state_dict("qwen2_5_vl_text_tiny", 5))
    rng = np.random.default_rng(3)
    # cache lengths on both sides of the decode attention's 128-key chunks (1 .. 5 chunks), with 3 sequences (GEMV projections) and 6
    # (matrix-core projections + fused finishing passes)
    for B, Ls in ((3, (1, 31, 32, 33, 127, 128, 129, 200, 300, 517)), (6, (5, 130, 300))):
        for L in Ls:
            p = rng.integers(0, 2048, (B, L + 1)).astype(np.int32)
            _, full = e.vlm_prefill(p, want_logits=True)
            e.vlm_prefill(p[:, :L], max_seq=L + 8)
            _, step = e.vlm_decode_step(p[:, L], np.full((3, B), L, np.int32), want_logits=True)
            assert np.abs(full - step).max() <= 0.0625, (B, L, float(np.abs(full - step).max()))


def _hf_model(cfg, sd, mrope, theta, eps):
    import torch
    from transformers import Qwen2_5_VLForConditionalGeneration, Qwen2_5_VLConfig
    c = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                         num_key_value_heads=cfg["kv_heads"], intermediate_size=cfg["inter"], vocab_size=cfg["vocab"], rms_norm_eps=eps,
                         max_position_embeddings=4096, tie_word_embeddings=False, bos_token_id=None, eos_token_id=None, pad_token_id=None,
                         rope_parameters={"rope_theta": theta, "rope_type": "default", "mrope_section": mrope}),
        vision_config=dict(depth=1, hidden_size=64, intermediate_size=128, num_heads=2, out_hidden_size=cfg["hidden"], patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[0]),
        bos_token_id=None, eos_token_id=None, pad_token_id=None)
    m = Qwen2_5_VLForConditionalGeneration(c).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all("visual" in k for k in missing)
    return m.to(torch.bfloat16)


@pytest.mark.parametrize("name,cfg,B,L,new", [
    ("tiny, 3 prompts of 37", VLM_TINY, 3, 37, 12),
    ("7B width, 1 layer", dict(hidden=3584, layers=1, heads=28, kv_heads=4, inter=18944, vocab=4096), 2, 150, 6),
])
def test_live_against_transformers_when_importable(vlm_engine, name, cfg, B, L, new):
    torch = pytest.importorskip("torch")
    pytest.importorskip("transformers")
    e = vlm_engine
    sd = synthetic_state_dict(None, 21, spec=qwen2_5_vl_text_spec(**cfg))
    e.vlm_configure(n_heads=cfg["heads"], n_kv_heads=cfg["kv_heads"], head_dim=128, rope_theta=1e6, rms_eps=1e-6, mrope_section=(16, 24, 24))
    e.load_weights(FE_MODEL_VLM, sd)
    m = _hf_model(cfg, sd, [16, 24, 24], 1e6, 1e-6)
    prompts = np.random.default_rng(9).integers(0, cfg["vocab"], (B, L)).astype(np.int64)
    with torch.no_grad():
        out = m.generate(input_ids=torch.from_numpy(prompts), attention_mask=torch.ones(B, L, dtype=torch.long), max_new_tokens=new, do_sample=False,
                         output_logits=True, return_dict_in_generate=True, pad_token_id=0, eos_token_id=None)
    ref_tok = out.sequences[:, L:].numpy()
    ref = torch.stack(out.logits, 1).float().numpy()
    toks, logits = e.vlm_generate(prompts, new, want_logits=True, forced_tokens=ref_tok)
    diff = np.abs(logits - ref)
    scale = np.abs(ref).max()
    tol = scale * 2.0 ** -6          # ~4 bf16 units of the largest logit
    top2 = np.sort(ref, -1)[..., -2:]
    decisive = (top2[..., 1] - top2[..., 0]) > 2 * tol
    print(f"[vlm live: {name}] max |logit diff| {diff.max():.4f} of scale {scale:.2f} (tol {tol:.4f}); argmax equal on "
          f"{int(((toks == ref_tok) & decisive).sum())} of {int(decisive.sum())} decisive steps")
    assert diff.max() <= tol and ((toks == ref_tok) | ~decisive).all()


# ---- slice 2: the vision tower and prompts with images ---------------------------------------------------------------------------------
GV = np.load(os.path.join(os.path.dirname(__file__), "golden", "vlm_vision_golden.npz"))


def _planted_full(seed):
    sd = synthetic_state_dict("qwen2_5_vl_tiny", seed)
    perm = np.random.default_rng([seed, 77]).permutation(VLM_TINY["vocab"])
    sd["lm_head.weight"] = (sd["model.language_model.embed_tokens.weight"][perm] / 16.0).astype(np.float32)
    return sd


def test_vision_tower_and_image_prompt_against_the_reference_class(vlm_engine):
    """tests/golden/make_vlm_vision_golden.py: two images (10x12 and 6x6 patches: ragged windows, a one-window image) through
    `model.visual` of the reference's class, and `generate` on a prompt that carries both. The merged image embeddings must agree within
    0.0625 (|max| 2.7; transformers' own sdpa and eager paths differ by 0.031), the M-RoPE position ids and the index arrays exactly
    (host restatements), the 16 greedy tokens exactly, first / last logits within 0.25."""
    from facet_amd.vlm_tagger import vision_indices, rope_index
    e = vlm_engine
    e.vlm_vision_configure(int(GV["vis_heads"]), [int(v) for v in GV["fullatt"]])
    e.load_weights(FE_MODEL_VLM, _planted_full(int(GV["seed_w"])))
    grid = GV["grid_thw"]
    n_patches = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
    pv = np.random.default_rng(int(GV["pixel_seed"])).normal(0, 1, (n_patches, 1176)).astype(np.float32)
    idx = vision_indices(grid)
    emb = e.vlm_encode_images(pv, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"])
    err = np.abs(emb - GV["embeds"]).max()
    print(f"[vlm vision] embeds {emb.shape}: max |diff| {err:.4f} (|max| {np.abs(GV['embeds']).max():.2f}, transformers sdpa-vs-eager {float(GV['attn_impl_spread_embeds']):.4f})")
    assert emb.shape == GV["embeds"].shape and err <= 0.0625
    ids = GV["input_ids"]
    pos, nxt = rope_index(ids, grid, int(GV["image_token_id"]))
    assert np.array_equal(pos, GV["position_ids"])
    rows = np.flatnonzero(ids.reshape(-1) == int(GV["image_token_id"])).astype(np.int32)
    toks, logits = e.vlm_generate(ids, GV["tokens"].shape[1], position_ids=pos, want_logits=True, image_rows=rows)
    e0, e1 = np.abs(logits[:, 0] - GV["logits_step0"]).max(), np.abs(logits[:, -1] - GV["logits_last"]).max()
    print(f"[vlm image prompt] tokens equal: {np.array_equal(toks, GV['tokens'])}; logits |diff| step 0 {e0:.4f}, last {e1:.4f} (transformers sdpa-vs-eager {float(GV['attn_impl_spread_logits']):.3f})")
    assert np.array_equal(toks, GV["tokens"]) and e0 <= 0.25 and e1 <= 0.25
    # the product path (device-resident decode loop) through the host mirror
    from facet_amd.vlm_tagger import VLMTagger
    t = VLMTagger({"model_path": "Qwen/Qwen2.5-VL-7B-Instruct", "max_new_tokens": GV["tokens"].shape[1]}, engine=e)
    t.model = e
    assert np.array_equal(t.generate_with_images(ids, pv, grid, int(GV["image_token_id"])), GV["tokens"])


def test_vision_tower_live_at_the_7b_width_when_transformers_imports(vlm_engine):
    """One block pair at Qwen2.5-VL-7B's vision width (1280 = 16 heads of 80, intermediate 3420, merger 5120 -> 512) on a 34 x 46-patch
    image (ragged windows on both axes, a 1564-patch full-attention segment) against transformers' vision class itself."""
    torch = pytest.importorskip("torch")
    pytest.importorskip("transformers")
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig
    from facet_amd.weights import qwen2_5_vl_vision_spec
    from facet_amd.vlm_tagger import vision_indices
    vc = dict(hidden=1280, depth=2, inter=3420, out_hidden=512)
    sd = synthetic_state_dict(None, 23, spec=qwen2_5_vl_text_spec(**VLM_TINY) + qwen2_5_vl_vision_spec(**vc))
    e = vlm_engine
    e.vlm_vision_configure(16, [1])
    e.load_weights(FE_MODEL_VLM, sd)
    cfg = Qwen2_5_VLVisionConfig(depth=2, hidden_size=1280, intermediate_size=3420, num_heads=16, out_hidden_size=512, patch_size=14, spatial_merge_size=2,
                                 temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1])
    m = Qwen2_5_VisionTransformerPretrainedModel(cfg).eval()
    missing, unexpected = m.load_state_dict({k[len("model.visual."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("model.visual.")}, strict=False)
    assert not missing and not unexpected
    m = m.to(torch.bfloat16)
    grid = np.array([[1, 34, 46]], np.int64)
    pv = np.random.default_rng(3).normal(0, 1, (34 * 46, 1176)).astype(np.float32)
    with torch.no_grad():
        ref = m(torch.from_numpy(pv).to(torch.bfloat16), grid_thw=torch.from_numpy(grid)).pooler_output.float().numpy()
    idx = vision_indices(grid)
    emb = e.vlm_encode_images(pv, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"])
    scale = np.abs(ref).max()
    print(f"[vlm vision live, 7B width] max |diff| {np.abs(emb - ref).max():.4f} of |max| {scale:.2f}")
    assert np.abs(emb - ref).max() <= scale * 2.0 ** -5

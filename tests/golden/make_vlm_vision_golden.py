"""Golden vectors for the VLM tagger's VISION path (SURVEY 8(f)-4 / BASELINE configs[4], slice 2) from the reference's own model class.

models/vlm_tagger.py feeds `processor(text=..., images=...)` - pixel patches `pixel_values [n_patches, 1176]` and `image_grid_thw` - into
`Qwen2_5_VLForConditionalGeneration.generate` (:245-259, :346-360). This script builds that class from a reduced config (vision tower: 3
blocks of 2 heads x 80 - Qwen2.5-VL's head_dim -, window attention 112 px with one full-attention block, SwiGLU, 2x2 patch merger; text
decoder as make_vlm_golden.py), loads the seeded checkpoint `qwen2_5_vl_tiny` of facet_amd/weights.py with the planted read-out (see
make_vlm_golden.py for why), and stores for two images of different grids (10x12 and 6x6 patches: ragged windows, a one-window image):
the merged image embeddings (`model.visual(...).pooler_output`, bf16), the M-RoPE position ids transformers computed for the prompt
(captured at the decoder's input), the greedy continuation and its first / last logits. Run in the build container:
    python tests/golden/make_vlm_vision_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from facet_amd.weights import synthetic_state_dict, VLM_TINY, VLM_VISION_TINY  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vlm_vision_golden.npz")
IMG, VSTART, VEND = 2000, 2002, 2003
VIS_HEADS, FULLATT = 2, [1]


def planted(seed):
    sd = synthetic_state_dict("qwen2_5_vl_tiny", seed)
    perm = np.random.default_rng([seed, 77]).permutation(VLM_TINY["vocab"])
    sd["lm_head.weight"] = (sd["model.language_model.embed_tokens.weight"][perm] / 16.0).astype(np.float32)
    return sd


def build(seed, attn="sdpa"):
    from transformers import Qwen2_5_VLForConditionalGeneration, Qwen2_5_VLConfig
    c, v = VLM_TINY, VLM_VISION_TINY
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"], num_key_value_heads=c["kv_heads"],
                         intermediate_size=c["inter"], vocab_size=c["vocab"], rms_norm_eps=1e-6, max_position_embeddings=4096,
                         tie_word_embeddings=False, bos_token_id=None, eos_token_id=None, pad_token_id=None,
                         rope_parameters={"rope_theta": 1000000.0, "rope_type": "default", "mrope_section": [16, 24, 24]}),
        vision_config=dict(depth=v["depth"], hidden_size=v["hidden"], intermediate_size=v["inter"], num_heads=VIS_HEADS, out_hidden_size=v["out_hidden"],
                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=FULLATT),
        image_token_id=IMG, video_token_id=2001, vision_start_token_id=VSTART, vision_end_token_id=VEND, bos_token_id=None, eos_token_id=None,
        pad_token_id=None)
    cfg._attn_implementation = attn
    cfg.vision_config._attn_implementation = attn
    m = Qwen2_5_VLForConditionalGeneration(cfg).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(val) for k, val in planted(seed).items()}, strict=False)
    assert not unexpected and not missing, (missing[:4], unexpected[:4])
    return m.to(torch.bfloat16)


def main():
    seed, NEW = 16, 16
    grid = np.array([[1, 10, 12], [1, 6, 6]], np.int64)
    n_patches = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
    pv = np.random.default_rng(5).normal(0, 1, (n_patches, 1176)).astype(np.float32)
    n_img = [int(g[1] * g[2] // 4) for g in grid]
    ids = np.array([[5, 6, VSTART] + [IMG] * n_img[0] + [VEND, 7, 8, 9, VSTART] + [IMG] * n_img[1] + [VEND, 11, 12]], np.int64)
    res = {}
    for attn in ("sdpa", "eager"):
        m = build(seed, attn)
        got = {}
        def grab(mod, args, kwargs):
            if kwargs.get("position_ids") is not None and "pos" not in got:
                got["pos"] = kwargs["position_ids"].clone()
        hook = m.model.language_model.register_forward_pre_hook(grab, with_kwargs=True)
        mm = torch.from_numpy((ids == IMG).astype(np.int32))      # what the processor hands over: 0 text, 1 image tokens
        with torch.no_grad():
            emb = m.model.visual(torch.from_numpy(pv).to(torch.bfloat16), grid_thw=torch.from_numpy(grid)).pooler_output
            out = m.generate(input_ids=torch.from_numpy(ids), attention_mask=torch.ones(1, ids.shape[1], dtype=torch.long), pixel_values=torch.from_numpy(pv),
                             image_grid_thw=torch.from_numpy(grid), mm_token_type_ids=mm, max_new_tokens=NEW, do_sample=False, output_logits=True,
                             return_dict_in_generate=True, pad_token_id=0, eos_token_id=None)
        hook.remove()
        pos = got["pos"]
        pos = pos[-3:] if pos.shape[0] == 4 else pos          # (text, t, h, w) in newer transformers: the rotary ids are the last three
        res[attn] = dict(emb=emb.float().numpy(), toks=out.sequences[:, ids.shape[1]:].numpy().astype(np.int32),
                         logits=torch.stack(out.logits, 1).float().numpy(), pos=pos.numpy().astype(np.int32))
    a, b = res["sdpa"], res["eager"]
    top2 = np.sort(a["logits"], -1)[..., -2:]
    print(f"embeds {a['emb'].shape} |max| {np.abs(a['emb']).max():.3f}, sdpa vs eager: embeds {np.abs(a['emb'] - b['emb']).max():.4f}, logits {np.abs(a['logits'] - b['logits']).max():.4f}, "
          f"tokens equal {np.array_equal(a['toks'], b['toks'])}, min margin {(top2[..., 1] - top2[..., 0]).min():.2f}, positions {a['pos'].shape} max {a['pos'].max()}")
    assert np.array_equal(a["toks"], b["toks"])
    np.savez_compressed(OUT, seed_w=seed, grid_thw=grid.astype(np.int32), pixel_seed=5, input_ids=ids.astype(np.int32), position_ids=a["pos"],
                        embeds=a["emb"].astype(np.float32), tokens=a["toks"], logits_step0=a["logits"][:, 0].astype(np.float32),
                        logits_last=a["logits"][:, -1].astype(np.float32), attn_impl_spread_embeds=np.float32(np.abs(a["emb"] - b["emb"]).max()),
                        attn_impl_spread_logits=np.float32(np.abs(a["logits"] - b["logits"]).max()), vis_heads=np.int32(VIS_HEADS),
                        fullatt=np.asarray(FULLATT, np.int32), image_token_id=np.int32(IMG))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""CPU: the host side of the VLM tagger mirror (facet_amd/vlm_tagger.py) against what the reference's own VLMTagger returned in the
build container (tests/golden/make_vlm_host_golden.py -> vlm_host_golden.json): prompt text, model-family detection, batch sizes,
edit distance and the parsing of generated text into vocabulary tags."""
import json
import os

import pytest

from facet_amd.vlm_tagger import VLMTagger, edit_distance

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vlm_host_golden.json")))


class _Cfg:
    def __init__(self):
        self.cats = [{"name": "scene_type", "tags": {"landscape": ["scenery"], "portrait": [], "street": [], "architecture": []}},
                     {"name": "subject", "tags": {"person": [], "animal": [], "portrait": [], "black_and_white": []}},
                     {"name": "empty", "tags": {}},
                     {"name": "mood", "tags": {"dramatic": [], "peaceful": [], "long_exposure": []}}]
        self.config = {"standalone_tags": {"sunset": ["dusk"], "person": [], "macro": []}}

    def get_categories(self):
        return self.cats

    def get_tag_vocabulary(self):
        v = {}
        for c in self.cats:
            v.update(c["tags"])
        v.update(self.config["standalone_tags"])
        return v


def test_edit_distance_matches_the_reference():
    for a, b, d in G["levenshtein"]:
        assert edit_distance(a, b) == d and edit_distance(b, a) == d, (a, b)


@pytest.mark.parametrize("name", ["with_config", "no_config"])
def test_prompt_and_tag_parsing_match_the_reference(name):
    g = G[name]
    t = VLMTagger({"model_path": "Qwen/Qwen2.5-VL-7B-Instruct"}, _Cfg() if name == "with_config" else None)
    assert t.family == g["family"] and t.batch_size == g["batch_size"] and sorted(t.valid_tags) == g["valid_tags"]
    assert t._build_prompt() == g["prompt"]
    assert t._build_prompt() is t._build_prompt()              # cached
    for text, m, want in g["parse"]:
        assert t._parse_tags(text, m) == want, (text, m)


def test_family_detection_and_batch_size():
    q3 = VLMTagger({"model_path": "Qwen/Qwen3-VL-2B-Instruct"})
    assert q3.family == G["qwen3_family"]["family"] and q3.batch_size == G["qwen3_family"]["batch_size"]
    assert VLMTagger({"model_path": "x", "vlm_batch_size": 7}).batch_size == G["qwen3_family"]["custom_batch"]
    with pytest.raises(RuntimeError):
        VLMTagger({"model_path": "Qwen/Qwen2.5-VL-7B-Instruct"}).generate_ids([[1, 2, 3]])      # not loaded: fails loudly


def test_rope_index_and_vision_indices_match_the_reference():
    """The index arithmetic the host mirror restates from transformers (M-RoPE position ids of a prompt with two images; window order and
    segment bounds of the vision tower) against what the reference's class computed (tests/golden/make_vlm_vision_golden.py) and, when
    transformers is importable, against its utilities for more grids."""
    import numpy as np
    from facet_amd.vlm_tagger import vision_indices, rope_index
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vlm_vision_golden.npz"))
    pos, nxt = rope_index(g["input_ids"], g["grid_thw"], int(g["image_token_id"]))
    assert np.array_equal(pos, g["position_ids"]) and nxt.tolist() == [int(g["position_ids"].max()) + 1]
    idx = vision_indices(g["grid_thw"])
    n = int((g["grid_thw"][:, 0] * g["grid_thw"][:, 1] * g["grid_thw"][:, 2]).sum())
    assert idx["patch_pos_hw"].shape == (n, 2) and sorted(idx["window_index"].tolist()) == list(range(n // 4))
    assert idx["cu_window_seqlens"][0] == 0 and idx["cu_window_seqlens"][-1] == n and (np.diff(idx["cu_window_seqlens"]) > 0).all()
    assert idx["cu_seqlens"].tolist() == [0, 120, 156]
    with pytest.raises(ValueError):
        rope_index(g["input_ids"][:, :20], g["grid_thw"], int(g["image_token_id"]))      # a truncated placeholder run
    try:
        import torch
        import transformers.vision_utils as vu
    except Exception:
        return
    for grid in ([[1, 16, 16]], [[1, 8, 24], [1, 22, 18], [1, 2, 2]], [[1, 34, 46]]):
        gt = torch.tensor(grid)
        got = vision_indices(grid)
        wi, cw = vu.get_vision_window_index(gt, 2, 112, 14)
        p = vu.get_vision_position_ids(gt, 2)
        k = p.shape[0]
        assert np.array_equal(got["window_index"], wi.numpy()) and np.array_equal(got["cu_window_seqlens"], cw.numpy())
        assert np.array_equal(got["patch_pos_hw"], p.reshape(k // 4, 4, 2)[wi].reshape(k, 2).numpy())
        assert np.array_equal(got["cu_seqlens"], vu.get_vision_cu_seqlens(gt).numpy())

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

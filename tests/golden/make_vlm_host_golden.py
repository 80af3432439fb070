"""Generates tests/golden/vlm_host_golden.json by RUNNING THE REFERENCE'S OWN VLMTagger host logic in the build container
(models/vlm_tagger.py: `_levenshtein` :29-42, `_build_prompt` / `_fallback_prompt` :88-148, `_parse_tags` :446-495). No model is loaded:
these methods are pure Python. Inputs are written here; only what the reference returned is stored.

    python tests/golden/make_vlm_host_golden.py
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference")
from models.vlm_tagger import VLMTagger, _levenshtein  # noqa: E402


class FakeScoringConfig:
    """The three accessors VLMTagger uses (get_tag_vocabulary, get_categories, .config['standalone_tags'])."""

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


TEXTS = [
    "landscape, portrait, street",
    "Tags: Landscape, PORTRAIT , dramatic",
    "Here are the tags: landscap, portrat, peacefull, xyzzy",
    "tags: 1. landscape, 2) person, - animal, \"sunset\", 'macro'",
    "Scene: landscape, Subject: person, Mood: long exposure",
    "black and white, long exposure, black_and_white",
    "a, , b, street,street, STREET",
    "The tags are: architecture,architectur,architectures,arch",
    "",
    "sunsets, macros, peacful, dramatc, prson, animl, landscapee, portraitt",
    "Tags: tags: landscape",
    "night, astro, food, something entirely different, dramatic: peaceful",
]

out = {"levenshtein": [[a, b, _levenshtein(a, b)] for a, b in
                       [("", ""), ("a", ""), ("", "abc"), ("kitten", "sitting"), ("flaw", "lawn"), ("landscape", "landscap"),
                        ("portrait", "portrat"), ("long_exposure", "long exposure"), ("abc", "abc"), ("abc", "cba"), ("street", "streets")]]}
for name, cfg in (("with_config", FakeScoringConfig()), ("no_config", None)):
    t = VLMTagger({"model_path": "Qwen/Qwen2.5-VL-7B-Instruct"}, cfg)
    if cfg is not None:
        # set iteration order of valid_tags decides ties between equally distant vocabulary tags: pin it by storing the reference's
        # answers only for inputs whose best match is unique (checked here)
        for text in TEXTS:
            for piece in text.split(","):
                tag = piece.strip().lower().lstrip("0123456789.-) ").strip("\"'")
                if ":" in tag:
                    tag = tag.split(":", 1)[1].strip()
                tag = tag.replace(" ", "_")
                if len(tag) > 1 and tag not in t.valid_tags:
                    ds = sorted(_levenshtein(tag, v) for v in t.valid_tags)
                    assert not (ds[0] <= 2 and ds[1] == ds[0]), (text, tag, ds[:3])
    out[name] = {"family": t.family, "batch_size": t.batch_size, "valid_tags": sorted(t.valid_tags), "prompt": t._build_prompt(),
                 "parse": [[text, m, t._parse_tags(text, m)] for text in TEXTS for m in (5, 2, 50)]}
out["qwen3_family"] = {"family": VLMTagger({"model_path": "Qwen/Qwen3-VL-2B-Instruct"}).family,
                       "batch_size": VLMTagger({"model_path": "Qwen/Qwen3-VL-2B-Instruct"}).batch_size,
                       "custom_batch": VLMTagger({"model_path": "x", "vlm_batch_size": 7}).batch_size}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vlm_host_golden.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, os.path.getsize(path), "bytes")

"""Host-side mirrors against outputs of the REFERENCE'S OWN functions (tests/golden/host_golden.json, produced in the build
container by tests/golden/make_host_golden.py, which imports /root/reference; the reference is not available on the GPU box).
Pinned here: PyIQAScorer._normalize_score / _preprocess_image, SAMPNetScorer post-processing, CLIPTagger selection, EAR. CPU only."""
import hashlib
import json
import os
import types

import numpy as np
import pytest
from PIL import Image

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_golden.json")))


def test_normalize_score_matches_reference():
    from facet_amd.pyiqa_scorer import PyIQAScorer
    s = PyIQAScorer("topiq", device="cpu")
    for raw, want in zip(G["normalize_score"]["raw"], G["normalize_score"]["out"]):
        raw = float("nan") if raw is None else raw
        if isinstance(want, str):
            with pytest.raises(Exception):
                s._normalize_score(raw)
            continue
        got = s._normalize_score(raw)
        if want is None:
            assert got != got
        else:
            assert got == pytest.approx(want, abs=1e-12), raw


def test_preprocess_image_matches_reference_bytes():
    """Same seeded images; the mirror keeps uint8 HWC (the /255 + CHW happen on the GPU), so apply exactly those two steps on the
    host and compare the float32 NCHW bytes with what the reference's tensor held."""
    from facet_amd.pyiqa_scorer import PyIQAScorer
    s = PyIQAScorer("topiq", device="cpu")
    rng = np.random.default_rng(17)
    for case in G["preprocess_image"]:
        ch = {"RGB": 3, "L": 1, "RGBA": 4}[case["mode"]]
        arr = rng.integers(0, 256, (case["h"], case["w"], ch) if ch > 1 else (case["h"], case["w"]), dtype=np.uint8)
        u8 = s._preprocess_image(Image.fromarray(arr, case["mode"]))
        t = np.ascontiguousarray((u8.astype(np.float32) / 255.0).transpose(2, 0, 1)[None])
        assert list(t.shape) == case["shape"] and str(t.dtype) == case["dtype"]
        assert hashlib.sha256(t.tobytes()).hexdigest() == case["sha256"], (case["h"], case["w"], case["mode"])


def test_samp_postprocess_matches_reference():
    from facet_amd.samp_net import postprocess
    g = G["samp_postprocess"]
    for pw, at, sd, want in zip(g["pattern_logits"], g["attributes"], g["score_dist"], g["dicts"]):
        got = postprocess(np.asarray(pw, np.float32), np.asarray(at, np.float32), np.asarray(sd, np.float32))
        assert set(got) == set(want)
        for k in ("comp_score", "raw_score", "pattern", "pattern_index", "power_point_score"):
            assert got[k] == want[k], k
        assert got["score_distribution"] == want["score_distribution"] and got["attributes"] == want["attributes"]
        assert list(got["pattern_weights"]) == list(want["pattern_weights"])
        for k, v in want["pattern_weights"].items():
            assert got["pattern_weights"][k] == pytest.approx(v, rel=2e-6, abs=1e-9)


def test_tagger_selection_matches_reference():
    from facet_amd.tagger import CLIPTagger
    g = G["tagger"]
    cfg = types.SimpleNamespace(get_tag_vocabulary=lambda: g["vocabulary"], get_art_tags=lambda: set(g["art_tags"]))
    t = CLIPTagger(clip_model=None, device="cpu", config=cfg)
    assert t.get_tags_from_embedding(b"\0" * 3072) == [] and t.get_tags_from_embedding(None) == g["none_bytes"] == []
    names = [tag for tag, descs in g["vocabulary"].items() for _ in descs]
    assert t.prompts()[0] == names
    t.set_text_embeddings(names, np.asarray(g["text_embeddings"], np.float32))
    for c in g["cases"]:
        b = np.asarray(c["embedding"], np.float32).tobytes()
        assert t.get_tags_from_embedding(b) == c["tags_default"]
        assert t.get_tags_from_embedding(b, threshold=0.22, max_tags=5) == c["tags_t22_m5"]
        assert t.get_tags_from_embedding(b, threshold=0.05, max_tags=3) == c["tags_t05_m3"]
        ws = t.get_tags_with_scores(b, threshold=0.1)
        assert set(ws) == set(c["with_scores"])
        for k, v in c["with_scores"].items():
            assert ws[k] == pytest.approx(v, abs=1.1e-3)
        assert t.is_artwork(b, threshold=0.2) == c["is_artwork"]


def test_ear_matches_reference():
    from facet_amd.face import FaceAnalyzer
    for c in G["ear"]:
        lm = np.asarray(c["landmarks"], np.float32)
        assert FaceAnalyzer.calculate_ear(lm, FaceAnalyzer.LEFT_EYE_INDICES) == pytest.approx(c["left"], rel=1e-6)
        assert FaceAnalyzer.calculate_ear(lm, FaceAnalyzer.RIGHT_EYE_INDICES) == pytest.approx(c["right"], rel=1e-6)
        assert FaceAnalyzer.compute_avg_ear(lm) == pytest.approx(c["avg"], rel=1e-6)


def test_model_manager_sizing_matches_reference():
    from facet_amd.model_manager import ModelManager
    g = G["model_manager"]
    mm = ModelManager.__new__(ModelManager)       # sizing helpers need no engine
    mm.config = None
    for name, gb in g["vram_gb"].items():
        assert mm.get_model_vram(name) == gb
    for v, prof in g["profiles"].items():
        assert ModelManager.get_recommended_profile(float(v)) == prof
    for v, model in g["quality"].items():
        assert mm.select_quality_model(float(v)) == model, v
    for c in g["packs"]:
        assert mm.group_passes_by_vram(list(c["models"]), c["vram"]) == c["passes"], (c["models"], c["vram"])


def test_placement_data_matches_reference():
    from facet_amd.batch import placement_data
    for c in G["placement"]:
        got = placement_data(None if c["bbox"] is None else np.array(c["bbox"]), c["w"], c["h"], *c["weights"])
        assert got == c["out"], (c["bbox"], got, c["out"])


def test_detect_silhouette_matches_reference():
    from facet_amd.batch import detect_silhouette
    for c in G["silhouette"]:
        assert detect_silhouette(c["hist"], c["tags"], c["faces"]) == c["out"], c


def test_leading_line_scoring_matches_reference():
    """Segment scoring of detect_leading_lines and integrate_leading_lines (analyzers/composition.py:231-283), golden made with the
    reference's function and its cv2 calls mocked to return the stored segments."""
    from facet_amd.composition import CompositionAnalyzer, score_lines
    for c in G["leading_lines_scoring"]:
        got = score_lines(None if c["lines"] is None else np.array(c["lines"], np.int32), c["h"], c["w"])
        assert float(got["leading_lines_score"]) == c["out"]["leading_lines_score"] and got["line_count"] == c["out"]["line_count"], c
    for c in G["integrate_leading_lines"]:
        assert CompositionAnalyzer.integrate_leading_lines(c["base"], c["lines"], c["faces"]) == c["out"]

"""CPU: host-side mirror of the reference's PyIQAScorer interface (no engine calls)."""
import numpy as np
import pytest
from PIL import Image

from facet_amd.pyiqa_scorer import PyIQAScorer


def test_unknown_model_raises_valueerror():
    with pytest.raises(ValueError, match="Unknown model"):
        PyIQAScorer("nope")  # reference models/pyiqa_scorer.py:88-90


def test_interface_attrs():
    s = PyIQAScorer("topiq", device="cuda")
    assert s.model_name == "topiq" and s.vram_gb == 2 and "TOPIQ" in s.description and s.model is None


def test_preprocess_caps_long_edge_with_lanczos():
    s = PyIQAScorer("topiq")
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (600, 2048, 3), dtype=np.uint8))
    a = s._preprocess_image(img)
    assert a.shape == (int(600 * 0.5), 1024, 3) and a.dtype == np.uint8  # int(w*s), int(h*s), reference :150-153
    small = Image.fromarray(np.zeros((100, 120, 3), np.uint8)).convert("L")
    assert s._preprocess_image(small).shape == (100, 120, 3)  # mode converted to RGB, no resize <= 1024


def test_normalize_score_clamps_and_scales():
    s = PyIQAScorer("topiq")
    assert s._normalize_score(0.42) == pytest.approx(4.2)
    assert s._normalize_score(np.float32(1.5)) == 10.0 and s._normalize_score(-3) == 0.0
    assert isinstance(s._normalize_score(0.5), float)


def test_samp_postprocess_matches_oracle_restatement():
    """facet_amd.samp_net.postprocess (product) == oracle.sampnet.samp_postprocess (pinned to samp_net.py:957-989)."""
    import os
    from facet_amd.samp_net import postprocess, COMPOSITION_PATTERNS
    from oracle.sampnet import samp_postprocess
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "samp_golden.npz"))
    for i in range(2):
        a = postprocess(g["pattern_weights"][i], g["attributes"][i], g["score_dist"][i])
        b = samp_postprocess(g["pattern_weights"][i], g["attributes"][i], g["score_dist"][i])
        assert a["pattern"] == b["pattern"] and a["comp_score"] == b["comp_score"] and a["raw_score"] == b["raw_score"]
        assert a["power_point_score"] == b["power_point_score"] and list(a) == list(b)
        assert all(abs(a["pattern_weights"][k] - b["pattern_weights"][k]) < 1e-6 for k in COMPOSITION_PATTERNS)
    assert len(COMPOSITION_PATTERNS) == 8  # reference samp_net.py:23-32 (not the validator's 14-name list)


def test_samp_input_conventions():
    from facet_amd.samp_net import SAMPNetScorer
    bgr = np.zeros((10, 12, 3), np.uint8)
    a, is_bgr = SAMPNetScorer._to_array(bgr)
    assert is_bgr and a.shape == (10, 12, 3)           # ndarray = OpenCV BGR (reference :916-921)
    a, is_bgr = SAMPNetScorer._to_array(Image.fromarray(bgr).convert("L"))
    assert not is_bgr and a.shape == (10, 12, 3)       # PIL converted to RGB
    with pytest.raises(ValueError):
        SAMPNetScorer._to_array(3.14)


def test_clip_preprocess_matches_open_clip_transform_shape_and_values():
    from facet_amd.clip import clip_preprocess, CLIP_MEAN, CLIP_STD
    img = Image.fromarray(np.full((300, 500, 3), 128, np.uint8))
    t = clip_preprocess(img)
    assert tuple(t.shape) == (3, 224, 224)
    assert np.allclose(t.numpy()[:, 0, 0], (128 / 255.0 - CLIP_MEAN) / CLIP_STD, atol=1e-6)


class _Cfg:
    def get_tag_vocabulary(self):
        return {"dog": ["dog", "puppy"], "cat": ["cat"], "painting": ["painting"]}

    def get_art_tags(self):
        return {"painting"}


def test_tagger_selection_semantics():
    """Reference tagger.py:77-114: per-tag max over synonyms, threshold, sort desc, top max_tags; [] without text embeddings."""
    from facet_amd.tagger import CLIPTagger
    tg = CLIPTagger(None, "cuda", _Cfg())
    emb = np.zeros(768, np.float32); emb[0] = 1.0
    assert tg.get_tags_from_embedding(emb.tobytes()) == [] and tg.get_tags_with_scores(emb.tobytes()) == {}
    names, texts = tg.prompts()
    assert names == ["dog", "dog", "cat", "painting"] and texts[1] == "a photo of puppy"
    te = np.zeros((4, 768), np.float32)
    te[0, 0], te[0, 1] = 0.3, 0.954      # dog/"dog"     sim 0.30
    te[1, 0], te[1, 1] = 0.6, 0.8        # dog/"puppy"   sim 0.60  (max wins)
    te[2, 0], te[2, 1] = 0.26, 0.9656    # cat           sim 0.26
    te[3, 0], te[3, 1] = 0.1, 0.995      # painting      sim 0.10 (below threshold)
    tg.set_text_embeddings(names, te)
    assert tg.get_tags_from_embedding(emb.tobytes(), threshold=0.25, max_tags=5) == ["dog", "cat"]
    assert tg.get_tags_from_embedding(emb.tobytes(), threshold=0.25, max_tags=1) == ["dog"]
    assert tg.get_tags_from_embedding(None) == []
    assert not tg.is_artwork(emb.tobytes()) and tg.is_artwork(emb.tobytes(), threshold=0.05)
    assert abs(tg.get_tags_with_scores(emb.tobytes())["dog"] - 0.6) < 2e-3


def test_model_manager_pass_packing_and_profiles():
    from facet_amd.model_manager import ModelManager
    mm = ModelManager(config=None, engine=object())
    # first-fit decreasing under (vram - 1): 4+2 fit in 7, remaining 2+2 in the next bin... (reference :768-814)
    assert mm.group_passes_by_vram(["topiq", "clip", "samp_net", "insightface"], 8) == [["clip", "topiq"], ["samp_net", "insightface"]]
    assert mm.group_passes_by_vram(["topiq", "clip", "samp_net"], 24) == [["clip", "topiq", "samp_net"]]
    assert mm.group_passes_by_vram(["clip", "topiq"], 4) == [["clip"], ["topiq"]]
    assert [mm.get_recommended_profile(v) for v in (288, 16, 8, 2)] == ["24gb", "16gb", "8gb", "legacy"]
    assert mm.select_quality_model(24) == "topiq" and mm._cache_hits == 0 and mm._cache_misses == 0
    assert mm.load_model_only("nope") is None and mm._cache_misses == 1   # failure -> None, caller raises (multi_pass.py:344-348)


def test_face_analyzer_mirror_unavailable_contract_and_ear():
    """No detector in the engine yet: the mirror behaves like the reference when InsightFace is missing (face.py:90-97),
    and the EAR arithmetic (face.py:241-256) is exact on hand-made landmarks."""
    from facet_amd.face import FaceAnalyzer
    fa = FaceAnalyzer(min_confidence=0.65, blink_ear_threshold=0.28)
    assert fa.available is False
    r = fa.analyze_faces(np.zeros((64, 64, 3), np.uint8))
    assert r["face_count"] == 0 and r["bbox"] is None and r["face_details"] == [] and len(r) == 11
    lm = np.zeros((106, 2), np.float32)
    for idx in (FaceAnalyzer.LEFT_EYE_INDICES, FaceAnalyzer.RIGHT_EYE_INDICES):
        outer, inner, u1, u2, l1, l2 = idx
        lm[outer] = (0, 0); lm[inner] = (10, 0)          # eye width 10
        lm[u1] = (3, 2); lm[l1] = (3, -1)                # vertical 3
        lm[u2] = (7, 1.5); lm[l2] = (7, -1.5)            # vertical 3
    assert FaceAnalyzer.compute_avg_ear(lm) == pytest.approx(0.3)
    assert FaceAnalyzer.calculate_ear(np.zeros((106, 2)), FaceAnalyzer.LEFT_EYE_INDICES) == 0.3   # degenerate width
    face = type("F", (), {"landmark_2d_106": lm})()
    assert not fa.is_blinking(face)
    lm2 = lm.copy(); lm2[:, 1] *= 0.5
    assert fa.is_blinking(type("F", (), {"landmark_2d_106": lm2})())
    assert not fa.is_blinking(object())


def test_wrappers_refuse_to_score_with_made_up_weights(monkeypatch):
    """No checkpoint path and no explicit opt-in -> FileNotFoundError, never a silent synthetic checkpoint (the reference downloads
    the weights or fails: models/pyiqa_scorer.py:108, model_manager.py:140, processing/scorer.py:560-577)."""
    import pytest
    from facet_amd.pyiqa_scorer import PyIQAScorer
    from facet_amd.clip import load_clip, ClipAestheticScorer
    from facet_amd.weights import checkpoint_or_synthetic
    monkeypatch.delenv("FACET_AMD_SYNTHETIC", raising=False)
    monkeypatch.delenv("FACET_AMD_TOPIQ_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError, match="topiq"):
        PyIQAScorer("topiq").load()
    with pytest.raises(FileNotFoundError, match="clip"):
        load_clip(engine=object())
    with pytest.raises(FileNotFoundError, match="aesthetic"):
        ClipAestheticScorer(object(), {"model": None, "preprocess": None})
    # the opt-ins
    assert "h_emb" in checkpoint_or_synthetic("topiq", None, True, 3, None)
    monkeypatch.setenv("FACET_AMD_SYNTHETIC", "1")
    assert "0.weight" in checkpoint_or_synthetic("aesthetic", None, False, 3, None)

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

"""facet_amd/aggregate.py against outputs of the reference's own `Facet.calculate_aggregate_logic` (tests/golden/aggregate_golden.json,
made by tests/golden/make_aggregate_golden.py on configurations and rows defined there). Scores must be the same float64."""
import json
import os

import numpy as np
import pytest

from facet_amd.aggregate import AggregatePolicy, aggregate, aggregate_batch, filter_matches, parse_shutter_speed, safe_float

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "aggregate_golden.json")))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_batch_equals_reference_outputs(name):
    g = GOLD[name]
    scores, cats = aggregate_batch(g["rows"], AggregatePolicy(g["config"]))
    assert cats == g["categories"]
    bad = [i for i, (a, b) in enumerate(zip(scores.tolist(), g["scores"])) if a != b]
    assert not bad, (bad[:5], [(scores[i], g["scores"][i], cats[i]) for i in bad[:5]])


def test_single_row_and_chunking_agree_with_batch():
    g = GOLD["rich"]
    pol = AggregatePolicy(g["config"])
    for i in (0, 3, 27, 101):
        s, c = aggregate(g["rows"][i], pol)
        assert (s, c) == (g["scores"][i], g["categories"][i])
    a, _ = aggregate_batch(g["rows"][:50], pol)
    b, _ = aggregate_batch(g["rows"][50:], pol)
    assert np.concatenate([a, b]).tolist() == g["scores"]
    s, c = aggregate_batch([], pol)
    assert len(s) == 0 and c == []


def test_helpers_follow_the_reference_quirks():
    assert safe_float(1600, None) is None and safe_float(b"x") == 5.0 and safe_float("3.5") == 3.5 and safe_float("no", 1.0) == 1.0
    assert parse_shutter_speed("1/500") == 0.002 and parse_shutter_speed("1/0") is None and parse_shutter_speed(2) == 2.0
    assert filter_matches({}, {}) and not filter_matches({"iso_min": 100}, {"iso": None})
    assert filter_matches({"required_tags": ["A", "b"], "tag_match_mode": "all"}, {"tags": " a , B,c"})
    assert not filter_matches({"excluded_tags": ["c"]}, {"tags": "a,C"})
    with pytest.raises(ValueError):
        AggregatePolicy({})


def test_batchscorer_mapping_feeds_the_aggregate():
    """metrics_for_aggregate() output is a valid input row (the keys batch_processor.py:272-296 passes)."""
    from facet_amd.batch import BatchScorer
    res = {"aesthetic": 6.1, "tech_sharpness": 5.5, "color_score": 4.0, "exposure_score": 7.0, "shadow_clipped": 0, "highlight_clipped": 1,
           "histogram_spread": 55.0, "face_count": 1, "face_quality": 6.0, "eye_sharpness": 5.0, "face_ratio": 0.2, "comp_score": 6.5,
           "_isolation_bonus_raw": 1.4, "is_blink": 0, "is_silhouette": 0, "quality_score": 6.1, "scoring_model": "topiq"}
    m = BatchScorer.metrics_for_aggregate(res, exif={"iso": 64, "f_stop": 2.0})
    s, c = aggregate(m, AggregatePolicy(GOLD["rich"]["config"]))
    assert c == "portrait" and 0.0 < s < 10.0

"""Generates tests/golden/aggregate_golden.json by RUNNING THE REFERENCE'S OWN `Facet.calculate_aggregate_logic`
(processing/scorer.py:769-950, with config.ScoringConfig / config.category_filter.CategoryFilter behind it) in the build
container on a configuration and metric rows made up HERE (seeded) - nothing of the reference's scoring_config.json or source
is stored, only our inputs and the (score, category) pairs its code returned.

    python tests/golden/make_aggregate_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))          # import-time dependency only
from config import ScoringConfig                                 # noqa: E402
from processing.scorer import Facet                              # noqa: E402


def W(**kw):
    return {k + "_percent": v for k, v in kw.items()}


CONFIGS = {
    # every filter kind, modifiers, weights that do not sum to 100, equal priorities (stable order), a quality weight
    "rich": {
        "categories": [
            {"name": "silhouette", "priority": 5, "filters": {"is_silhouette": True},
             "weights": W(aesthetic=50, composition=30, exposure=20)},
            {"name": "group_portrait", "priority": 10, "filters": {"has_face": True, "is_group_portrait": True, "face_count_min": 2},
             "weights": W(aesthetic=30, face_quality=30, eye_sharpness=10, composition=15, exposure=15), "modifiers": {"bonus": 0.25}},
            {"name": "portrait_bw", "priority": 20, "filters": {"has_face": True, "is_monochrome": True, "face_ratio_min": 0.05},
             "weights": W(aesthetic=35, face_quality=25, eye_sharpness=20, contrast=20)},
            {"name": "portrait", "priority": 20, "filters": {"has_face": True, "face_ratio_min": 0.05, "face_ratio_max": 0.9},
             "weights": W(aesthetic=30, quality=10, face_quality=25, face_sharpness=5, eye_sharpness=15, tech_sharpness=5, isolation=10),
             "modifiers": {"bonus": 0.5}},
            {"name": "night", "priority": 30, "filters": {"luminance_max": 0.2, "iso_min": 50, "required_tags": ["Night", "stars"], "tag_match_mode": "any"},
             "weights": W(aesthetic=40, exposure=10, noise=20, color=30), "modifiers": {"noise_tolerance_multiplier": 0.3}},
            {"name": "macro", "priority": 35, "filters": {"required_tags": ["macro", "flower"], "tag_match_mode": "all", "excluded_tags": ["people"]},
             "weights": W(aesthetic=45, tech_sharpness=45, saturation=30, power_point=10)},               # sums to 130 -> renormalised
            {"name": "tele", "priority": 40, "filters": {"focal_length_min": 200, "shutter_speed_max": 0.002, "f_stop_max": 8},
             "weights": W(aesthetic=40, tech_sharpness=30, leading_lines=10, dynamic_range=20),
             "modifiers": {"_clipping_multiplier": 2.0, "_apply_blink_penalty": True, "_skip_oversaturation_penalty": True}},
            {"name": "default", "priority": 100, "filters": {},
             "weights": W(aesthetic=35, tech_sharpness=15, exposure=10, composition=20, color=10, contrast=5, dynamic_range=5)},
        ],
        "scoring": {"score_min": 0.0, "score_max": 10.0},
        "thresholds": {"portrait_face_ratio_percent": 5, "blink_penalty_percent": 40},
        "penalties": {"noise_sigma_threshold": 3.0, "noise_max_penalty_points": 1.2, "noise_penalty_per_sigma": 0.25, "bimodality_threshold": 2.0,
                      "bimodality_penalty_points": 0.4, "oversaturation_threshold": 0.8, "oversaturation_penalty_points": 0.6,
                      "leading_lines_blend_percent": 25},
        "exif_adjustments": {"iso_sharpness_compensation": True, "aperture_isolation_boost": True},
        "exposure": {"silhouette_detection": True},
    },
    # all optional sections absent (the reference's built-in defaults), no fallback category -> viewer default, narrow limits
    "sparse": {
        "categories": [
            {"name": "portrait", "priority": 1, "filters": {"has_face": True}, "weights": W(aesthetic=60, face_quality=40)},
            {"name": "landscape", "filters": {"has_face": False, "required_tags": ["landscape"]}, "weights": W(aesthetic=50, composition=25, dynamic_range=25)},
        ],
        "viewer": {"default_category": "misc"},
        "scoring": {"score_min": 1.0, "score_max": 9.0},
    },
    # switches off
    "switched": {
        "categories": [{"name": "default", "priority": 1, "filters": {}, "weights": W(aesthetic=50, tech_sharpness=25, isolation=25),
                        "modifiers": {"_skip_clipping_penalty": False}}],
        "exif_adjustments": {"iso_sharpness_compensation": False, "aperture_isolation_boost": False},
        "exposure": {"silhouette_detection": False},
        "thresholds": {"blink_penalty_percent": 0},
    },
}

TAGS = [None, "", "night", "Night, city", "macro,flower", "macro, flower, people", "landscape", "portrait,group", "stars , sky", "flower"]


def make_rows(rng, n):
    rows = []
    for i in range(n):
        r = lambda lo, hi: float(rng.uniform(lo, hi))        # noqa: E731
        faces = int(rng.integers(0, 4)) if rng.random() < 0.6 else 0
        m = {
            "aesthetic": r(0, 10), "face_count": faces, "face_quality": r(0, 10) if faces else 0, "eye_sharpness": r(0, 12) if faces else 0,
            "face_sharpness": r(0, 10) if faces else 0, "tech_sharpness": r(0, 11), "color_score": r(0, 10), "exposure_score": r(0, 10),
            "face_ratio": r(0, 0.5) if faces else 0, "comp_score": r(0, 10), "isolation_bonus": r(1.0, 3.5) if faces else 1.0,
            "is_blink": int(rng.random() < 0.3), "shadow_clipped": int(rng.random() < 0.3), "highlight_clipped": int(rng.random() < 0.3),
            "is_silhouette": int(rng.random() < 0.15), "histogram_spread": r(0, 120), "iso": [None, 100, 50, 64, 800, 3200][int(rng.integers(0, 6))],
            "f_stop": [None, 1.4, 2.0, 2.8, 4.0, 11, 0][int(rng.integers(0, 7))], "quality_score": r(0, 10), "scoring_model": "topiq",
            "tags": TAGS[int(rng.integers(0, len(TAGS)))], "is_group_portrait": int(faces >= 2 and rng.random() < 0.7),
            "is_monochrome": int(rng.random() < 0.2), "mean_luminance": r(0, 1), "noise_sigma": r(0, 12), "histogram_bimodality": r(0, 4),
            "mean_saturation": r(0, 1), "leading_lines_score": r(0, 8) if rng.random() < 0.5 else 0, "contrast_score": r(0, 10),
            "power_point_score": r(0, 10), "shutter_speed": [None, "1/500", "1/4000", 0.01, "2", "x/3", "1/0"][int(rng.integers(0, 7))],
            "focal_length": [None, 24, 85, 300][int(rng.integers(0, 4))],
        }
        # the odd inputs the reference's helpers absorb
        if i % 11 == 3:
            m["aesthetic"] = None
        if i % 13 == 5:
            m["tech_sharpness"] = "7.25"
        if i % 17 == 7:
            m["exposure_score"] = "n/a"
        if i % 19 == 2:
            m["comp_score"] = 250.0
        if i % 23 == 4:
            for k in ("noise_sigma", "histogram_bimodality", "mean_saturation", "leading_lines_score", "contrast_score", "power_point_score",
                      "mean_luminance", "is_monochrome", "tags", "focal_length", "shutter_speed"):
                m.pop(k)                          # the mapping batch_processor.py:272-296 passes has none of these
        rows.append(m)
    return rows


def main():
    out = {}
    scorer = object.__new__(Facet)                    # the method needs no instance state beyond .config
    for name, conf in CONFIGS.items():
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump(conf, f)
        try:
            cfg = ScoringConfig(f.name, validate=False)
            rows = make_rows(np.random.default_rng({"rich": 101, "sparse": 202, "switched": 303}[name]), 160)
            scorer.config = cfg                       # `cfg = config or self.config if hasattr(self, 'config') else None` (:774)
            res = [scorer.calculate_aggregate_logic(dict(m), cfg) for m in rows]
        finally:
            os.unlink(f.name)
        out[name] = {"config": conf, "rows": rows, "scores": [float(s) for s, _ in res], "categories": [c for _, c in res]}
        print(name, "categories:", {c: out[name]["categories"].count(c) for c in sorted(set(out[name]["categories"]))})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aggregate_golden.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

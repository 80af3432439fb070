"""Mirror of the reference's `CompositionAnalyzer` (analyzers/composition.py) over the engine.

    CompositionAnalyzer.get_placement_data(bbox, w, h, config)        rule-of-thirds placement (:111-187; facet_amd/batch.py)
    CompositionAnalyzer.detect_leading_lines(img_cv, cache=None)      :191-261 - one image
    CompositionAnalyzer.detect_leading_lines_batch(engine, bgr_batch) the same for a whole batch in one engine call
    CompositionAnalyzer.integrate_leading_lines(base, lines, faces)   :262-283

The reference runs cv2.GaussianBlur + cv2.Canny + cv2.HoughLinesP per image; here `fe_leading_lines` does the pixel scans on
the GPU and the sequential Hough stage on host threads (include/facet_engine.h). Scoring of the segments follows :231-256 with
the same numpy types (int32 coordinates, float64 arithmetic, numpy's round)."""
import numpy as np

from . import default_engine
from .batch import placement_data


def score_lines(lines, h, w):
    """lines: int32 [k,4] as cv2.HoughLinesP returns them (k may be 0 = the reference's `lines is None`)."""
    if lines is None or len(lines) == 0:
        return {'leading_lines_score': 0, 'line_count': 0}
    seg = np.asarray(lines, dtype=np.int32).reshape(-1, 4)
    dx, dy = seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1]                # int32, as the reference's per-segment scalars
    length = np.sqrt(dx ** 2 + dy ** 2)
    with np.errstate(divide='ignore', invalid='ignore'):
        angle = np.where(dx != 0, np.abs(np.degrees(np.arctan(dy / dx))), 90.0)
    bonus = np.where((angle >= 15) & (angle <= 75), 1.5, 1.0)           # diagonals guide the eye (:244-247)
    terms = (length / np.sqrt(h ** 2 + w ** 2)) * 10 * bonus
    total_score = np.float64(sum(terms.tolist()))                       # left-to-right float64 sum, like the reference's loop
    score = min(10.0, total_score / max(1, len(seg)) * 2)
    return {'leading_lines_score': round(score, 2), 'line_count': len(lines)}


class CompositionAnalyzer:
    @staticmethod
    def get_placement_data(bbox, img_w, img_h, config=None):
        wts = config.get_composition_weights() if config is not None else {}
        return placement_data(bbox, img_w, img_h, wts.get('power_point_weight', 2.0), wts.get('line_weight', 1.0))

    @staticmethod
    def detect_leading_lines_batch(engine, bgr_batch):
        bgr = np.ascontiguousarray(bgr_batch, dtype=np.uint8)
        h, w = bgr.shape[1:3]
        return [score_lines(l, h, w) for l in engine.leading_lines(bgr)]

    @staticmethod
    def detect_leading_lines(img_cv, cache=None, engine=None):
        if img_cv is None:
            return {'leading_lines_score': 0, 'line_count': 0}
        engine = engine or default_engine()          # `cache` is accepted for signature parity; the engine computes its own gray
        return CompositionAnalyzer.detect_leading_lines_batch(engine, np.asarray(img_cv)[None])[0]

    @staticmethod
    def integrate_leading_lines(base_comp_score, leading_lines_score, has_faces):
        if has_faces:
            return base_comp_score
        return min(10.0, base_comp_score + min(2.0, leading_lines_score / 5.0))

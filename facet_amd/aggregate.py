"""The aggregate score as a whole-batch epilogue: what reference `Facet.calculate_aggregate_logic`
(processing/scorer.py:769-950) computes per image - category by priority-ordered filter rules
(config/scoring_config.py:792-812, config/category_filter.py:55-149, processing/scorer.py:738-767), category weights
(config/scoring_config.py:301-338), EXIF adjustments, the weighted sum over 16 metrics, blink / clipping / noise /
bimodality / oversaturation penalties (processing/scorer.py:363-403) - evaluated over the columns of a batch.

    policy = AggregatePolicy(config_dict)            # the dict json.load() gives for a v4 scoring_config.json
    scores, categories = aggregate_batch([BatchScorer.metrics_for_aggregate(r) ...], policy)

The policy object only READS the configuration (no validation, normalisation proposals or saving - that is the reference's
config subsystem and stays there). Arithmetic is done column-wise in float64 with the reference's operation order, so the
results equal the per-image Python floats of the reference bit for bit (pinned: tests/test_aggregate.py against outputs of
the reference's own function, tests/golden/aggregate_golden.json).
"""
import numpy as np

# metric name -> position in the weighted sum; the order is the reference's `metrics_map` (processing/scorer.py:876-897)
METRICS = ('aesthetic', 'quality', 'face_quality', 'face_sharpness', 'eye_sharpness', 'tech_sharpness', 'composition', 'power_point',
           'leading_lines', 'exposure', 'color', 'contrast', 'dynamic_range', 'saturation', 'noise', 'isolation')
_NUMERIC_FILTERS = (('face_ratio', 'face_ratio'), ('face_count', 'face_count'), ('iso', 'iso'), ('shutter_speed', 'shutter_speed'),
                    ('luminance', 'mean_luminance'), ('focal_length', 'focal_length'), ('f_stop', 'f_stop'))
_BOOL_FILTERS = ('has_face', 'is_monochrome', 'is_silhouette', 'is_group_portrait')


def safe_float(val, default=5.0):
    """processing/scorer.py:345-360: None / bytes / unparsable / outside [-100, 100] -> default (so an ISO of 1600 reads as
    'no ISO', and a histogram spread above 100 as 0 - the reference's behaviour, kept)."""
    if val is None or isinstance(val, bytes):
        return default
    if isinstance(val, str):
        try:
            val = float(val)
        except ValueError:
            return default
    if isinstance(val, (int, float)):
        return default if (val < -100 or val > 100) else float(val)
    return default


def _category_float(val, default):
    """The narrower helper inside `_determine_photo_category` (processing/scorer.py:738-744): strings are not parsed."""
    if val is None or isinstance(val, bytes) or not isinstance(val, (int, float)):
        return default
    return float(val) if -100 <= val <= 100 else default


def parse_shutter_speed(val):
    """processing/scorer.py:710-725: '1/500' -> 0.002."""
    if val is None:
        return None
    if isinstance(val, (int, float)):
        return float(val)
    if isinstance(val, str):
        try:
            if '/' in val:
                num, den = val.split('/')
                return float(num) / float(den)
            return float(val)
        except (ValueError, ZeroDivisionError):
            return None
    return None


def filter_matches(filters, photo):
    """config/category_filter.py:55-149. An empty filter set matches everything; a numeric bound on a missing value fails."""
    if not filters:
        return True
    for field, key in _NUMERIC_FILTERS:
        actual = photo.get(key)
        lo, hi = filters.get(field + '_min'), filters.get(field + '_max')
        if lo is not None and (actual is None or actual < lo):
            return False
        if hi is not None and (actual is None or actual > hi):
            return False
    for field in _BOOL_FILTERS:
        want = filters.get(field)
        if want is not None:
            have = (photo.get('face_count') or 0) > 0 if field == 'has_face' else bool(photo.get(field, 0))
            if have != want:
                return False
    required, excluded = filters.get('required_tags', []), filters.get('excluded_tags', [])
    if required or excluded:
        have = [t.strip().lower() for t in (photo.get('tags') or '').split(',') if t.strip()]
        if required:
            hits = [t.lower() in have for t in required]
            if not (any(hits) if filters.get('tag_match_mode', 'any') == 'any' else all(hits)):
                return False
        if excluded and any(t.lower() in have for t in excluded):
            return False
    return True


class AggregatePolicy:
    """Read-only view of a v4 scoring configuration with the accessors `calculate_aggregate_logic` uses."""

    def __init__(self, config):
        if 'categories' not in config:
            raise ValueError("not a v4.0 scoring configuration (missing 'categories')")       # scoring_config.py:106-110
        self.config = config
        self.categories = sorted(config.get('categories', []), key=lambda c: c.get('priority', 100))   # :782-790 (stable)
        self.default_category = config.get('viewer', {}).get('default_category', 'default')
        limits = config.get('scoring', {})
        self.score_min, self.score_max = limits.get('score_min', 0.0), limits.get('score_max', 10.0)
        thresholds = config.get('thresholds', {})
        self.portrait_ratio = (thresholds.get('portrait_face_ratio_percent', 0) or 5) / 100
        self.blink_penalty = (thresholds.get('blink_penalty_percent', 0) or 50) / 100
        self.exif = config.get('exif_adjustments', {'iso_sharpness_compensation': True, 'aperture_isolation_boost': True})
        self.exposure = config.get('exposure', {'silhouette_detection': True})
        self.penalties = config.get('penalties', {})
        self._weights = {}

    def weights(self, category):
        """config/scoring_config.py:301-338: `x_percent` -> x / 100, renormalised to sum 1 when off by more than 0.001,
        modifiers merged on top."""
        if category not in self._weights:
            conv = {}
            for cat in self.config.get('categories', []):
                if cat.get('name') == category:
                    keys = []
                    for k, v in cat.get('weights', {}).items():
                        if k.endswith('_percent'):
                            conv[k[:-8]] = v / 100
                            keys.append(k[:-8])
                        else:
                            conv[k] = v
                    if keys:
                        total = sum(conv[k] for k in keys)
                        if total > 0 and abs(total - 1.0) > 0.001:
                            for k in keys:
                                conv[k] = conv[k] / total
                    conv.update(cat.get('modifiers', {}))
                    break
            self._weights[category] = conv
        return self._weights[category]

    def category_of(self, m):
        """processing/scorer.py:727-767 + config/scoring_config.py:792-812."""
        photo = {
            'tags': m.get('tags', '') or '', 'face_count': int(_category_float(m.get('face_count'), 0)),
            'face_ratio': _category_float(m.get('face_ratio'), 0), 'is_silhouette': m.get('is_silhouette', 0),
            'is_group_portrait': m.get('is_group_portrait', 0), 'is_monochrome': m.get('is_monochrome', 0),
            'mean_luminance': _category_float(m.get('mean_luminance'), 0.5), 'iso': m.get('iso'),
            'shutter_speed': parse_shutter_speed(m.get('shutter_speed')), 'focal_length': m.get('focal_length'), 'f_stop': m.get('f_stop'),
        }
        for cat in self.categories:
            if filter_matches(cat.get('filters', {}), photo):
                return cat['name']
        return self.default_category


def _col(rows, key, default):
    return np.array([safe_float(m.get(key), default) for m in rows], dtype=np.float64)


def aggregate_batch(rows, policy):
    """rows: one metrics mapping per image (the mapping batch_processor.py:272-296 builds). Returns (float64 [n] aggregate
    scores, list of category names)."""
    n = len(rows)
    if n == 0:
        return np.zeros(0), []
    cats = [policy.category_of(m) for m in rows]
    wts = [policy.weights(c) for c in cats]
    pen = policy.penalties

    # EXIF-aware adjustments (:797-813). iso / f_stop pass through safe_float with default None -> NaN marks "absent".
    sharp = _col(rows, 'tech_sharpness', 5.0)
    if policy.exif.get('iso_sharpness_compensation', True):
        iso = np.array([safe_float(m.get('iso'), None) or np.nan for m in rows], dtype=np.float64)
        with np.errstate(invalid='ignore', divide='ignore'):
            sharp = np.where(iso > 800, np.minimum(10.0, sharp + 0.5 * np.log2(iso / 800)), sharp)
    isolation = np.array([m.get('isolation_bonus', 1.0) for m in rows], dtype=np.float64)
    if policy.exif.get('aperture_isolation_boost', True):
        fst = np.array([safe_float(m.get('f_stop'), None) or np.nan for m in rows], dtype=np.float64)
        with np.errstate(invalid='ignore'):
            mult = np.where(fst <= 2.0, 1.5, 1.3)
            isolation = np.where(fst <= 2.8, np.minimum(3.0, isolation * mult), isolation)
    isolation_score = np.minimum(10.0, (isolation - 1.0) * 5.0)

    # clipping penalty (:818-831)
    sil_on = policy.exposure.get('silhouette_detection', True)
    silhouette = np.array([bool(m.get('is_silhouette', 0)) if sil_on else False for m in rows])
    shadow = np.array([m.get('shadow_clipped', 0) or 0 for m in rows], dtype=np.float64)
    highlight = np.array([m.get('highlight_clipped', 0) or 0 for m in rows], dtype=np.float64)
    clipping = np.where(~silhouette & ((shadow != 0) | (highlight != 0)), shadow * 0.5 + highlight * 1.0, 0.0)
    dynamic_range = np.minimum(10.0, _col(rows, 'histogram_spread', 0) / 6.0)

    # penalties (:363-403)
    sigma = _col(rows, 'noise_sigma', 0)
    n_thr = pen.get('noise_sigma_threshold', 4.0)
    noise_pen = np.where(sigma > n_thr, np.minimum(pen.get('noise_max_penalty_points', 1.5), (sigma - n_thr) * pen.get('noise_penalty_per_sigma', 0.3)), 0.0)
    bimod_pen = np.where(_col(rows, 'histogram_bimodality', 0) > pen.get('bimodality_threshold', 2.5), pen.get('bimodality_penalty_points', 0.5), 0.0)
    oversat_pen = np.where(_col(rows, 'mean_saturation', 0) > pen.get('oversaturation_threshold', 0.9), pen.get('oversaturation_penalty_points', 0.5), 0.0)
    lines = np.minimum(10.0, _col(rows, 'leading_lines_score', 0) * 1.77)
    blend = pen.get('leading_lines_blend_percent', 30) / 100

    # metric columns (:846-897)
    aes = _col(rows, 'aesthetic', 5.0)
    col = np.where(np.array([bool(m.get('is_monochrome', 0)) for m in rows]), 5.0, _col(rows, 'color_score', 5.0))
    comp_raw = _col(rows, 'comp_score', 5.0)
    non_portrait = np.array([c not in ('portrait', 'group_portrait') for c in cats])
    comp = np.where(non_portrait & (lines > 0), np.minimum(10.0, comp_raw + lines * blend), comp_raw)
    w_aes = np.array([w.get('aesthetic', 0) for w in wts], dtype=np.float64)
    aes_extra = np.array([w.get('quality', 0.0) for w in wts], dtype=np.float64)
    aes_eff = np.where(w_aes > 0, aes + aes_extra / np.maximum(w_aes, 0.01), aes)
    values = {
        'aesthetic': aes_eff, 'quality': np.zeros(n), 'face_quality': _col(rows, 'face_quality', 5.0),
        'face_sharpness': _col(rows, 'face_sharpness', 5.0), 'eye_sharpness': _col(rows, 'eye_sharpness', 5.0), 'tech_sharpness': sharp,
        'composition': comp, 'power_point': _col(rows, 'power_point_score', 5.0), 'leading_lines': lines,
        'exposure': _col(rows, 'exposure_score', 5.0), 'color': col, 'contrast': _col(rows, 'contrast_score', 5.0),
        'dynamic_range': dynamic_range, 'saturation': np.minimum(10.0, _col(rows, 'mean_saturation', 0.5) * 10.0),
        'noise': np.maximum(0.0, np.minimum(10.0, 10.0 - sigma * 0.7)), 'isolation': isolation_score,
    }

    # weighted sum in the reference's order (:920-926); a weight <= 0 contributes nothing
    score = np.zeros(n)
    for name in METRICS:
        wcol = np.array([w.get(name, 0.0) for w in wts], dtype=np.float64)
        score = score + np.where(wcol > 0, np.maximum(0.0, np.minimum(10.0, values[name])) * wcol, 0.0)

    # per-category switches (:899-917) and the penalty chain (:928-946)
    def flag(key, fallback):
        return np.array([bool(w.get(key, fallback(c))) for w, c in zip(wts, cats)])
    blink = flag('_apply_blink_penalty', lambda c: c in ('portrait', 'portrait_bw', 'group_portrait')) & np.array([bool(m.get('is_blink')) for m in rows])
    score = np.where(blink, score * policy.blink_penalty, score)
    score = score + np.array([w.get('bonus', 0.0) for w in wts], dtype=np.float64)
    clip_mult = np.array([w.get('_clipping_multiplier', 1.5 if c == 'default' else 1.0) for w, c in zip(wts, cats)], dtype=np.float64)
    score = np.where(flag('_skip_clipping_penalty', lambda c: c == 'silhouette'), score, score - clipping * clip_mult)
    score = score - noise_pen * np.array([w.get('noise_tolerance_multiplier', 1.0) for w in wts], dtype=np.float64)
    score = score - bimod_pen
    score = np.where(flag('_skip_oversaturation_penalty', lambda c: c in ('night', 'astro', 'concert')), score, score - oversat_pen)
    return np.minimum(policy.score_max, np.maximum(policy.score_min, score)), cats


def aggregate(m, policy):
    """One image: (score, category) as `calculate_aggregate_logic(m)` returns them."""
    s, c = aggregate_batch([m], policy)
    return float(s[0]), c[0]

"""The batch step: what reference processing/batch_processor.py `_process_batch` (:169-360) and the multi-pass `_pass_*`
functions (processing/multi_pass.py:481-644) sequence per image - one model after another, one image after another - as a
handful of engine calls over a whole same-sized batch.

    BatchScorer(engine, tagger=None, face_analyzer=None).process_batch(images_rgb) -> list[dict]

Each dict carries, under the reference's own key names (batch_processor.py:298-354), every value that comes out of a model or
a pixel scan: CLIP aesthetic + embedding blob, TOPIQ quality, SAMP-Net composition score / pattern, the face dict's fields,
the seven technical-metric groups, tags, and the two cross terms the reference derives on the spot (face_ratio :244,
isolation_bonus :264-269). What stays with the caller because it needs files, configuration policy or libraries outside the
hot path: path / EXIF columns, phash (imagehash), leading lines (CompositionAnalyzer.detect_leading_lines: Canny + probabilistic
Hough; `detect_lines=True` computes them through `fe_leading_lines`, or pass `leading_lines=` scores in). With `policy=` (an `aggregate.AggregatePolicy` made from the
scoring configuration) the category and aggregate score (`Facet.calculate_aggregate_logic`) are computed for the whole batch
as the last step, from the multi-pass metrics mapping (multi_pass.py:713-752); `metrics_for_aggregate()` returns the
narrower mapping of the single-pass path (batch_processor.py:272-296).

Engine calls per batch: fe_ensemble_score (TOPIQ + CLIP + aesthetic + U2-Net-P + SAMP-Net), fe_image_stats (technical scans),
fe_face_analyze + fe_roi_laplacian (through FaceAnalyzer.analyze_faces_batch), fe_tag_similarities (through CLIPTagger).

Overlap: a context runs one call at a time (one arena, one stream), and the face / statistics / leading-lines calls spend most of
their time in host glue (NMS, similarity transforms, Hough votes, percentile arithmetic) with the GPU idle. Give the scorer a second
context on the same GPU (`aux_engine=`, and build the FaceAnalyzer on it): those calls then run on a worker thread beside
fe_ensemble_score - ctypes drops the GIL, the hardware queues interleave the two streams - and the step costs max(models, rest)
instead of their sum. Results are identical to the single-context path (tests/test_batch_gpu.py).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .aggregate import aggregate_batch
from .image_stats import TechnicalAnalyzer
from .samp_net import postprocess as samp_postprocess


def tags_to_string(tags):
    """utils/tags.py:8-20."""
    return ','.join(tags) if tags else None


def placement_data(bbox, img_w, img_h, power_weight=2.0, line_weight=1.0):
    """CompositionAnalyzer.get_placement_data(bbox, w, h, config) as `_process_batch` calls it (batch_processor.py:245-247, without
    img_cv): rule-of-thirds power points / lines vs centred composition; analyzers/composition.py:111-187. The two weights are
    config.get_composition_weights()'s power_point_weight / line_weight."""
    if bbox is None:
        return {'score': 7.0, 'power_point_score': 5.0, 'line_score': 5.0, 'center_score': 7.0}
    cx = (bbox[0] + bbox[2]) / 2 / img_w
    cy = (bbox[1] + bbox[3]) / 2 / img_h
    thirds = [1 / 3, 2 / 3]
    nearest_pp = min(np.sqrt((cx - px) ** 2 + (cy - py) ** 2) for px in thirds for py in thirds)
    pp = max(0, 10 - nearest_pp * 25)
    line = max(0, 10 - (min(abs(cx - t) for t in thirds) + min(abs(cy - t) for t in thirds)) * 15)
    centre = max(0, 10 - (abs(cx - 0.5) + abs(cy - 0.5)) * 10)
    best = max((pp * power_weight + line * line_weight) / (power_weight + line_weight), centre)
    return {'score': round(best, 2), 'power_point_score': round(pp, 2), 'line_score': round(line, 2), 'center_score': round(centre, 2)}


def detect_silhouette(histogram_silhouette, tags, face_count):
    """utils/detection.py:8-29: (histogram silhouette or a 'silhouette' tag) and a human (a face, or a portrait / group tag).
    `tags` is the comma-joined tag string (substring tests, as in the reference)."""
    tagged = ('silhouette' in tags) if tags else False
    human = face_count > 0 or (any(t in tags for t in ('portrait', 'group')) if tags else False)
    return 1 if ((histogram_silhouette or tagged) and human) else 0


class BatchScorer:
    def __init__(self, engine, tagger=None, face_analyzer=None, tag_threshold=0.22, max_tags=5, mono_threshold=0.10,
                 shadow_threshold=0.15, highlight_threshold=0.10, power_weight=2.0, line_weight=1.0, policy=None, detect_lines=False,
                 aux_engine=None):
        self.engine, self.tagger, self.face_analyzer, self.policy, self.detect_lines = engine, tagger, face_analyzer, policy, detect_lines
        # second context on the same GPU for statistics / faces / lines (see module docstring); None = everything on `engine`, in sequence
        self.aux_engine = aux_engine
        self._pool = ThreadPoolExecutor(max_workers=1) if aux_engine is not None else None
        if aux_engine is not None and face_analyzer is not None and getattr(face_analyzer, "available", False):
            fe = getattr(getattr(face_analyzer, "face_app", None), "engine", None)
            if fe is not None and fe is engine:
                raise ValueError("with aux_engine the FaceAnalyzer must be built on the aux engine (one call at a time per context)")
        self.power_weight, self.line_weight = power_weight, line_weight
        self.tag_threshold, self.max_tags = tag_threshold, max_tags           # utils/tags.py:50-51 defaults
        self.mono_threshold, self.shadow_threshold, self.highlight_threshold = mono_threshold, shadow_threshold, highlight_threshold

    def process_batch(self, images_rgb, exif=None, leading_lines=None):
        """images_rgb: uint8 [n,h,w,3] (the PIL images of a batch as one array). Returns one dict per image. exif: optional list
        of per-image dicts (iso / f_stop / shutter_speed / focal_length) and leading_lines: optional per-image
        leading_lines_score, both only used by the aggregate step (policy given)."""
        imgs = np.ascontiguousarray(images_rgb, dtype=np.uint8)
        n, h, w, _ = imgs.shape
        # one upload; the BGR copy the reference keeps as img_cv is made on the device, and every engine call reads the resident batch
        e = self.engine
        d_rgb, d_bgr = e.dev_alloc(imgs.nbytes), e.dev_alloc(imgs.nbytes)
        try:
            e.h2d(d_rgb, imgs)
            e.swap_rb(d_rgb, n * h * w, d_bgr)
            rgb_dev, bgr_dev = (d_rgb, n, h, w), (d_bgr, n, h, w)

            def rest(eng):      # everything that is not a model of the ensemble; reads the BGR copy only
                tech_ = TechnicalAnalyzer.analyze_batch(eng, bgr_dev, self.shadow_threshold, self.highlight_threshold, self.mono_threshold)
                faces_ = None
                if self.face_analyzer is not None and self.face_analyzer.available:
                    faces_ = self.face_analyzer.analyze_faces_batch([imgs[i][..., ::-1] for i in range(n)], resident=bgr_dev)
                lines_ = leading_lines
                if lines_ is None and self.detect_lines:      # CompositionAnalyzer.detect_leading_lines (multi_pass.py:702-705), batched
                    from .composition import score_lines
                    lines_ = [score_lines(l, h, w)['leading_lines_score'] for l in eng.leading_lines(bgr_dev)]
                return tech_, faces_, lines_

            if self._pool is not None:
                fut = self._pool.submit(rest, self.aux_engine)      # fe_swap_rb_u8 has synchronised: the BGR copy is complete
                try:
                    rec, mask = e.ensemble_score(rgb_dev)
                finally:
                    tech, faces, leading_lines = fut.result()       # also on error: the worker must be done before the buffers go
            else:
                rec, mask = e.ensemble_score(rgb_dev)
                tech, faces, leading_lines = rest(e)
        finally:
            e.dev_free(d_rgb)
            e.dev_free(d_bgr)
        tags = None
        if self.tagger is not None and self.tagger.text_embeddings is not None and mask & 2:
            tags = self.tagger.get_tags_batch(rec[:, 21:789], self.engine, self.tag_threshold, self.max_tags)
        out = []
        for i in range(n):
            t = tech[i]
            res = {'image_width': w, 'image_height': h}
            if mask & 2:          # CLIP + aesthetic head: Facet.get_aesthetic_and_quality_batch (scorer.py:640-673)
                aesthetic = max(0.0, min(10.0, (float(rec[i, 1]) + 1) * 5))
                res.update({'aesthetic': round(aesthetic, 2), 'clip_embedding': rec[i, 21:789].astype(np.float32).tobytes(),
                            'scoring_model': 'clip-mlp'})
            res['quality_score'] = None
            if mask & 1:          # TOPIQ pass (multi_pass.py:631-642): normalised score (pyiqa_scorer.py:166-195: clamp to [0,1], x10)
                q = max(0.0, min(10.0, max(0.0, min(1.0, float(rec[i, 0]))) * 10.0))      # becomes aesthetic AND quality_score
                if 'aesthetic' in res:
                    res['clip_aesthetic'] = res['aesthetic']
                res.update({'aesthetic': round(q, 2), 'quality_score': q, 'scoring_model': 'topiq'})
            if mask & 4:          # SAMP-Net: get_composition_scores (scorer.py:675-700) overwrites comp_data['score']
                sp = samp_postprocess(rec[i, 2:10], rec[i, 10:16], rec[i, 16:21])
                res.update({'comp_score': round(sp['comp_score'], 2), 'composition_pattern': sp['pattern']})
            res.update({
                'tech_sharpness': round(t['sharpness']['normalized'], 2), 'raw_sharpness_variance': float(t['sharpness']['raw_variance']),
                'color_score': round(t['color']['normalized'], 2), 'raw_color_entropy': float(t['color']['raw_entropy']),
                'exposure_score': round(t['histogram']['exposure_score'], 2), 'histogram_data': t['histogram']['histogram_bytes'],
                'histogram_spread': float(t['histogram']['spread']), 'mean_luminance': float(t['histogram']['mean_luminance']),
                'histogram_bimodality': float(t['histogram']['bimodality']), 'shadow_clipped': t['histogram'].get('shadow_clipped', 0),
                'highlight_clipped': t['histogram'].get('highlight_clipped', 0), 'is_monochrome': t['mono']['is_monochrome'],
                'mean_saturation': t['mono']['mean_saturation'], 'dynamic_range_stops': t['dynamic_range']['dynamic_range_stops'],
                'noise_sigma': t['noise']['noise_sigma'], 'contrast_score': t['contrast']['contrast_score'],
            })
            res['tags'] = tags_to_string(tags[i]) if tags is not None else None
            if faces is not None:
                f = faces[i]
                isolation, blink = 1.0, 0
                if f['face_count'] > 0:      # batch_processor.py:264-269
                    isolation = max(1.0, f['face_sharpness'] / (t['cache'].laplacian_variance + 1))
                    blink = f.get('is_blink', 0)
                res.update({'face_count': f['face_count'], 'face_quality': f['face_quality'], 'eye_sharpness': f['eye_sharpness'],
                            'face_sharpness': f['face_sharpness'], 'face_ratio': f.get('face_area', 0) / (h * w),
                            'raw_eye_sharpness': float(f.get('raw_eye_sharpness', 0)), 'is_group_portrait': f.get('is_group_portrait', 0),
                            'face_confidence': f.get('max_face_confidence', 0), 'isolation_bonus': round(isolation, 2), 'is_blink': blink,
                            'face_details': f.get('face_details', []), '_face_bbox': f.get('bbox'), '_isolation_bonus_raw': isolation})
            # rule-based placement of the (union) face box, then SAMP-Net's score on top when it ran (scorer.py:675-690)
            comp = placement_data(res.get('_face_bbox'), w, h, self.power_weight, self.line_weight)
            res['power_point_score'] = float(comp['power_point_score'])
            res.setdefault('comp_score', round(comp['score'], 2))
            res['is_silhouette'] = detect_silhouette(t['histogram'].get('is_silhouette', 0), res.get('tags'), res.get('face_count', 0))
            if leading_lines is not None:
                res['leading_lines_score'] = float(leading_lines[i])
            out.append(res)
        if self.policy is not None:
            rows = [self.metrics_multi_pass(r, exif[i] if exif else None) for i, r in enumerate(out)]
            scores, cats = aggregate_batch(rows, self.policy)
            for r, s, c in zip(out, scores.tolist(), cats):
                r['aggregate'], r['category'] = s, c
        return out

    def process_images(self, images_rgb, exif=None, leading_lines=None):
        """Images of ANY sizes (a list of uint8 [h,w,3] arrays / PIL images, as a chunk of the reference's loader holds them): grouped by
        shape, each group goes through process_batch as one resident batch; results come back in input order."""
        arrs = [np.asarray(im.convert('RGB') if hasattr(im, 'convert') else im, dtype=np.uint8) for im in images_rgb]
        groups = {}
        for i, a in enumerate(arrs):
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f"image {i}: expected [h,w,3], got {a.shape}")
            groups.setdefault(a.shape, []).append(i)
        out = [None] * len(arrs)
        for idx in groups.values():
            res = self.process_batch(np.stack([arrs[i] for i in idx]), exif=[exif[i] for i in idx] if exif else None,
                                     leading_lines=[leading_lines[i] for i in idx] if leading_lines is not None else None)
            for i, r in zip(idx, res):
                out[i] = r
        return out

    @staticmethod
    def metrics_multi_pass(res, exif=None):
        """The `metrics` mapping of the multi-pass path (multi_pass.py:713-752) from a process_batch dict."""
        exif = exif or {}
        return {
            'aesthetic': res.get('aesthetic', 5.0), 'quality_score': res.get('quality_score'), 'scoring_model': res.get('scoring_model', 'clip-mlp'),
            'comp_score': res.get('comp_score', 5.0), 'face_count': res.get('face_count', 0), 'face_quality': res.get('face_quality', 0),
            'eye_sharpness': res.get('eye_sharpness', 0), 'face_sharpness': res.get('face_sharpness', 0), 'tech_sharpness': res['tech_sharpness'],
            'color_score': res['color_score'], 'exposure_score': res['exposure_score'], 'face_ratio': res.get('face_ratio', 0),
            'tags': res.get('tags') or '', 'isolation_bonus': res.get('_isolation_bonus_raw', 1.0), 'is_blink': res.get('is_blink', 0),
            'is_group_portrait': res.get('is_group_portrait', 0), 'shadow_clipped': res['shadow_clipped'], 'highlight_clipped': res['highlight_clipped'],
            'is_silhouette': res.get('is_silhouette', 0), 'histogram_spread': res['histogram_spread'], 'histogram_bimodality': res['histogram_bimodality'],
            'mean_luminance': res['mean_luminance'], 'is_monochrome': res['is_monochrome'], 'mean_saturation': res['mean_saturation'],
            'contrast_score': res['contrast_score'], 'noise_sigma': res['noise_sigma'], 'leading_lines_score': res.get('leading_lines_score', 0),
            'power_point_score': res.get('power_point_score', 5.0), 'topiq_score': res.get('quality_score'),
            'iso': exif.get('iso'), 'f_stop': exif.get('f_stop'), 'shutter_speed': exif.get('shutter_speed'), 'focal_length': exif.get('focal_length'),
        }

    @staticmethod
    def metrics_for_aggregate(res, exif=None, is_silhouette=None, comp_score=None):
        """The `metrics` mapping `Facet.calculate_aggregate_logic` is called with (batch_processor.py:272-296), built from a
        process_batch dict. comp_score: the caller's rule-based placement score when SAMP-Net is not loaded."""
        exif = exif or {}
        return {
            'aesthetic': res.get('aesthetic'), 'face_count': res.get('face_count', 0), 'face_quality': res.get('face_quality', 0),
            'eye_sharpness': res.get('eye_sharpness', 0), 'tech_sharpness': res['tech_sharpness'], 'color_score': res['color_score'],
            'exposure_score': res['exposure_score'], 'face_ratio': res.get('face_ratio', 0),
            'comp_score': res.get('comp_score', comp_score), 'isolation_bonus': res.get('_isolation_bonus_raw', 1.0),
            'is_blink': res.get('is_blink', 0), 'shadow_clipped': res['shadow_clipped'], 'highlight_clipped': res['highlight_clipped'],
            'is_silhouette': res.get('is_silhouette', 0) if is_silhouette is None else is_silhouette, 'histogram_spread': res['histogram_spread'], 'iso': exif.get('iso'), 'f_stop': exif.get('f_stop'),
            'quality_score': res.get('quality_score'), 'scoring_model': res.get('scoring_model', 'clip-mlp'),
        }

"""The batch step (facet_amd/batch.py): the per-image dict assembled from whole-batch engine calls must equal
what the per-model mirrors give image by image (the reference's own sequencing, processing/batch_processor.py:169-360)."""
import types

import numpy as np
import pytest

from standins import synthetic_onnx as S
from facet_amd._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.batch import BatchScorer
from facet_amd.face import FaceAnalyzer
from facet_amd.image_stats import ImageCache, TechnicalAnalyzer
from facet_amd.samp_net import postprocess
from facet_amd.tagger import CLIPTagger
from facet_amd.weights import synthetic_state_dict, synthetic_images

pytestmark = pytest.mark.gpu


def test_process_batch_equals_per_model_calls():
    from facet_amd import Engine
    e = Engine(0, arena_bytes=12 << 30)
    for mid, name in ((FE_MODEL_TOPIQ, "topiq"), (FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"), (FE_MODEL_SAMP, "samp_net")):
        e.load_weights(mid, synthetic_state_dict(name, 4))
    models = {"det": S.scrfd_like(seed=12, size=320)[0], "lmk": S.landmark_like(seed=13)[0], "rec": S.arcface_iresnet(layers=(1, 1, 1, 1), seed=14)[0]}
    fa = FaceAnalyzer(min_confidence=0.55, min_face_size=10, engine=e, models=models)
    fa.face_app.det_size, fa.face_app.max_candidates, fa.face_app.max_faces = (320, 320), 4096, 64
    vocab = {"portrait": ["a person", "a face"], "landscape": ["a mountain"], "night": ["the night sky", "stars"]}
    tg = CLIPTagger(config=types.SimpleNamespace(get_tag_vocabulary=lambda: vocab, get_art_tags=lambda: set()))
    names = [t for t, d in vocab.items() for _ in d]
    tg.set_text_embeddings(names, np.random.default_rng(1).standard_normal((len(names), 768)).astype(np.float32))
    imgs = synthetic_images(6, 5, 224, 256)
    bs = BatchScorer(e, tagger=tg, face_analyzer=fa, tag_threshold=-1.0, max_tags=2)
    out = bs.process_batch(imgs)
    assert len(out) == 5
    topiq = e.topiq_score(imgs)
    feat, emb, aes = e.clip_encode_images(imgs)
    pw, at, sd = e.samp_score_images(imgs)
    for i, r in enumerate(out):
        q = max(0.0, min(1.0, float(topiq[i]))) * 10.0
        assert r['quality_score'] == pytest.approx(q, rel=1e-5) and r['aesthetic'] == round(r['quality_score'], 2) and r['scoring_model'] == 'topiq'
        assert r['clip_aesthetic'] == pytest.approx(round(max(0.0, min(10.0, (float(aes[i]) + 1) * 5)), 2), abs=0.011)
        assert len(r['clip_embedding']) == 3072 and np.abs(np.frombuffer(r['clip_embedding'], np.float32) - emb[i]).max() < 2e-5
        sp = postprocess(pw[i], at[i], sd[i])
        assert r['composition_pattern'] == sp['pattern'] and r['comp_score'] == pytest.approx(sp['comp_score'], abs=0.011)
        bgr = np.ascontiguousarray(imgs[i][..., ::-1])
        c = ImageCache(bgr, engine=e)
        assert r['raw_sharpness_variance'] == pytest.approx(c.laplacian_variance, rel=1e-12)
        assert r['noise_sigma'] == TechnicalAnalyzer.get_noise_estimate(bgr, c)['noise_sigma']
        assert r['contrast_score'] == TechnicalAnalyzer.get_contrast_score(bgr, c)['contrast_score']
        assert r['histogram_data'] == TechnicalAnalyzer.get_histogram_data(bgr, 0.15, 0.10, c)['histogram_bytes']
        f = fa.analyze_faces(bgr)
        assert r['face_count'] == f['face_count'] and r['face_quality'] == f['face_quality']
        assert r['face_sharpness'] == pytest.approx(f['face_sharpness'], rel=1e-9) and r['is_blink'] == (f['is_blink'] if f['face_count'] else 0)
        if f['face_count']:
            assert r['isolation_bonus'] == round(max(1.0, f['face_sharpness'] / (c.laplacian_variance + 1)), 2)
            assert r['face_ratio'] == pytest.approx(f['face_area'] / (224 * 256), rel=1e-12)
        assert r['tags'] is not None and len(r['tags'].split(',')) == 2
        m = BatchScorer.metrics_for_aggregate(r, exif={'iso': 400, 'f_stop': 2.8})
        assert m['aesthetic'] == r['aesthetic'] and m['iso'] == 400 and m['comp_score'] == r['comp_score'] and m['quality_score'] == r['quality_score']
    # the aggregate step: whole-batch epilogue == the per-image function on the multi-pass mapping
    import json, os
    from facet_amd.aggregate import AggregatePolicy, aggregate
    pol = AggregatePolicy(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "aggregate_golden.json")))["rich"]["config"])
    exif = [{'iso': 64, 'f_stop': 1.8, 'shutter_speed': '1/1000', 'focal_length': 300}] * 5
    out2 = BatchScorer(e, tagger=tg, face_analyzer=fa, tag_threshold=-1.0, max_tags=2, policy=pol).process_batch(imgs, exif=exif, leading_lines=[0, 1, 2, 3, 4])
    for r, r2, x in zip(out, out2, exif):
        assert (r2['aggregate'], r2['category']) == aggregate(BatchScorer.metrics_multi_pass(r2, x), pol)
        assert 0.0 <= r2['aggregate'] <= 10.0 and all(r2[k] == r[k] for k in ('aesthetic', 'quality_score', 'comp_score', 'tags', 'face_count', 'tech_sharpness', 'noise_sigma', 'clip_embedding'))
    # ragged input (real photo chunks mix portrait / landscape sizes): grouped by shape, results in input order
    mixed = [imgs[0], synthetic_images(8, 1, 160, 224)[0], imgs[1], synthetic_images(9, 1, 224, 160)[0], imgs[2]]
    bsr = BatchScorer(e, tagger=tg, face_analyzer=fa, tag_threshold=-1.0, max_tags=2, policy=pol)
    rag = bsr.process_images(mixed, exif=[exif[0]] * 5, leading_lines=[0, 9, 1, 9, 2])
    assert [(r['image_height'], r['image_width']) for r in rag] == [(224, 256), (160, 224), (224, 256), (224, 160), (224, 256)]
    for k, j in ((0, 0), (2, 1), (4, 2)):
        assert rag[k]['quality_score'] == pytest.approx(out2[j]['quality_score'], rel=1e-5) and rag[k]['raw_sharpness_variance'] == out2[j]['raw_sharpness_variance']
        assert rag[k]['category'] == out2[j]['category'] and rag[k]['aggregate'] == pytest.approx(out2[j]['aggregate'], abs=2e-2)
    alone = bsr.process_batch(mixed[1][None], exif=[exif[0]], leading_lines=[9])[0]
    assert rag[1]['quality_score'] == pytest.approx(alone['quality_score'], rel=1e-5) and rag[1]['histogram_data'] == alone['histogram_data']
    # the format invariants the reference's only runtime self-check holds rows to (validation/database_validator.py:14-36, 89-116,
    # 143+, 332-378, 400-425); composition patterns are the 8 names samp_net.py:23-32 emits (the validator's list differs)
    for r in out2:
        for k in ('aesthetic', 'face_quality', 'eye_sharpness', 'tech_sharpness', 'color_score', 'exposure_score', 'comp_score', 'contrast_score',
                  'aggregate', 'quality_score'):
            assert 0.0 <= r[k] <= 10.0, (k, r[k])
        assert len(r['clip_embedding']) == 3072 and len(r['histogram_data']) == 1024
        for k in ('is_blink', 'is_monochrome', 'is_silhouette', 'is_group_portrait', 'shadow_clipped', 'highlight_clipped'):
            assert r[k] in (0, 1), (k, r[k])
        if r['face_count'] == 0:
            assert r['face_quality'] == 0 and r['eye_sharpness'] == 0 and r['face_sharpness'] == 0 and r['face_ratio'] == 0
        assert 0.0 <= r['face_ratio'] <= 1.0
        assert r['composition_pattern'] in ('global', 'horizontal', 'vertical', 'triangular', 'surround', 'quarter', 'cross', 'rule_of_thirds')
    fa.face_app.unload()
    e.close()


def test_aux_engine_overlap_gives_identical_dicts():
    """BatchScorer(aux_engine=...) runs statistics / faces / leading lines on a second context beside fe_ensemble_score: every
    value of every dict must equal the single-context result."""
    from facet_amd import Engine
    e, e2 = Engine(0, arena_bytes=10 << 30), Engine(0, arena_bytes=6 << 30)
    for mid, name in ((FE_MODEL_TOPIQ, "topiq"), (FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"), (FE_MODEL_SAMP, "samp_net")):
        e.load_weights(mid, synthetic_state_dict(name, 4))
    models = {"det": S.scrfd_like(seed=12, size=320)[0], "lmk": S.landmark_like(seed=13)[0], "rec": S.arcface_iresnet(layers=(1, 1, 1, 1), seed=14)[0]}
    fas = []
    for eng in (e, e2):
        fa = FaceAnalyzer(min_confidence=0.55, min_face_size=10, engine=eng, models=models)
        fa.face_app.det_size, fa.face_app.max_candidates, fa.face_app.max_faces = (320, 320), 4096, 64
        fas.append(fa)
    imgs = synthetic_images(6, 7, 224, 256)
    one = BatchScorer(e, face_analyzer=fas[0], detect_lines=True).process_batch(imgs)
    with pytest.raises(ValueError):
        BatchScorer(e, face_analyzer=fas[0], aux_engine=e2)          # the analyzer must live on the aux context
    two = BatchScorer(e, face_analyzer=fas[1], detect_lines=True, aux_engine=e2).process_batch(imgs)
    assert len(one) == len(two) == 7
    for a, b in zip(one, two):
        assert a.keys() == b.keys()
        for k in a:
            if k == 'face_details':
                assert len(a[k]) == len(b[k])
                continue
            va, vb = a[k], b[k]
            if isinstance(va, np.ndarray):
                assert np.array_equal(va, vb), k
            else:
                assert va == vb, (k, va, vb)
    e2.close(); e.close()

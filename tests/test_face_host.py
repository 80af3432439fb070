"""Host-side face logic (no GPU): the product's vectorised helpers against the loop/SVD restatements in oracle/face_ref.py,
and the reference's 'unavailable' contract (analyzers/face.py:39-40, 90-97)."""
import numpy as np

from facet_amd import face as F
from oracle import face_ref


def test_similarity_closed_form_equals_umeyama():
    rng = np.random.default_rng(0)
    for _ in range(50):
        ang, sc = rng.uniform(-3, 3), rng.uniform(0.2, 5)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
        src = (F.ARCFACE_DST.astype(np.float64) @ R.T + rng.uniform(-200, 200, 2) + rng.normal(0, 2.0, (5, 2))).astype(np.float32)
        a = F.similarity_from_5pts(src, F.ARCFACE_DST.astype(np.float64))
        b = face_ref.estimate_norm(src, 112)
        assert np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(b).max())


def test_nms_equals_reference_loop():
    rng = np.random.default_rng(1)
    for n in (1, 7, 200):
        xy = rng.uniform(0, 300, (n, 2))
        wh = rng.uniform(5, 120, (n, 2))
        dets = np.concatenate([xy, xy + wh, rng.uniform(0.5, 1, (n, 1))], axis=1).astype(np.float32)
        dets = dets[dets[:, 4].argsort()[::-1]]
        assert F.nms(dets, 0.4) == [int(k) for k in face_ref.nms(dets, 0.4)]
    assert F.nms(np.zeros((0, 5), np.float32)) == []


def test_gray_laplacian_and_invert():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(F.bgr2gray(img), face_ref.bgr2gray(img))
    g = F.bgr2gray(img)
    want = 0.0
    lap = np.zeros(g.shape)
    for y in range(g.shape[0]):            # explicit reflect-101 loop
        for x in range(g.shape[1]):
            def px(yy, xx):
                yy = -yy if yy < 0 else (2 * (g.shape[0] - 1) - yy if yy >= g.shape[0] else yy)
                xx = -xx if xx < 0 else (2 * (g.shape[1] - 1) - xx if xx >= g.shape[1] else xx)
                return float(g[yy, xx])
            lap[y, x] = px(y - 1, x) + px(y + 1, x) + px(y, x - 1) + px(y, x + 1) - 4 * px(y, x)
    want = lap.var()
    assert abs(F.laplacian_var(g) - want) < 1e-9 * want and abs(face_ref.laplacian_var(g) - want) < 1e-9 * want
    M = rng.normal(0, 1, (4, 2, 3))
    inv = F.invert_affine(M)
    for k in range(4):
        assert np.allclose(inv[k], face_ref.invert_affine(M[k]), rtol=1e-12, atol=1e-12)
        full = np.vstack([M[k], [0, 0, 1]]) @ np.vstack([inv[k], [0, 0, 1]])
        assert np.allclose(full, np.eye(3), atol=1e-9)


def test_unavailable_contract(tmp_path, capsys):
    fa = F.FaceAnalyzer(root=str(tmp_path))            # no buffalo_l files -> same behaviour as a failed insightface import
    assert fa.available is False
    assert "InsightFace not available" in capsys.readouterr().out
    out = fa.analyze_faces(np.zeros((64, 64, 3), np.uint8))
    assert out == {'face_count': 0, 'face_quality': 0, 'eye_sharpness': 0, 'is_blink': 0, 'face_area': 0, 'bbox': None,
                   'face_sharpness': 0, 'raw_eye_sharpness': 0, 'is_group_portrait': 0, 'max_face_confidence': 0, 'face_details': []}
    lm = np.zeros((106, 2), np.float32)
    assert F.FaceAnalyzer.compute_avg_ear(lm) == 0.3


def test_face_dict_attribute_access():
    f = F.Face(bbox=np.arange(4.0), det_score=0.9)
    assert f.det_score == 0.9 and f.embedding is None and f.landmark_2d_106 is None
    f.embedding = np.ones(3)
    assert f["embedding"].sum() == 3

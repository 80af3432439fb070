"""Face path - engine vs the CPU restatement in oracle/face_ref.py.

Parity is UNPINNED against insightface / OpenCV / onnxruntime themselves (absent offline); the oracle restates their published
algorithms and both sides run the same seeded stand-in .onnx graphs (facet_amd/synthetic_onnx.py). Integer pixel work
(cv2.resize, cv2.warpAffine) must be bit-exact; network outputs within 1e-3 relative; boxes after astype(int) identical.
"""
import numpy as np
import pytest

from standins import synthetic_onnx as S
from facet_amd._lib import FE_GRAPH_FACE_REC
from facet_amd.face import ARCFACE_DST, FaceAnalyzer, FaceEngine, similarity_from_5pts
from oracle import face_ref

from conftest import assert_int_boxes_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models():
    det, _ = S.scrfd_like(seed=12, size=320)
    lmk, _ = S.landmark_like(seed=13)
    rec, _ = S.arcface_iresnet(layers=(1, 1, 1, 1), seed=14)
    return {"det": det, "lmk": lmk, "rec": rec}


@pytest.mark.parametrize("shape,out", [((1024, 1024), (640, 640)), ((300, 500), (384, 640)), ((200, 120), (320, 192)), ((640, 480), (320, 240)),
                                       ((97, 131), (211, 57)), ((64, 64), (64, 64))])
def test_cv_resize_linear_bit_exact(engine, shape, out):
    img = np.random.default_rng(shape[0]).integers(0, 256, (2, shape[0], shape[1], 3), dtype=np.uint8)
    got = engine.cv_resize_linear(img, out[0], out[1])
    for i in range(2):
        want = face_ref.cv_resize_linear_u8(img[i], out[0], out[1])
        assert np.array_equal(got[i], want)


def test_warp_affine_bit_exact(engine):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (3, 240, 320, 3), dtype=np.uint8)
    Ms, idx = [], []
    for f in range(10):
        ang, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.3, 2.5)
        c, s = np.cos(ang) * sc, np.sin(ang) * sc
        Ms.append([[c, -s, rng.uniform(-80, 60)], [s, c, rng.uniform(-80, 60)]])     # some crops hang over the border
        idx.append(f % 3)
    Ms = np.asarray(Ms, np.float64)
    for size in (112, 192):
        _, crops = engine.face_crops_run(FE_GRAPH_FACE_REC, img, idx, Ms, size, 0.0, 1.0, out_dim=0, want_crops=True)
        for f in range(10):
            want = face_ref.warp_affine_u8(img[idx[f]], Ms[f], size)
            assert np.array_equal(crops[f], want), (size, f)


def test_detect_matches_oracle(engine, models):
    fe = FaceEngine(engine, models, det_size=(320, 320), max_candidates=4096)
    rng = np.random.default_rng(8)
    for h, w in ((256, 320), (400, 300)):
        imgs = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        got = fe.detect(imgs)
        for i in range(2):
            det_w, kps_w = face_ref.scrfd_detect(models["det"], imgs[i], (320, 320))
            det_g, kps_g = got[i]
            assert det_g.shape == det_w.shape and det_w.shape[0] > 0
            assert np.abs(det_g[:, 4] - det_w[:, 4]).max() < 1e-4
            assert np.abs(det_g[:, :4] - det_w[:, :4]).max() < 2e-2 and np.abs(kps_g - kps_w).max() < 2e-2
            assert_int_boxes_match(det_g[:, :4], det_w[:, :4])
    fe.unload()


def test_face_analysis_end_to_end(engine, models):
    fe = FaceEngine(engine, models, det_size=(320, 320), max_candidates=4096, max_faces=256)
    assert fe.norm["lmk"] == (0.0, 1.0) and fe.norm["rec"] == (127.5, 127.5) and fe.norm["lmk_size"] == 192 and fe.norm["rec_size"] == 112
    imgs = np.random.default_rng(9).integers(0, 256, (2, 288, 352, 3), dtype=np.uint8)
    got = fe.get_batch(imgs)
    ref_models = {"det": (models["det"], 127.5, 128.0), "lmk": (models["lmk"], 0.0, 1.0), "rec": (models["rec"], 127.5, 127.5)}
    for i in range(2):
        want = face_ref.face_analysis_get({"det": ref_models["det"]}, imgs[i], (320, 320))
        assert len(got[i]) == len(want) > 0
        for g, w in zip(got[i], want):
            assert_int_boxes_match(g.bbox, w["bbox"])
        # Crops are cut at 1/32-pixel fixed point, so a 1e-5 px difference in a box (GPU vs CPU detector arithmetic) moves a few
        # percent of the crop's samples by one sub-pixel step; on noise images that alone shifts landmarks by ~1e-3. The crop ->
        # network -> back-projection chain is therefore checked from the SAME boxes / keypoints (the product's).
        for g in got[i][:10]:
            lm = face_ref.landmark_get(models["lmk"], imgs[i], g.bbox, 192, 0.0, 1.0)
            em = face_ref.arcface_get(models["rec"], imgs[i], g.kps, 127.5, 127.5)
            assert np.abs(g.landmark_2d_106 - lm).max() < 1e-4 * max(1.0, np.abs(lm).max())
            assert np.abs(g.embedding - em).max() < 1e-4 * np.abs(em).max()
    # native glue (fe_face_analyze: C++ sort/NMS/crop matrices/back-projection) == numpy glue on the same engine kernels
    host = fe.get_batch_host(imgs)
    for i in range(2):
        assert len(host[i]) == len(got[i])
        for g, hface in zip(got[i], host[i]):
            assert np.array_equal(g.bbox, hface.bbox) and g.det_score == hface.det_score and np.array_equal(g.kps, hface.kps)
            assert np.abs(g.landmark_2d_106 - hface.landmark_2d_106).max() <= 1e-4 * max(1.0, np.abs(hface.landmark_2d_106).max())
            assert np.abs(g.embedding - hface.embedding).max() <= 1e-5 * np.abs(hface.embedding).max()
    fe.max_faces = 3      # capacity smaller than the detections: best-scoring faces are kept, counts still report all
    capped = fe.get_batch(imgs)
    assert all(len(c) == min(3, len(g)) for c, g in zip(capped, got))
    assert all(np.array_equal(c[0].bbox, g[0].bbox) for c, g in zip(capped, got))
    fe.unload()


def test_face_analyzer_dict_matches_reference_logic(engine, models):
    fa = FaceAnalyzer(min_confidence=0.55, min_face_size=10, engine=engine, models=models)
    assert fa.available
    fa.face_app.det_size = (320, 320)
    fa.face_app.max_candidates = 4096
    fa.face_app.max_faces = 256
    imgs = np.random.default_rng(10).integers(0, 256, (2, 320, 320, 3), dtype=np.uint8)
    res = fa.analyze_faces_batch(list(imgs))
    single = fa.analyze_faces(imgs[1])
    for i in range(2):
        faces = fa.face_app.get(imgs[i])
        want = face_ref.analyze_faces(faces, imgs[i], 0.55, 10, fa.blink_ear_threshold, fa.min_faces_for_group)
        got = res[i]
        assert got["face_count"] == want["face_count"] > 0
        for k in ("face_quality", "eye_sharpness", "is_blink", "face_area", "is_group_portrait"):
            assert got[k] == want[k], k
        for k in ("raw_eye_sharpness", "face_sharpness", "max_face_confidence"):
            assert abs(got[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), k
        assert np.array_equal(got["bbox"], want["bbox"])
        assert [d["bbox"] for d in got["face_details"]] == [d["bbox"] for d in want["face_details"]]
        d0 = got["face_details"][0]
        assert len(d0["embedding"]) == 2048 and len(d0["landmark_2d_106"]) == 848          # db/schema.py:105 blob sizes
        assert d0["thumbnail"][:2] == b"\xff\xd8"
    assert single["face_count"] == res[1]["face_count"] and single["face_quality"] == res[1]["face_quality"]
    fa.face_app.unload()
    assert fa.analyze_faces(None)["face_count"] == 0


def test_roi_laplacian_matches_numpy(engine):
    rng = np.random.default_rng(21)
    imgs = rng.integers(0, 256, (3, 90, 140, 3), dtype=np.uint8)
    rois, idx = [], []
    for f in range(40):
        x1, y1 = int(rng.integers(0, 139)), int(rng.integers(0, 89))
        x2, y2 = int(rng.integers(x1 + 1, 141)), int(rng.integers(y1 + 1, 91))
        rois.append([x1, y1, x2, y2]); idx.append(f % 3)
    rois += [[5, 5, 6, 30], [5, 5, 30, 6], [7, 7, 8, 8], [0, 0, 140, 90], [10, 10, 10, 20]]     # 1-wide, 1-high, 1 pixel, full, empty
    idx += [0, 1, 2, 0, 1]
    st = engine.roi_laplacian(imgs, idx, rois)
    for (x1, y1, x2, y2), i, row in zip(rois, idx, st):
        g = face_ref.bgr2gray(imgs[i][y1:y2, x1:x2])
        if g.size == 0:
            assert not row.any()
            continue
        gg = g.astype(np.float64)
        lap = np.zeros_like(gg)                     # explicit per-pixel reflect-101 (a length-1 axis repeats its sample)
        r101 = lambda k, n: 0 if n == 1 else (-k if k < 0 else (2 * (n - 1) - k if k >= n else k))
        for yy in range(gg.shape[0]):
            for xx in range(gg.shape[1]):
                lap[yy, xx] = (gg[r101(yy - 1, gg.shape[0]), xx] + gg[r101(yy + 1, gg.shape[0]), xx] + gg[yy, r101(xx - 1, gg.shape[1])] +
                               gg[yy, r101(xx + 1, gg.shape[1])] - 4.0 * gg[yy, xx])
        assert abs(face_ref.laplacian_var(g) - lap.var()) <= 1e-9 * max(1.0, lap.var())
        assert row[0] == lap.sum() and row[1] == (lap * lap).sum() and row[2] == gg.sum() and row[3] == g.size

"""Known-answer checks of the restated OpenCV semantics (oracle/technical_ref.py, oracle/face_ref.py) - CPU only.

cv2 cannot be imported offline, so these pin the restatements to answers that are common knowledge about cv2's 8-bit
conversions [DEP-KNOWLEDGE]: the primary / secondary colours in HSV (H in [0,180)), Rec.601 gray levels, and structural identities
(identity / integer-shift warps, same-size and constant-image resizes, exact 2x area average). The GPU kernels are then held
bit-exact to these restatements by tests/test_stats_gpu.py and tests/test_face_gpu.py.
"""
import numpy as np

from oracle import face_ref as F
from oracle import technical_ref as T


def _px(b, g, r):
    return np.array([[[b, g, r]]], np.uint8)


def test_gray_known_answers():
    for bgr, want in (((255, 255, 255), 255), ((0, 0, 0), 0), ((0, 0, 255), 76), ((0, 255, 0), 150), ((255, 0, 0), 29), ((128, 128, 128), 128),
                      ((10, 200, 90), round(0.114 * 10 + 0.587 * 200 + 0.299 * 90))):
        assert int(T.bgr2gray(_px(*bgr))[0, 0]) == want, bgr
    assert np.array_equal(T.bgr2gray(_px(1, 2, 3)), F.bgr2gray(_px(1, 2, 3)))


def test_hsv_known_answers():
    cases = {(0, 0, 255): (0, 255, 255), (0, 255, 0): (60, 255, 255), (255, 0, 0): (120, 255, 255), (0, 255, 255): (30, 255, 255),
             (255, 255, 0): (90, 255, 255), (255, 0, 255): (150, 255, 255), (255, 255, 255): (0, 0, 255), (0, 0, 0): (0, 0, 0),
             (77, 77, 77): (0, 0, 77), (128, 64, 32): (110, 191, 128), (32, 64, 128): (10, 191, 128)}
    for bgr, want in cases.items():
        assert tuple(int(v) for v in T.bgr2hsv(_px(*bgr))[0, 0]) == want, bgr
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    hsv = T.bgr2hsv(img)
    assert hsv[..., 0].max() < 180 and np.array_equal(hsv[..., 2], img.max(axis=2))
    # float reference of the textbook formula: the fixed-point tables stay within one unit of it
    b, g, r = (img[..., k].astype(np.float64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r); mn = np.minimum(np.minimum(b, g), r); d = v - mn
    s = np.where(v > 0, 255.0 * d / np.maximum(v, 1), 0)
    assert np.abs(hsv[..., 1].astype(np.float64) - s).max() <= 1.0


def test_laplacian_and_immerkaer_kernels():
    g = np.zeros((5, 5), np.uint8); g[2, 2] = 10
    lap = T.laplacian64(g)
    assert lap[2, 2] == -40 and lap[1, 2] == lap[3, 2] == lap[2, 1] == lap[2, 3] == 10 and lap[0, 0] == 0
    imm = T.filter2d_immerkaer(g.astype(np.float64))
    assert imm[2, 2] == 40 and imm[1, 2] == -20 and imm[1, 1] == 10
    flat = np.full((7, 9), 123, np.uint8)                       # reflect-101 borders: a constant image has zero response everywhere
    assert not T.laplacian64(flat).any() and not T.filter2d_immerkaer(flat.astype(np.float64)).any()
    ramp = np.tile(np.arange(9, dtype=np.uint8) * 10, (7, 1))   # linear ramp: zero inside, reflect-101 doubles the step at the edge
    lap = T.laplacian64(ramp)
    assert not lap[:, 1:-1].any() and (lap[:, 0] == 20).all() and (lap[:, -1] == -20).all()


def test_resize_identities():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    assert np.array_equal(F.cv_resize_linear_u8(img, 40, 60), img)
    const = np.full((33, 47, 3), 201, np.uint8)
    for oh, ow in ((66, 94), (17, 23), (50, 50)):
        assert (F.cv_resize_linear_u8(const, oh, ow) == 201).all()
    half = F.cv_resize_linear_u8(img, 20, 30)
    want = (img.astype(np.int64).reshape(20, 2, 30, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(half, want.astype(np.uint8))
    up = F.cv_resize_linear_u8(img, 80, 120)                   # 2x upscale: every output lies between its two source neighbours
    assert up.min() >= img.min() and up.max() <= img.max()


def test_warp_identities():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (112, 112, 3), dtype=np.uint8)
    eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(F.warp_affine_u8(img, eye, 112), img)
    shift = np.array([[1.0, 0, 7], [0, 1.0, -5]])              # dst(x, y) = src(x - 7, y + 5); uncovered pixels are the border value 0
    out = F.warp_affine_u8(img, shift, 112)
    assert np.array_equal(out[:107, 7:], img[5:, :105]) and not out[:, :7].any() and not out[107:].any()
    M = F.invert_affine(F.invert_affine(np.array([[0.8, -0.3, 12.0], [0.3, 0.8, -4.0]])))
    assert np.allclose(M, [[0.8, -0.3, 12.0], [0.3, 0.8, -4.0]], atol=1e-12)


def test_umeyama_recovers_a_known_similarity():
    rng = np.random.default_rng(3)
    ang, sc, t = 0.4, 1.7, np.array([12.0, -30.0])
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
    src = rng.uniform(0, 100, (5, 2))
    dst = src @ R.T + t
    T3 = F.umeyama(src, dst)
    assert np.allclose(T3[:2, :2], R, atol=1e-10) and np.allclose(T3[:2, 2], t, atol=1e-9)

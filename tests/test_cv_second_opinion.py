"""Second opinions for the restated OpenCV pixel operators (oracle/face_ref.py, technical_ref.py, lines_ref.py): cv2 itself is absent, so
the restatements' GEOMETRY and BORDER conventions are held to independent implementations that are installed - scipy.ndimage for the
integer filters (exact), torch's interpolate / grid_sample and colorsys for the resamplers and colour conversion (to within the
few levels OpenCV's fixed-point arithmetic may differ from float). These do not pin cv2's rounding - the known answers in
tests/test_cv_semantics.py and tests/test_lines_host.py speak to that - but a wrong pixel-centre, border mode, axis or sign shows."""
import colorsys

import numpy as np
import torch
import torch.nn.functional as TF
from scipy import ndimage

from oracle import face_ref as F
from oracle import lines_ref as L
from oracle import technical_ref as T


def _smooth(h, w, seed):
    """Band-limited colour image: interpolation schemes agree on it to within rounding, so geometry errors stand out."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:h, :w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(4):
            fx, fy, ph = rng.uniform(0.02, 0.12), rng.uniform(0.02, 0.12), rng.uniform(0, 6.28)
            img[..., c] += rng.uniform(10, 30) * np.sin(fx * xx + fy * yy + ph)
        img[..., c] += 128
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_integer_filters_equal_scipy():
    rng = np.random.default_rng(0)
    for shape in ((37, 53), (8, 5), (64, 64)):
        g = rng.integers(0, 256, shape, dtype=np.uint8)
        gi = g.astype(np.int64)
        # cv2.Laplacian(ksize=1, CV_64F): 4-neighbour Laplacian, BORDER_REFLECT_101 (scipy 'mirror')
        assert np.array_equal(T.laplacian64(g), ndimage.laplace(gi, mode="mirror").astype(np.float64))
        # cv2.GaussianBlur(5x5, sigma 0) on 8-bit: binomial kernel, reflect-101, (sum + 128) >> 8
        k = np.outer([1, 4, 6, 4, 1], [1, 4, 6, 4, 1])
        assert np.array_equal(L.gaussian5_u8(g), ((ndimage.correlate(gi, k, mode="mirror") + 128) >> 8).astype(np.uint8))
        # Sobel 3x3 under BORDER_REPLICATE (scipy 'nearest'); scipy's sobel = smoothing [1,2,1] x derivative [-1,0,1]
        dx, dy = L.sobel3(g)
        assert np.array_equal(dx, ndimage.sobel(gi, axis=1, mode="nearest")) and np.array_equal(dy, ndimage.sobel(gi, axis=0, mode="nearest"))


def test_gray_and_hsv_close_to_float_formulas():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)             # BGR
    b, g, r = (img[..., i].astype(np.float64) for i in range(3))
    assert np.abs(T.bgr2gray(img).astype(np.float64) - (0.299 * r + 0.587 * g + 0.114 * b)).max() <= 0.51
    hsv = T.bgr2hsv(img).astype(np.float64)
    ref = np.array([colorsys.rgb_to_hsv(*(px[::-1] / 255.0)) for px in img.reshape(-1, 3).astype(np.float64)]).reshape(40, 50, 3)
    assert np.abs(hsv[..., 2] - ref[..., 2] * 255).max() <= 0.51
    assert np.abs(hsv[..., 1] - ref[..., 1] * 255).max() <= 1.01         # 12-bit division tables
    dh = np.abs(hsv[..., 0] - ref[..., 0] * 180)
    dh = np.minimum(dh, 180 - dh)                                        # hue wraps at 180
    sat = ref[..., 1] * ref[..., 2] * 255 > 12                           # hue is ill-conditioned on near-gray pixels
    assert dh[sat].max() <= 1.01


def test_resize_linear_geometry_matches_torch():
    """cv2.resize INTER_LINEAR = half-pixel centres, edge clamp, no antialias = torch bilinear(align_corners=False, antialias=False);
    OpenCV's 11-bit coefficients keep it within one level of the float result."""
    for (h, w, oh, ow, seed) in ((97, 130, 64, 64, 2), (50, 40, 77, 93, 3), (64, 64, 64, 31, 4), (30, 200, 45, 45, 5)):
        img = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = F.cv_resize_linear_u8(img, oh, ow).astype(np.float64)
        ref = TF.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(oh, ow), mode="bilinear", align_corners=False,
                             antialias=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - ref).max() <= 1.0, (h, w, oh, ow, np.abs(got - ref).max())


def test_warp_affine_geometry_matches_grid_sample():
    """cv2.warpAffine: integer pixel coordinates (no half-pixel shift), dst -> src through the inverted matrix, zeros outside. torch's
    grid_sample(align_corners=True) evaluates the same mapping in float; 1/32-pixel coordinate quantisation stays within two levels on
    a smooth image."""
    img = _smooth(120, 150, 7)
    h, w = img.shape[:2]
    for (s, ang, tx, ty) in ((0.6, 0.2, 10.0, -5.0), (1.3, -0.5, -40.0, 20.0), (0.9, 0.0, 3.25, 7.5)):
        M = np.array([[s * np.cos(ang), -s * np.sin(ang), tx], [s * np.sin(ang), s * np.cos(ang), ty]])
        size = 96
        got = F.warp_affine_u8(img, M, size).astype(np.float64)
        Mi = F.invert_affine(M)
        ys, xs = np.mgrid[:size, :size].astype(np.float64)
        sx, sy = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2], Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
        grid = torch.from_numpy(np.stack([2 * sx / (w - 1) - 1, 2 * sy / (h - 1) - 1], -1))[None]
        ref = TF.grid_sample(torch.from_numpy(img).permute(2, 0, 1)[None].double(), grid, mode="bilinear", padding_mode="zeros",
                             align_corners=True)[0].permute(1, 2, 0).numpy()
        inside = (sx >= 1) & (sx <= w - 2) & (sy >= 1) & (sy <= h - 2)           # borders blend with zeros differently by a sub-pixel
        assert inside.sum() > 1000 and np.abs(got - ref)[inside].max() <= 2.0, np.abs(got - ref)[inside].max()
        outside = (sx < -1) | (sx > w) | (sy < -1) | (sy > h)
        assert not got[outside].any()

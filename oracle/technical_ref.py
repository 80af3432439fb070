"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's per-image technical metrics.

Follows reference analyzers/image_cache.py:28-33 (ImageCache: gray, hsv, laplacian_variance) and analyzers/technical.py
(get_sharpness_data :39-58, get_color_harmony_data :80-116, get_histogram_data :129-216, detect_monochrome :219-243,
get_dynamic_range :245-274, get_noise_estimate :276-306, get_contrast_score :308-342) line by line, with the OpenCV calls they
make restated in numpy [DEP-KNOWLEDGE: cv2 is not importable offline]: COLOR_BGR2GRAY 8-bit = (3735 B + 19235 G + 9798 R + 2^14)
>> 15; COLOR_BGR2HSV 8-bit = the sdiv / hdiv180 table arithmetic at 12 fractional bits; Laplacian(CV_64F, ksize 1) and filter2D
with BORDER_REFLECT_101; calcHist as exact counts in float32. Parity status: UNPINNED (the reference has no tests or fixtures for
these functions and cv2 cannot be run here). Deliberately straightforward numpy on whole arrays.
"""
import struct

import numpy as np


def bgr2gray(img):
    a = img.astype(np.int64)
    return ((a[..., 0] * 3735 + a[..., 1] * 19235 + a[..., 2] * 9798 + (1 << 14)) >> 15).astype(np.uint8)


_SDIV = np.array([0] + [int(np.rint((255 << 12) / (1.0 * i))) for i in range(1, 256)], np.int64)
_HDIV = np.array([0] + [int(np.rint((180 << 12) / (6.0 * i))) for i in range(1, 256)], np.int64)


def bgr2hsv(img):
    a = img.astype(np.int64)
    b, g, r = a[..., 0], a[..., 1], a[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * _HDIV[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


def _pad101(g):
    p = np.pad(g, ((1, 1), (0, 0)), mode="reflect" if g.shape[0] > 1 else "edge")      # BORDER_REFLECT_101 per axis; a length-1
    return np.pad(p, ((0, 0), (1, 1)), mode="reflect" if g.shape[1] > 1 else "edge")   # axis repeats its only sample


def laplacian64(gray):
    p = _pad101(gray.astype(np.float64))
    return p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4.0 * p[1:-1, 1:-1]


def filter2d_immerkaer(gray64):
    p = _pad101(gray64)
    M = np.array([[1, -2, 1], [-2, 4, -2], [1, -2, 1]], np.float64)
    out = np.zeros_like(gray64)
    for dy in range(3):
        for dx in range(3):
            out += M[dy, dx] * p[dy:dy + gray64.shape[0], dx:dx + gray64.shape[1]]
    return out


class ImageCache:
    def __init__(self, img_cv):
        self.height, self.width = img_cv.shape[:2]
        self.gray = bgr2gray(img_cv)
        self.hsv = bgr2hsv(img_cv)
        self.laplacian_variance = laplacian64(self.gray).var()


def sharpness_data(cache):
    v = cache.laplacian_variance
    return {'raw_variance': v, 'normalized': float(min(10.0, v / 50.0))}


def color_harmony_data(cache):
    hsv = cache.hsv
    hist = np.zeros((180, 256), np.float32)
    np.add.at(hist, (hsv[..., 0].ravel().astype(np.int64), hsv[..., 1].ravel().astype(np.int64)), 1)
    hist_sum = hist.sum()
    if hist_sum > 0:
        p = hist / hist_sum
        m = p > 0
        ent = -np.sum(p[m] * np.log2(p[m]))
    else:
        ent = 0
    return {'raw_entropy': ent, 'normalized': float(min(10.0, ent * 10.0 / 15.5))}


def histogram_data(cache, shadow_threshold=0.15, highlight_threshold=0.10):
    hist = np.bincount(cache.gray.ravel(), minlength=256).astype(np.float32)
    total = hist.sum()
    hn = hist / total if total > 0 else hist
    bins = np.arange(256)
    mean_val = np.sum(bins * hn)
    spread = np.sqrt(np.sum(((bins - mean_val) ** 2) * hn))
    mean_lum = mean_val / 255.0
    shadow_mass = np.sum(hn[:30])
    highlight_mass = np.sum(hn[225:])
    lower, upper = np.sum(hn[:85]), np.sum(hn[170:])
    sil = 1 if (lower > 0.35 and upper > 0.25) else 0
    from scipy.stats import kurtosis
    bimod = -kurtosis(hn * 256, fisher=True)
    lum_pen = abs(mean_lum - 0.5) * 8
    spread_bonus = min(4.0, spread / 20.0)
    bimod_pen = max(0, bimod - 1.0) * 0.6
    clip_pen = 0 if sil else shadow_mass * 4.0 + highlight_mass * 5.0
    score = max(0, min(10.0, 7.0 - lum_pen + spread_bonus - bimod_pen - clip_pen))
    return {'histogram_bytes': struct.pack('256f', *hn), 'spread': round(spread, 4), 'mean_luminance': round(mean_lum, 4),
            'bimodality': round(bimod, 4), 'exposure_score': round(score, 2), 'shadow_clipped': 1 if shadow_mass > shadow_threshold else 0,
            'highlight_clipped': 1 if highlight_mass > highlight_threshold else 0, 'is_silhouette': sil}


def monochrome(cache, threshold=0.1):
    ms = np.mean(cache.hsv[:, :, 1]) / 255.0
    return {'is_monochrome': 1 if ms < threshold else 0, 'mean_saturation': round(ms, 4)}


def dynamic_range(cache):
    p2, p98 = np.percentile(cache.gray, 2), np.percentile(cache.gray, 98)
    if p2 < 1:
        p2 = 1
    return {'dynamic_range_stops': round(np.log2(max(p98, 1) / p2), 2)}


def noise_estimate(cache):
    g = cache.gray.astype(np.float64)
    h, w = g.shape
    sigma = np.sum(np.abs(filter2d_immerkaer(g)))
    sigma = sigma * np.sqrt(0.5 * np.pi) / (6 * (w - 2) * (h - 2))
    return {'noise_sigma': round(sigma, 2)}


def contrast_score(cache):
    g = cache.gray.astype(np.float64)
    p5, p95 = np.percentile(g, [5, 95])
    pc = (p95 - p5) / 255.0
    rms = np.std(g) / 255.0
    return {'contrast_score': round(min(10.0, (pc * 5.0) + (rms * 20.0)), 2), 'percentile_contrast': round(pc, 4), 'rms_contrast': round(rms, 4)}


def all_metrics(img_cv, shadow_threshold=0.15, highlight_threshold=0.10, mono_threshold=0.1):
    c = ImageCache(img_cv)
    return {'sharpness': sharpness_data(c), 'color': color_harmony_data(c), 'histogram': histogram_data(c, shadow_threshold, highlight_threshold),
            'mono': monochrome(c, mono_threshold), 'dynamic_range': dynamic_range(c), 'noise': noise_estimate(c), 'contrast': contrast_score(c)}

"""Per-image technical statistics: fe_image_stats + facet_amd/image_stats.py vs oracle/technical_ref.py.

Parity UNPINNED against cv2 (absent offline): the oracle restates cv2's fixed-point gray/HSV conversions and border rules and the
reference formulas of analyzers/technical.py. Integer work (planes, histogram, Laplacian / Immerkaer sums, saturation sum) must
be bit-exact; float64 metrics within 1e-9; the entropy within 1e-5 (the reference accumulates it in float32).
"""
import numpy as np
import pytest

from facet_amd.image_stats import ImageCache, TechnicalAnalyzer
from oracle import technical_ref as R

pytestmark = pytest.mark.gpu


def _images(kind, n, h, w, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if kind == "flat":          # monochrome, large flat areas: all pixels of a wavefront land in one histogram bin
        a = np.zeros((n, h, w, 3), np.uint8)
        a[:, : h // 2] = 17
        a[:, h // 2:, : w // 3] = 250
        a[:, h // 2:, w // 3:] = rng.integers(100, 104, (n, h - h // 2, w - w // 3, 1), dtype=np.uint8)
        return a
    yy, xx = np.mgrid[0:h, 0:w]     # smooth colour gradients + mild noise, like a photograph
    base = np.stack([(xx * 255 // max(1, w - 1)), (yy * 255 // max(1, h - 1)), ((xx + yy) * 127 // max(1, h + w - 2))], axis=-1)
    return np.clip(base[None] + rng.integers(-12, 13, (n, h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind,shape", [("noise", (96, 128)), ("flat", (120, 200)), ("photo", (257, 131)), ("photo", (64, 64)), ("noise", (3, 5))])
def test_planes_and_integer_sums_bit_exact(engine, kind, shape):
    imgs = _images(kind, 3, shape[0], shape[1], 5)
    st, gray, hsv = engine.image_stats(imgs, want_gray=True, want_hsv=True)
    for i in range(3):
        c = R.ImageCache(imgs[i])
        assert np.array_equal(gray[i], c.gray) and np.array_equal(hsv[i], c.hsv)
        assert np.array_equal(st[i, :256], np.bincount(c.gray.ravel(), minlength=256).astype(np.float64))
        lap = R.laplacian64(c.gray)
        assert st[i, 256] == lap.sum() and st[i, 257] == (lap * lap).sum()
        assert st[i, 258] == np.abs(R.filter2d_immerkaer(c.gray.astype(np.float64))).sum()
        assert st[i, 259] == c.hsv[..., 1].astype(np.int64).sum()
        cnt = np.bincount(c.hsv[..., 0].ravel().astype(np.int64) * 256 + c.hsv[..., 1].ravel(), minlength=180 * 256).astype(np.float64)
        want = float((cnt[cnt > 0] * np.log2(cnt[cnt > 0])).sum())
        assert abs(st[i, 260] - want) <= 1e-12 * max(1.0, want)


@pytest.mark.parametrize("kind", ["noise", "flat", "photo"])
def test_metric_dicts_match_reference_formulas(engine, kind):
    imgs = _images(kind, 4, 192, 256, 9)
    got = TechnicalAnalyzer.analyze_batch(engine, imgs, 0.15, 0.10, 0.1)
    for i in range(4):
        want = R.all_metrics(imgs[i], 0.15, 0.10, 0.1)
        g = got[i]
        assert abs(g['sharpness']['raw_variance'] - want['sharpness']['raw_variance']) <= 1e-9 * max(1.0, want['sharpness']['raw_variance'])
        assert abs(g['sharpness']['normalized'] - want['sharpness']['normalized']) <= 1e-9
        assert abs(g['color']['raw_entropy'] - float(want['color']['raw_entropy'])) <= 1e-5 * max(1.0, float(want['color']['raw_entropy']))
        assert abs(g['color']['normalized'] - want['color']['normalized']) <= 1e-5
        assert g['histogram'] == want['histogram']          # same numpy arithmetic on an identical histogram: identical bytes and rounding
        assert g['mono'] == want['mono'] and g['dynamic_range'] == want['dynamic_range']
        assert g['noise'] == want['noise'] and g['contrast'] == want['contrast']


def test_percentiles_from_histogram_equal_numpy(engine):
    imgs = _images("photo", 2, 101, 77, 3)
    caches = ImageCache.from_batch(engine, imgs, keep_planes=True)
    for c in caches:
        for q in (0, 2, 5, 33.3, 50, 95, 98, 100):
            assert c.percentile(q) == np.percentile(c.gray, q)


def test_single_image_cache_and_device_resident_batch(engine):
    imgs = _images("noise", 5, 64, 96, 1)
    c = ImageCache(imgs[2], engine=engine, keep_planes=True)
    assert (c.height, c.width) == (64, 96) and np.array_equal(c.gray, R.bgr2gray(imgs[2]))
    d = engine.dev_alloc(imgs.nbytes)
    engine.h2d(d, imgs)
    st_dev, _, _ = engine.image_stats((d, 5, 64, 96))
    st_host, _, _ = engine.image_stats(imgs)
    engine.dev_free(d)
    assert np.array_equal(st_dev[:, :260], st_host[:, :260]) and np.allclose(st_dev[:, 260], st_host[:, 260], rtol=1e-13)
    assert np.array_equal(c.stats[:260], st_host[2, :260])
    # cache=None: the reference converts the image on the spot; the mirror scans it on the process-wide default engine
    assert TechnicalAnalyzer.get_noise_estimate(imgs[0], cache=None) == R.noise_estimate(R.ImageCache(imgs[0]))
    with pytest.raises(TypeError):
        TechnicalAnalyzer.get_noise_estimate(imgs[0], cache="not a cache")
    assert TechnicalAnalyzer.get_noise_estimate(None) == {'noise_sigma': 0}


def test_cacheless_conveniences(engine):
    img = _images("photo", 1, 80, 120, 4)[0]
    c = R.ImageCache(img)
    assert abs(TechnicalAnalyzer.get_sharpness(img) - min(10.0, c.laplacian_variance / 50.0)) < 1e-9
    assert abs(TechnicalAnalyzer.get_color_harmony(img) - R.color_harmony_data(c)['normalized']) < 1e-5
    want = max(0, 10 - (np.sum(c.gray <= 5) / c.gray.size + np.sum(c.gray >= 250) / c.gray.size) * 10)      # technical.py:118-127
    assert abs(TechnicalAnalyzer.get_exposure_score(img) - want) < 1e-12
    assert TechnicalAnalyzer.get_sharpness(None) == 0

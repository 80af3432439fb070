"""fe_leading_lines (reference analyzers/composition.py:191-261) against oracle/lines_ref.py: the Canny edge image
and the Hough segments must be identical (integer work), hence the score too. OpenCV itself is absent -> parity unpinned."""
import numpy as np
import pytest

from oracle import lines_ref as R
from facet_amd.composition import CompositionAnalyzer, score_lines

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from facet_amd import Engine
    e = Engine(0, arena_bytes=2 << 30)
    yield e
    e.close()


def scene(h, w, seed, noise):
    """Bars, a diagonal band, a disc and graded noise: edges of every orientation, ties and weak/strong chains."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w, 3), 50, np.int32)
    img[h // 5:h // 5 + 5, w // 10:w - w // 10] = (200, 180, 90)
    img[h // 8:h - h // 8, w // 3:w // 3 + 4] = (30, 220, 220)
    for t in range(min(h, w) - 30):
        img[12 + t, 8 + t:14 + t] = (240, 240, 240)
    yy, xx = np.mgrid[:h, :w]
    img[(yy - h * 0.7) ** 2 + (xx - w * 0.7) ** 2 < (min(h, w) * 0.15) ** 2] = (120, 40, 200)
    img += (xx[..., None] * 40 // w)
    img += rng.integers(-noise, noise + 1, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,seed,noise", [(200, 260, 1, 0), (130, 97, 2, 6), (256, 256, 3, 25), (64, 300, 4, 3)])
def test_edges_and_segments_equal_oracle(engine, h, w, seed, noise):
    img = scene(h, w, seed, noise)
    lines, edges = engine.leading_lines(img[None], want_edges=True)
    ref_res, ref_lines, ref_edges = R.detect_leading_lines(img)
    assert np.array_equal(edges[0], ref_edges)
    assert edges[0].any() and np.array_equal(lines[0], ref_lines)
    got = score_lines(lines[0], h, w)
    assert got == ref_res and got["line_count"] >= 1


def test_batch_of_images_and_thresholds(engine):
    imgs = np.stack([scene(120, 160, s, n) for s, n in ((5, 0), (6, 10), (7, 40), (8, 2), (9, 0))])
    res = CompositionAnalyzer.detect_leading_lines_batch(engine, imgs)
    for i in range(len(imgs)):
        assert res[i] == R.detect_leading_lines(imgs[i])[0]
        assert res[i] == CompositionAnalyzer.detect_leading_lines(imgs[i], engine=engine)
    # other thresholds / a tiny line budget (the binding re-runs with more room)
    l1, e1 = engine.leading_lines(imgs[2:3], canny_low=20, canny_high=60, threshold=30, min_line_length=10, max_line_gap=3, max_lines=1, want_edges=True)
    ref_e = R.canny_u8(R.gaussian5_u8(R.bgr2gray(imgs[2])), 20, 60)
    assert np.array_equal(e1[0], ref_e) and np.array_equal(l1[0], R.hough_lines_p(ref_e, 30, 10, 3)) and len(l1[0]) > 1


def test_degenerate_inputs(engine):
    flat = np.full((2, 40, 50, 3), 99, np.uint8)
    lines, edges = engine.leading_lines(flat, want_edges=True)
    assert not edges.any() and all(len(l) == 0 for l in lines)
    assert CompositionAnalyzer.detect_leading_lines_batch(engine, flat) == [{'leading_lines_score': 0, 'line_count': 0}] * 2
    tiny = np.random.default_rng(0).integers(0, 256, (1, 1, 7, 3), dtype=np.uint8)        # one row: borders on both sides of every pixel
    l, e = engine.leading_lines(tiny, want_edges=True)
    assert np.array_equal(e[0], R.canny_u8(R.gaussian5_u8(R.bgr2gray(tiny[0])), 50, 150))
    assert CompositionAnalyzer.detect_leading_lines(None) == {'leading_lines_score': 0, 'line_count': 0}


def test_fullsize_resident_batch(engine):
    """1024x1024 (BASELINE's size): device-resident input equals host input, and drawn lines are found."""
    imgs = np.stack([scene(1024, 1024, 11, 4), scene(1024, 1024, 12, 0)])
    host = engine.leading_lines(imgs)
    d = engine.dev_alloc(imgs.nbytes)
    engine.h2d(d, imgs)
    dev = engine.leading_lines((d, 2, 1024, 1024))
    engine.dev_free(d)
    for a, b in zip(host, dev):
        assert np.array_equal(a, b)
    for l in host:
        length = np.hypot(l[:, 2] - l[:, 0], l[:, 3] - l[:, 1])
        assert len(l) >= 3 and length.max() >= 0.5 * 1024

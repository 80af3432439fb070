"""Known answers for the leading-lines oracle (oracle/lines_ref.py): OpenCV's 8-bit Gaussian, Canny tie rules, the
multiply-with-carry generator's arithmetic and the probabilistic Hough walk on cases small enough to work out by hand."""
import numpy as np

from oracle import lines_ref as R


def test_gaussian5_fixed_point():
    assert (R.gaussian5_u8(np.full((7, 9), 200, np.uint8)) == 200).all()          # weights sum to 256
    g = np.zeros((9, 9), np.uint8)
    g[4, 4] = 255
    b = R.gaussian5_u8(g)
    assert b[4, 4] == (36 * 255 + 128) >> 8 and b[4, 3] == (24 * 255 + 128) >> 8 and b[3, 3] == (16 * 255 + 128) >> 8
    assert b[2, 2] == (1 * 255 + 128) >> 8 and b[4, 1] == 0 and b.sum() == b.T.sum()
    g = np.zeros((5, 5), np.uint8)
    g[0, 0] = 255                                                                 # reflect-101: the corner's neighbours count twice
    assert R.gaussian5_u8(g)[0, 0] == (36 * 255 + 128) >> 8 and R.gaussian5_u8(g)[1, 1] == (16 * 255 + 128) >> 8
    assert R.gaussian5_u8(np.array([[7]], np.uint8))[0, 0] == 7                   # a 1x1 image repeats its sample


def test_canny_step_edge_tie_rule():
    g = np.zeros((12, 16), np.uint8)
    g[:, 8:] = 255
    e = R.canny_u8(g, 50, 150)
    # columns 7 and 8 have the same gradient magnitude (4 * 255); `m > left and m >= right` keeps the left one only
    assert (e[:, 7] == 255).all() and e.sum() == 255 * 12
    e = R.canny_u8(g.T.copy(), 50, 150)
    assert (e[7, :] == 255).all() and e.sum() == 255 * 12
    assert R.canny_u8(np.full((8, 8), 90, np.uint8)).sum() == 0


def test_canny_hysteresis_keeps_weak_only_when_connected():
    g = np.zeros((20, 40), np.uint8)
    g[:, 10:] = 30          # weak step: |dx| = 120 -> between the thresholds
    g[:8, 10:] = 60         # strong on the top rows: |dx| = 240 > 150
    e = R.canny_u8(g, 50, 150)
    assert (e[:6, 9] == 255).all() and (e[11:, 9] == 255).all()   # the weak part (rows 11+) hangs on the strong part through the corner
    g2 = np.zeros((20, 40), np.uint8)
    g2[:, 10:] = 30
    assert R.canny_u8(g2, 50, 150).sum() == 0           # weak alone: dropped


def test_rng_is_multiply_with_carry():
    r = R._Rng()
    s = (1 << 64) - 1
    for _ in range(5):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & ((1 << 64) - 1)
        assert r.uniform(0, 1000) == (s & 0xFFFFFFFF) % 1000
    assert r.uniform(3, 3) == 3


def test_hough_single_segments():
    e = np.zeros((128, 160), np.uint8)
    e[50, 20:121] = 255                                 # 101 points on a row
    lines = R.hough_lines_p(e, threshold=80, min_len=19, max_gap=20)
    assert lines.tolist() == [[20, 50, 120, 50]]        # theta = 90 deg: walks to -x first
    e = np.zeros((160, 128), np.uint8)
    e[30:131, 40] = 255
    assert R.hough_lines_p(e, 80, 19, 20).tolist() == [[40, 130, 40, 30]]
    e = np.zeros((128, 160), np.uint8)
    e[50, 20:90] = 255                                  # 70 points: never reaches 80 votes
    assert len(R.hough_lines_p(e, 80, 19, 20)) == 0
    e = np.zeros((128, 160), np.uint8)
    e[50, 20:121] = 255
    e[50, 60:75] = 0                                    # a 15-pixel hole is bridged (gap <= 20) ...
    assert R.hough_lines_p(e, 80, 19, 20).tolist() == [[20, 50, 120, 50]]
    e[50, 60:85] = 0                                    # ... a 25-pixel one is not, and neither part has 80 points
    assert len(R.hough_lines_p(e, 80, 19, 20)) == 0


def test_detect_on_drawn_lines():
    img = np.full((200, 260, 3), 40, np.uint8)
    for t in range(180):                                # a bright diagonal band and a horizontal bar
        img[10 + t, 20 + t:26 + t] = 230
    img[150:156, 30:230] = 200
    res, lines, edges = R.detect_leading_lines(img)
    assert res["line_count"] == len(lines) >= 2 and 0 < res["leading_lines_score"] <= 10 and edges.max() == 255
    ang = np.degrees(np.arctan2(np.abs(lines[:, 3] - lines[:, 1]), np.abs(lines[:, 2] - lines[:, 0])))
    assert ((ang > 40) & (ang < 50)).any() and (ang < 3).any()


# ---- the product's host stage (facet_amd/csrc/lines_host.cpp) under AddressSanitizer + UBSan, against the oracle ----------------
import os
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("lines") / "lines_harness")
    src = os.path.join(ROOT, "facet_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-pthread",
                    "-I", src, os.path.join(ROOT, "tests", "native", "lines_harness.cpp"), os.path.join(src, "lines_host.cpp"), "-o", exe], check=True)
    return exe


def run_harness(exe, maps, thr, min_len, max_gap, max_lines, threads, tmp):
    n, h, w = maps.shape
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("8i", h, w, thr, min_len, max_gap, max_lines, threads, n))
        f.write(np.ascontiguousarray(maps, np.uint8).tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    raw = open(fout, "rb").read()
    edges = np.frombuffer(raw[:n * h * w], np.uint8).reshape(n, h, w)
    counts = np.frombuffer(raw[n * h * w:n * h * w + 4 * n], np.int32)
    lines = np.frombuffer(raw[n * h * w + 4 * n:], np.int32).reshape(n, max_lines, 4)
    return edges, counts, lines


def nms_like_map(rng, h, w, kind):
    """A 2 / 0 / 1 map as the NMS kernel writes it: noise of given densities plus straight and slanted chains."""
    m = np.ones((h, w), np.uint8)
    r = rng.random((h, w))
    dens = (0.02, 0.10, 0.35)[kind % 3]
    m[r < dens] = 0
    m[r < dens * 0.3] = 2
    for _ in range(4):
        if min(h, w) < 8:
            break
        y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
        dy, dx = rng.uniform(-1, 1), rng.uniform(-1, 1)
        for t in range(int(rng.integers(5, max(h, w)))):
            y, x = int(round(y0 + dy * t)), int(round(x0 + dx * t))
            if 0 <= y < h and 0 <= x < w:
                m[y, x] = 2 if (t % 7 == 0) else 0
    return m


def test_host_stage_sanitized_equals_oracle(harness, tmp_path):
    from scipy import ndimage
    rng = np.random.default_rng(42)
    shapes = [(1, 1), (1, 40), (37, 1), (2, 2), (3, 90), (64, 64), (50, 131), (120, 77)]
    for k, (h, w) in enumerate(shapes):
        maps = np.stack([nms_like_map(rng, h, w, k + j) for j in range(3)])
        thr, min_len, max_gap = [(10, 5, 2), (25, 12, 20), (5, 0, 0)][k % 3]
        edges, counts, lines = run_harness(harness, maps, thr, min_len, max_gap, 64, threads=1 + k % 3, tmp=str(tmp_path))
        for i in range(3):
            cand = maps[i] != 1
            lab, _ = ndimage.label(cand, structure=np.ones((3, 3), int))
            keep = np.zeros(lab.max() + 1, bool)
            keep[np.unique(lab[maps[i] == 2])] = True
            keep[0] = False
            ref_edges = np.where(keep[lab], 255, 0).astype(np.uint8)
            assert np.array_equal(edges[i], ref_edges), (h, w, i)
            ref_lines = R.hough_lines_p(ref_edges, thr, min_len, max_gap)
            assert counts[i] == len(ref_lines), (h, w, i, counts[i], len(ref_lines))
            assert np.array_equal(lines[i, :min(64, counts[i])], ref_lines[:64]), (h, w, i)


def test_host_stage_line_budget_and_threads(harness, tmp_path):
    rng = np.random.default_rng(7)
    maps = np.stack([nms_like_map(rng, 90, 110, 2) for _ in range(5)])
    e1, c1, l1 = run_harness(harness, maps, 8, 4, 3, 2, threads=1, tmp=str(tmp_path))      # room for 2 segments only
    e8, c8, l8 = run_harness(harness, maps, 8, 4, 3, 256, threads=8, tmp=str(tmp_path))
    assert np.array_equal(e1, e8) and np.array_equal(c1, c8) and c8.max() > 2
    for i in range(5):
        assert np.array_equal(l1[i, :2], l8[i, :2]) and (l8[i, c8[i]:] == -1).all()

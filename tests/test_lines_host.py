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

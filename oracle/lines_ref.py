"""CPU oracle of leading-line detection - TEST INFRASTRUCTURE ONLY (never imported by facet_amd/).

Restates, in numpy / plain Python, what reference analyzers/composition.py:191-261 asks OpenCV for:
    gray = cv2.cvtColor(img, COLOR_BGR2GRAY); blurred = cv2.GaussianBlur(gray, (5, 5), 0); edges = cv2.Canny(blurred, 50, 150)
    lines = cv2.HoughLinesP(edges, 1, np.pi / 180, 80, minLineLength=int(min(h, w) * 0.15), maxLineGap=20)
and the scoring of the segments (:231-256).

[DEP-KNOWLEDGE] cv2 is not installed here and cannot be fetched: the four operations follow OpenCV's documented algorithms
(8-bit GaussianBlur in 8.8 fixed point with the [1 4 6 4 1]/16 kernel and BORDER_REFLECT_101; Canny with a 3x3 Sobel under
BORDER_REPLICATE, L1 magnitude, tan(22.5) in 15-bit fixed point, thresholds low < m, high < m; the progressive probabilistic
Hough transform of Matas et al. with OpenCV's multiply-with-carry generator) -> "parity unpinned" against cv2 itself. The
segment scoring IS pinned (tests/golden/host_golden.json, made by the reference's function with its cv2 calls mocked).
Written independently of facet_amd/csrc/kernels_lines.hip (array-at-a-time numpy vs per-pixel kernels; connected-component
labelling vs flood fill), so agreement between the two checks both."""
import math

import numpy as np
from scipy import ndimage

from .technical_ref import bgr2gray


def gaussian5_u8(gray):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(gray.astype(np.int64), 2, mode="reflect")            # numpy 'reflect' = BORDER_REFLECT_101
    h, w = gray.shape
    rows = sum(k[i] * p[:, i:i + w] for i in range(5))
    full = sum(k[i] * rows[i:i + h, :] for i in range(5))
    return ((full + 128) >> 8).astype(np.uint8)


def sobel3(gray):
    p = np.pad(gray.astype(np.int32), 1, mode="edge")               # BORDER_REPLICATE
    h, w = gray.shape
    win = lambda dy, dx: p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]     # noqa: E731
    dx = (win(-1, 1) + 2 * win(0, 1) + win(1, 1)) - (win(-1, -1) + 2 * win(0, -1) + win(1, -1))
    dy = (win(1, -1) + 2 * win(1, 0) + win(1, 1)) - (win(-1, -1) + 2 * win(-1, 0) + win(-1, 1))
    return dx, dy


def canny_u8(gray, low=50, high=150):
    """-> uint8 edge image (0 / 255)."""
    h, w = gray.shape
    dx, dy = sobel3(gray)
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1)                                             # zero magnitude outside the image
    nb = lambda oy, ox: mp[1 + oy:1 + oy + h, 1 + ox:1 + ox + w]     # noqa: E731
    ax, ay = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    same = (dx ^ dy) >= 0                                           # gradient along the main diagonal (down-right / up-left)
    peak = np.zeros((h, w), bool)
    peak |= horiz & (mag > nb(0, -1)) & (mag >= nb(0, 1))
    peak |= vert & (mag > nb(-1, 0)) & (mag >= nb(1, 0))
    peak |= diag & same & (mag > nb(-1, -1)) & (mag > nb(1, 1))
    peak |= diag & ~same & (mag > nb(-1, 1)) & (mag > nb(1, -1))
    cand = peak & (mag > low)
    strong = cand & (mag > high)
    lab, _ = ndimage.label(cand, structure=np.ones((3, 3), int))
    keep = np.zeros(lab.max() + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


class _Rng:
    """OpenCV's RNG: 64-bit multiply-with-carry; HoughLinesP seeds it with all ones."""
    def __init__(self):
        self.state = (1 << 64) - 1

    def uniform(self, a, b):
        if a == b:
            return a
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & ((1 << 64) - 1)
        return (self.state & 0xFFFFFFFF) % (b - a) + a


def hough_lines_p(edges, threshold=80, min_len=0, max_gap=0):
    """Progressive probabilistic Hough transform, rho = 1, theta = pi/180 -> int32 [k,4] (x1,y1,x2,y2), in the order found."""
    h, w = edges.shape
    f32 = np.float32
    theta = f32(np.pi / 180)
    numangle, numrho = 180, int(round((w + h) * 2 + 1))
    ang = np.arange(numangle, dtype=np.float64) * np.float64(theta)
    cos_t, sin_t = np.cos(ang).astype(f32), np.sin(ang).astype(f32)
    accum = np.zeros((numangle, numrho), np.int32)
    rows = np.arange(numangle)
    mask = edges != 0
    nz = [(int(y), int(x)) for y, x in zip(*np.nonzero(mask))]       # row-major, as the image is scanned
    mask = mask.copy()
    half = (numrho - 1) // 2
    rng = _Rng()
    lines = []

    def bins(x, y):                                                  # float32 products and sum, round half to even
        return np.rint(f32(x) * cos_t + f32(y) * sin_t).astype(np.int64) + half

    count = len(nz)
    while count > 0:
        idx = rng.uniform(0, count)
        i, j = nz[idx]
        nz[idx] = nz[count - 1]
        count -= 1
        if not mask[i, j]:
            continue
        r = bins(j, i)
        accum[rows, r] += 1
        vals = accum[rows, r]
        n = int(np.argmax(vals))                                     # first maximum, like a strict '<' scan
        if vals[n] < threshold:
            continue
        a, b = -float(sin_t[n]), float(cos_t[n])
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag, dx0 = True, (1 if a > 0 else -1)
            dy0 = int(np.rint(f32(f32(b) * f32(65536.0)) / f32(abs(a))))
            y0 = (y0 << 16) + (1 << 15)
        else:
            xflag, dy0 = False, (1 if b > 0 else -1)
            dx0 = int(np.rint(f32(f32(a) * f32(65536.0)) / f32(abs(b))))
            x0 = (x0 << 16) + (1 << 15)
        ends = [None, None]
        for k in range(2):
            gap, x, y = 0, x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if j1 < 0 or j1 >= w or i1 < 0 or i1 >= h:
                    break
                if mask[i1, j1]:
                    gap, ends[k] = 0, (j1, i1)
                else:
                    gap += 1
                    if gap > max_gap:
                        break
                x, y = x + dx, y + dy
        good = abs(ends[1][0] - ends[0][0]) >= min_len or abs(ends[1][1] - ends[0][1]) >= min_len
        for k in range(2):
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if mask[i1, j1]:
                    if good:
                        accum[rows, bins(j1, i1)] -= 1
                    mask[i1, j1] = False
                if (j1, i1) == ends[k]:
                    break
                x, y = x + dx, y + dy
        if good:
            lines.append((ends[0][0], ends[0][1], ends[1][0], ends[1][1]))
    return np.array(lines, np.int32).reshape(-1, 4)


def score_lines(lines, h, w):
    """analyzers/composition.py:226-261 (pinned by tests/golden/host_golden.json)."""
    if lines is None or len(lines) == 0:
        return {"leading_lines_score": 0, "line_count": 0}
    total, valid = 0, 0
    for x1, y1, x2, y2 in np.asarray(lines, np.int32):
        length = np.sqrt((x2 - x1) ** 2 + (y2 - y1) ** 2)
        angle = abs(np.degrees(np.arctan((y2 - y1) / (x2 - x1)))) if x2 - x1 != 0 else 90
        total += (length / np.sqrt(h ** 2 + w ** 2)) * 10 * (1.5 if 15 <= angle <= 75 else 1.0)
        valid += 1
    return {"leading_lines_score": round(min(10.0, total / max(1, valid) * 2), 2), "line_count": len(lines)}


def detect_leading_lines(img_bgr):
    h, w = img_bgr.shape[:2]
    edges = canny_u8(gaussian5_u8(bgr2gray(img_bgr)), 50, 150)
    lines = hough_lines_p(edges, 80, int(min(h, w) * 0.15), 20)
    return score_lines(lines, h, w), lines, edges

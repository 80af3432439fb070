"""Rate of fe_leading_lines on a batch of 1024x1024 images (GPU scans + host Hough threads). usage: perf_lines.py [n] [hw]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facet_amd import Engine
from facet_amd.weights import synthetic_images

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
e = Engine(0)
rng = np.random.default_rng(0)
# photo-like content: smooth gradients + a few bars (pure noise has an edge on every other pixel, which no photo has)
yy, xx = np.mgrid[:hw, :hw]
base = (96 + 60 * np.sin(xx / 90.0) + 50 * np.cos(yy / 70.0)).astype(np.int32)
imgs = np.empty((n, hw, hw, 3), np.uint8)
for i in range(n):
    im = np.repeat(base[..., None], 3, 2) + rng.integers(-3, 4, (hw, hw, 3))
    for k in range(6):
        t = int(rng.integers(50, hw - 50))
        im[t:t + 4, 40:hw - 40] += 90
        im[40:hw - 40, t:t + 4] -= 60
    imgs[i] = np.clip(im, 0, 255)
for name, batch in (("structured", imgs), ("noise", synthetic_images(3, n, hw, hw)[..., ::-1].copy())):
    e.leading_lines(batch[:2])
    t0 = time.time(); lines, edges = e.leading_lines(batch, want_edges=True); dt = time.time() - t0
    print(f"{name}: {n} x {hw}x{hw}: {dt*1e3:.1f} ms  {n/dt:.1f} images/s  edge px/img {edges.astype(bool).sum()/n:.0f}  lines/img {np.mean([len(l) for l in lines]):.1f}", flush=True)
    t0 = time.time(); e.leading_lines(batch, threshold=10**9); dt2 = time.time() - t0      # threshold never reached: scans + hysteresis + votes only
    print(f"   without segment extraction: {dt2*1e3:.1f} ms", flush=True)
e.close()

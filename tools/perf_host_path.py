"""Host-buffer vs device-resident throughput of the image entry points (PCIe-inclusive rates for DESIGN.md)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from facet_amd._lib import Engine, FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_SAMP, FE_MODEL_U2NETP, FE_MODEL_AESTHETIC
from facet_amd import weights as W

e = Engine(0, arena_bytes=70 << 30)
e.set_microbatch(32)
e.load_weights(FE_MODEL_TOPIQ, W.synthetic_state_dict("topiq", 1))
e.load_weights(FE_MODEL_CLIP, W.synthetic_state_dict("clip", 2))
e.load_weights(FE_MODEL_AESTHETIC, W.synthetic_state_dict("aesthetic", 3))
e.load_weights(FE_MODEL_U2NETP, W.synthetic_state_dict("u2netp", 4))
e.load_weights(FE_MODEL_SAMP, W.synthetic_state_dict("samp_net", 5))

def timeit(fn, reps=3):
    fn()
    e.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    e.sync()
    return (time.perf_counter() - t) / reps

for name, n, hw in (("topiq", 128, 1024), ("clip", 256, 512), ("samp", 256, 512), ("ensemble", 128, 1024)):
    imgs = W.synthetic_images(7, n, hw, hw)
    d = e.dev_alloc(imgs.nbytes)
    e.h2d(d, imgs)
    dv = (d, n, hw, hw)
    if name == "topiq":
        fh = lambda: e.topiq_score(imgs); fd = lambda: e.topiq_score(dv)
    elif name == "clip":
        fh = lambda: e.clip_encode_images(imgs); fd = lambda: e.clip_encode_images(dv)
    elif name == "samp":
        fh = lambda: e.samp_score_images(imgs); fd = lambda: e.samp_score_images(dv)
    else:
        fh = lambda: e.ensemble_score(imgs); fd = lambda: e.ensemble_score(dv)
    th, td = timeit(fh), timeit(fd)
    print(f"{name:9s} n={n} {hw}x{hw}: host-buffer {n/th:8.1f} img/s   device-resident {n/td:8.1f} img/s   ratio {td/th:.3f}", flush=True)
    e.dev_free(d)

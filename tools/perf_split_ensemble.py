"""Does SAMP-Net / U2-Net-P (small launches) hide under TOPIQ + CLIP when it runs on a second context? usage: perf_split_ensemble.py [n] [prec]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FACET_AMD_SYNTHETIC"] = "1"
import numpy as np
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
a = Engine(0, arena_bytes=72 << 30, precision=prec)
b = Engine(0, arena_bytes=16 << 30, precision=prec)
for mid, name in ((FE_MODEL_TOPIQ, "topiq"), (FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"), (FE_MODEL_SAMP, "samp_net")):
    a.load_weights(mid, synthetic_state_dict(name, 3))
for mid, name in ((FE_MODEL_U2NETP, "u2netp"), (FE_MODEL_SAMP, "samp_net")):
    b.load_weights(mid, synthetic_state_dict(name, 3))
a.set_microbatch(32); b.set_microbatch(32)
imgs = np.random.default_rng(1).integers(0, 256, (n, 1024, 1024, 3), dtype=np.uint8)
d = a.dev_alloc(imgs.nbytes); a.h2d(d, imgs); dev = (d, n, 1024, 1024)

def timed(fn, reps=2):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps

a.ensemble_select(7)
t_one = timed(lambda: a.ensemble_score(dev))
a.ensemble_select(3); b.ensemble_select(4)
t_tc = timed(lambda: a.ensemble_score(dev))
t_s = timed(lambda: b.ensemble_score(dev))
def both():
    th = threading.Thread(target=lambda: b.ensemble_score(dev)); th.start()
    a.ensemble_score(dev); th.join()
t_two = timed(both)
print(f"{prec} n={n}: one context {n/t_one:.1f} img/s | topiq+clip alone {n/t_tc:.1f} | samp alone {n/t_s:.1f} | two contexts {n/t_two:.1f} img/s")

"""SAMP-Net + U2-Net-P throughput vs micro-batch (many of its convs run on 7x7..28x28 maps: launch- and tile-quantisation-bound)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.weights import synthetic_state_dict, synthetic_images
eng = Engine(0, arena_bytes=80 << 30)
eng.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", 7))
eng.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", 7))
n, hw = 512, 256
imgs = synthetic_images(3, n, hw, hw)
d = eng.dev_alloc(imgs.nbytes); eng.h2d(d, imgs)
for mb in (16, 32, 64, 128, 256):
    eng.set_microbatch(mb)
    eng.samp_score_images((d, n, hw, hw)); eng.sync()
    t = time.perf_counter()
    for _ in range(3):
        eng.samp_score_images((d, n, hw, hw))
    eng.sync()
    dt = (time.perf_counter() - t) / 3
    print(f"mb={mb:4d}: {n/dt:8.1f} img/s", flush=True)

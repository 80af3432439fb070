"""CLIP ViT-L/14 throughput vs micro-batch size (tile quantisation: rows = mb*257 over 128-row tiles x 256 CUs)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_CLIP, FE_MODEL_AESTHETIC
from facet_amd.weights import synthetic_state_dict
eng = Engine(0, arena_bytes=40 << 30)
eng.load_weights(FE_MODEL_CLIP, synthetic_state_dict("clip", 9))
eng.load_weights(FE_MODEL_AESTHETIC, synthetic_state_dict("aesthetic", 9))
for mb in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "31,32,47,62,63,64,95,127,128".split(","))]:
    n = mb * 4
    x = np.random.default_rng(0).standard_normal((n, 3, 224, 224), dtype=np.float32)
    d = eng.dev_alloc(x.nbytes); eng.h2d(d, x)
    eng.set_microbatch(mb)
    eng.clip_encode_image((d, n)); eng.sync()
    t = time.perf_counter()
    for _ in range(3):
        eng.clip_encode_image((d, n))
    eng.sync()
    dt = (time.perf_counter() - t) / 3
    print(f"mb={mb:4d}: {n/dt:8.1f} img/s  {162.0*n/dt/1e3:6.1f} TFLOP/s", flush=True)
    eng.dev_free(d)

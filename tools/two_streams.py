"""Does running two micro-batch streams concurrently (two contexts, two host threads) raise whole-GPU throughput?"""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_TOPIQ
from facet_amd.weights import synthetic_state_dict, synthetic_images
n, mb, hw = 128, 32, 1024
sd = synthetic_state_dict("topiq", 3)
imgs = synthetic_images(2, 32, hw, hw)
engs = []
for i in range(2):
    e = Engine(0, arena_bytes=70 << 30); e.load_weights(FE_MODEL_TOPIQ, sd); e.set_microbatch(mb)
    d = e.dev_alloc(n * hw * hw * 3)
    import ctypes
    for j in range(0, n, 32): e.h2d(ctypes.c_void_p(d.value + j * hw * hw * 3), imgs)
    engs.append((e, d))
def run(e, d, cnt): e.topiq_score((d, cnt, hw, hw))
for e, d in engs: run(e, d, 32)
t = time.perf_counter(); run(engs[0][0], engs[0][1], n); run(engs[0][0], engs[0][1], n); t1 = time.perf_counter() - t
print(f"one stream : {2*n/t1:.1f} img/s")
t = time.perf_counter()
ths = [threading.Thread(target=run, args=(e, d, n)) for e, d in engs]
[x.start() for x in ths]; [x.join() for x in ths]
t2 = time.perf_counter() - t
print(f"two streams: {2*n/t2:.1f} img/s")

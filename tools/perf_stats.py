"""fe_image_stats throughput on a resident 1024x1024 BGR batch, noise and flat content (flat = worst case for histogram atomics)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from facet_amd._lib import Engine
e = Engine(0, arena_bytes=8 << 30)
e.set_microbatch(32)
n, hw = 128, 1024
for kind in ("noise", "flat"):
    imgs = np.random.default_rng(1).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8) if kind == "noise" else np.full((n, hw, hw, 3), 128, np.uint8)
    d = e.dev_alloc(imgs.nbytes); e.h2d(d, imgs)
    e.image_stats((d, n, hw, hw)); e.sync()
    t = time.perf_counter()
    for _ in range(3):
        e.image_stats((d, n, hw, hw))
    e.sync()
    dt = (time.perf_counter() - t) / 3
    alg = n * hw * hw * (3 + 1 + 1)          # read BGR, write gray, read gray
    print(f"{kind:6s}: {n/dt:9.1f} img/s  {dt*1e3/n*1e3:7.1f} us/img  algorithmic {alg/dt/1e9:7.1f} GB/s", flush=True)
    e.dev_free(d)

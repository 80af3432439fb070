"""Developer timing of the fused TOPIQ gate (facet_amd/csrc/kernels_gate.hip) through its test hook: n images of h x w x 64.
Build the library with `make -C facet_amd/csrc CXXFLAGS+=-DG64_STAMPS` to get per-stage ticks of workgroup 0 on stderr."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from facet_amd import Engine

n, h, w = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 512, 512)))
prec = sys.argv[4] if len(sys.argv) > 4 else "f16"
rng = np.random.default_rng(0)
x = rng.normal(0, 1, (n, 64, h, w)).astype(np.float32)
w0 = rng.normal(0, 1 / 8, (64, 64)).astype(np.float32)
w2 = rng.normal(0, 1 / 24, (64, 64, 3, 3)).astype(np.float32)
w4 = rng.normal(0, 1 / 12, (1, 64, 3, 3)).astype(np.float32)
wx = rng.normal(0, 1 / 8, (64, 64)).astype(np.float32)
b = rng.normal(0, 0.2, 64).astype(np.float32)
e = Engine(0, arena_bytes=8 << 30, precision=prec)
for _ in range(3):
    t0 = time.time()
    y = e.topiq_gate64(x, w0, b, w2, b, w4, 0.1, wx, b)
    print(f"call (upload + kernel + download): {time.time() - t0:.3f} s, out {y.shape}")

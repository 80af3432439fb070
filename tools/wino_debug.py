import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from facet_amd import Engine
e = Engine(0, arena_bytes=4 << 30)
rng = np.random.default_rng(0)
cin, cout, H, W = 256, 256, 14, 14
x = rng.standard_normal((2, cin, H, W)).astype(np.float32)
w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
got = e.conv2d(x, w, stride=1, pad=1)
ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy()
err = np.abs(got - ref)
print("max err", err.max(), "ref max", np.abs(ref).max())
print("err by row   ", np.round(err.max(axis=(0, 1, 3)), 3))
print("err by col   ", np.round(err.max(axis=(0, 1, 2)), 3))
print("err by image ", np.round(err.max(axis=(1, 2, 3)), 3))
print("err by cout/32", np.round(err.max(axis=(0, 2, 3)).reshape(-1, 32).max(axis=1), 3))
# which input channels matter: zero all but a block of input channels
for c0 in (0, 16, 128, 240):
    x2 = np.zeros_like(x); x2[:, c0:c0 + 16] = x[:, c0:c0 + 16]
    g2 = e.conv2d(x2, w, stride=1, pad=1)
    r2 = F.conv2d(torch.from_numpy(x2), torch.from_numpy(w), padding=1).numpy()
    print("cin block", c0, "err", np.abs(g2 - r2).max(), "of", np.abs(r2).max())

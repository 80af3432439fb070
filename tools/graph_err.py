"""Prints engine-vs-oracle relative errors of the synthetic face graphs (sanity for tolerances)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from standins import synthetic_onnx as S
from facet_amd._lib import Engine
from oracle import onnx_ref
e = Engine(0, arena_bytes=6 << 30)
rng = np.random.default_rng(0)
for name, (data, info), x in (
        ("lmk", S.landmark_like(seed=13), rng.integers(0, 256, (4, 3, 192, 192)).astype(np.float32)),
        ("arc1111", S.arcface_iresnet(layers=(1, 1, 1, 1), seed=14), rng.uniform(-1, 1, (4, 3, 112, 112)).astype(np.float32)),
        ("arc_r50", S.arcface_iresnet(seed=5), rng.uniform(-1, 1, (2, 3, 112, 112)).astype(np.float32)),
        ("det320", S.scrfd_like(seed=12, size=320), rng.uniform(-1, 1, (1, 3, 320, 320)).astype(np.float32))):
    e.graph_load(3, data)
    got = e.graph_run(3, x)
    want = onnx_ref.run(data, x)
    errs = [float(np.abs(g - w).max() / max(np.abs(w).max(), 1e-9)) for g, w in zip(got, want)]
    print(name, " ".join(f"{v:.2e}" for v in errs), "| max|want|", " ".join(f"{float(np.abs(w).max()):.3g}" for w in want), flush=True)

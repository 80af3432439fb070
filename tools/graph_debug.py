"""Finds the first node whose engine output differs from the ONNX oracle: every node output is exported as a graph output
(which also disables epilogue fusion - compare with the fused run to tell kernel bugs from fusion bugs)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from standins import synthetic_onnx as S, onnx_writer as W
from facet_amd._lib import Engine
from oracle import onnx_ref

which = sys.argv[1] if len(sys.argv) > 1 else "det"
if which == "det":
    size = 160
    g_data, info = S.scrfd_like(seed=7, size=size)
    x = np.random.default_rng(3).uniform(-1, 1, (1, 3, size, size)).astype(np.float32)
elif which == "lmk":
    g_data, info = S.landmark_like(seed=9)
    x = np.random.default_rng(4).uniform(0, 255, (2, 3, 192, 192)).astype(np.float32)
m = onnx_ref.parse(g_data)
# rebuild the same bytes with all node outputs exported
import standins.onnx_writer as ow
nodes = []
for n in m["nodes"]:
    nodes.append(ow.node(n["op"], n["in"], n["out"], n["name"], **{k: (v if not isinstance(v, np.ndarray) else v) for k, v in n["attr"].items()}))
outs = [(n["out"][0], []) for n in m["nodes"] if n["op"] not in ("Shape", "Gather", "Unsqueeze", "Constant")]
data = ow.model(nodes, m["init"], [(m["inputs"][0], list(x.shape))], outs)
e = Engine(0, arena_bytes=4 << 30)
e.graph_load(3, data)
got = e.graph_run(3, x)
want = onnx_ref.run(data, x)
names = [o[0] for o in outs]
ops = {n["out"][0]: n["op"] for n in m["nodes"]}
bad = 0
for nm, g, w in zip(names, got, want):
    err = float(np.abs(g - w).max()) / max(float(np.abs(w).max()), 1e-6) if g.shape == w.shape else -1
    flag = "" if 0 <= err < 1e-3 else "   <<<<<<"
    if flag:
        bad += 1
    if flag or "-v" in sys.argv:
        print(f"{nm:16s} {ops[nm]:18s} {str(g.shape):22s} {str(w.shape):22s} rel_err {err:.3e}{flag}")
    if bad >= 6:
        break
print("done, bad =", bad)

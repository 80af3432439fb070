"""Face-path throughput on one GPU: SCRFD-style detector on 1024x1024 BGR batches (cv2-style resize to 640 included), ArcFace
and landmark graphs on batches of crops. Synthetic stand-in graphs (facet_amd/synthetic_onnx.py)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from standins import synthetic_onnx as S
from facet_amd._lib import Engine, FE_GRAPH_FACE_DET, FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC

e = Engine(0, arena_bytes=48 << 30)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
e.set_microbatch(mb)
det, dinfo = S.scrfd_like(seed=12, size=640)
lmk, linfo = S.landmark_like(seed=13)
rec, rinfo = S.arcface_iresnet(seed=14)
e.graph_load(FE_GRAPH_FACE_DET, det); e.graph_load(FE_GRAPH_FACE_LMK, lmk); e.graph_load(FE_GRAPH_FACE_REC, rec)
n, hw = 128, 1024
imgs = np.random.default_rng(3).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)
d = e.dev_alloc(imgs.nbytes); e.h2d(d, imgs); dev = (d, n, hw, hw)

def timeit(fn, reps=3):
    fn(); e.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    e.sync()
    return (time.perf_counter() - t) / reps

t = timeit(lambda: e.face_detect(dev, (640, 640), 0.5, 4096))
print(f"detect   n={n} mb={mb}: {n/t:8.1f} img/s  {2*dinfo['macs']*n/t/1e12:6.1f} TFLOP/s ({2*dinfo['macs']/1e9:.1f} GFLOP/img)", flush=True)
m = 256
rng = np.random.default_rng(4)
idx = rng.integers(0, n, m).astype(np.int32)
M = np.zeros((m, 2, 3)); sc = rng.uniform(0.3, 0.8, m)
M[:, 0, 0] = M[:, 1, 1] = sc; M[:, 0, 2] = -rng.uniform(100, 600, m) * sc + 56; M[:, 1, 2] = -rng.uniform(100, 600, m) * sc + 56
t = timeit(lambda: e.face_crops_run(FE_GRAPH_FACE_REC, dev, idx, M, 112, 127.5, 1 / 127.5, True, out_dim=512))
print(f"arcface  m={m} mb={2*mb}: {m/t:8.1f} face/s {2*rinfo['macs']*m/t/1e12:6.1f} TFLOP/s ({2*rinfo['macs']/1e9:.1f} GFLOP/face)", flush=True)
t = timeit(lambda: e.face_crops_run(FE_GRAPH_FACE_LMK, dev, idx, M, 192, 0.0, 1.0, True, out_dim=212))
print(f"landmark m={m} mb={2*mb}: {m/t:8.1f} face/s {2*linfo['macs']*m/t/1e12:6.1f} TFLOP/s ({2*linfo['macs']/1e9:.2f} GFLOP/face)", flush=True)
if "--profile" in sys.argv:
    for name, fn in (("detect", lambda: e.face_detect(dev, (640, 640), 0.5, 4096)),
                     ("arcface", lambda: e.face_crops_run(FE_GRAPH_FACE_REC, dev, idx[:64], M[:64], 112, 127.5, 1 / 127.5, True, out_dim=512))):
        e.profile_enable(True); fn()
        rows = [(r["name"], r["flops"], r["bytes"], r["ms"]) for r in e.profile_records()]
        e.profile_enable(False)
        tot = sum(r[3] for r in rows)
        print(f"--- {name}: {len(rows)} conv launches, {tot:.2f} ms in convs")
        agg = {}
        for nm, fl, by, ms in rows:
            a = agg.setdefault(nm, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
        for nm, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"  {nm:46s} x{cnt:3d} {ms:8.3f} ms {fl/ms/1e9:7.1f} TF/s")

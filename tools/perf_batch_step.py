"""Whole batch step (facet_amd/batch.py::BatchScorer.process_batch = reference _process_batch for a batch) at 1024x1024: engine calls +
host-side assembly of the per-image dicts. usage: perf_batch_step.py [n] [hw]"""
import cProfile, io, json, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facet_amd import Engine
from standins import synthetic_onnx as S
from facet_amd._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.aggregate import AggregatePolicy
from facet_amd.batch import BatchScorer
from facet_amd.face import FaceAnalyzer
from facet_amd.tagger import CLIPTagger
from facet_amd.weights import synthetic_state_dict, synthetic_images

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
e = Engine(0, arena_bytes=72 << 30)
e2 = Engine(0, arena_bytes=24 << 30)      # second context: statistics / faces / lines run beside the ensemble (BatchScorer aux_engine)
for mid, name in ((FE_MODEL_TOPIQ, "topiq"), (FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"), (FE_MODEL_SAMP, "samp_net")):
    e.load_weights(mid, synthetic_state_dict(name, 4))
models = {"det": S.scrfd_like(seed=12, size=640)[0], "lmk": S.landmark_like(seed=13)[0], "rec": S.arcface_iresnet(layers=(3, 4, 14, 3), seed=14)[0]}
fa = FaceAnalyzer(min_confidence=0.5, min_face_size=10, engine=e, models=models)
fa.face_app.max_faces = 8
fa2 = FaceAnalyzer(min_confidence=0.5, min_face_size=10, engine=e2, models=models)
fa2.face_app.max_faces = 8
vocab = {f"tag{i}": [f"p{i}a", f"p{i}b"] for i in range(40)}
tg = CLIPTagger(config=types.SimpleNamespace(get_tag_vocabulary=lambda: vocab, get_art_tags=lambda: set()))
names = [t for t, d in vocab.items() for _ in d]
tg.set_text_embeddings(names, np.random.default_rng(1).standard_normal((len(names), 768)).astype(np.float32))
pol = AggregatePolicy(json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aggregate_golden.json")))["rich"]["config"])
imgs = synthetic_images(6, n, hw, hw)
# photo-like content for the leading-lines leg (uniform noise has an edge on every third pixel, which no photograph has)
yy, xx = np.mgrid[:hw, :hw]
base = (96 + 60 * np.sin(xx / 90.0) + 50 * np.cos(yy / 70.0)).astype(np.int32)
rng = np.random.default_rng(0)
photo = np.empty_like(imgs)
for i in range(n):
    im = np.repeat(base[..., None], 3, 2) + rng.integers(-3, 4, (hw, hw, 3))
    for k in range(6):
        t = int(rng.integers(50, hw - 50))
        im[t:t + 4, 40:hw - 40] += 90
        im[40:hw - 40, t:t + 4] -= 60
    photo[i] = np.clip(im, 0, 255)
e.set_microbatch(32)
for label, kw in (("one context: models+stats+faces+tags+aggregate", dict(policy=pol, face_analyzer=fa)),
                  ("two contexts (aux_engine): models || stats+faces, +tags+aggregate", dict(policy=pol, face_analyzer=fa2, aux_engine=e2)),
                  ("one context ... + leading lines", dict(policy=pol, detect_lines=True, face_analyzer=fa)),
                  ("two contexts ... + leading lines", dict(policy=pol, detect_lines=True, face_analyzer=fa2, aux_engine=e2))):
    bs = BatchScorer(e, tagger=tg, **kw)
    bs.process_batch(imgs[:8])
    for kind, batch in (("noise images", imgs), ("photo-like images", photo)):
        t0 = time.time(); out = bs.process_batch(batch); dt = time.time() - t0
        print(f"{label}, {kind}: {n} x {hw}x{hw}: {dt*1e3:.0f} ms = {n/dt:.1f} images/s   faces/img {np.mean([r['face_count'] for r in out]):.2f}", flush=True)
pr = cProfile.Profile(); pr.enable(); bs.process_batch(imgs); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(22); print(st.getvalue()[:4500])
fa.face_app.unload(); fa2.face_app.unload(); e2.close(); e.close()

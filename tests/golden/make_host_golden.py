"""Generates tests/golden/host_golden.json by RUNNING THE REFERENCE'S OWN pure-Python functions in the build container
(/root/reference is importable here; it does not exist on the GPU box, so the outputs are committed as a fixture).

Covered (reference file:line): PyIQAScorer._normalize_score and ._preprocess_image (models/pyiqa_scorer.py:166-195, 137-164),
SAMPNetScorer.score_batch's post-processing (models/samp_net.py:1016-1043, driven with a fake model that returns fixed
tensors), CLIPTagger.get_tags_from_embedding / get_tags_with_scores / is_artwork (models/tagger.py:77-158, with hand-made text
embeddings), FaceAnalyzer.calculate_ear / compute_avg_ear (analyzers/face.py:241-257; cv2 is absent, so a stub module named cv2
is put in sys.modules for the import only - none of the pinned functions touches it).
Nothing from the reference is copied: only inputs (seeded) and the outputs the reference code returned are stored.

    python tests/golden/make_host_golden.py
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

REF = "/root/reference"
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))        # import-time dependency of analyzers/face.py only

out = {}

# ---- PyIQAScorer ----------------------------------------------------------------------------------------------------
from models.pyiqa_scorer import PyIQAScorer            # noqa: E402
s = PyIQAScorer("topiq", device="cpu")
raws = [-0.5, 0.0, 1e-4, 0.123456, 0.5, 0.731, 0.99995, 1.0, 1.7, float("nan")]
out["normalize_score"] = {"raw": [None if r != r else r for r in raws], "out": []}
for r in raws:
    try:
        v = s._normalize_score(r)
        out["normalize_score"]["out"].append(None if v != v else float(v))
    except Exception as e:       # record the failure mode too
        out["normalize_score"]["out"].append("error:" + type(e).__name__)
pre = []
rng = np.random.default_rng(17)
for (h, w, mode) in ((100, 120, "RGB"), (600, 2048, "RGB"), (1500, 900, "RGB"), (64, 64, "L"), (1024, 1024, "RGB"), (1025, 300, "RGBA")):
    ch = {"RGB": 3, "L": 1, "RGBA": 4}[mode]
    arr = rng.integers(0, 256, (h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
    t = s._preprocess_image(Image.fromarray(arr, mode))
    a = t.numpy() if hasattr(t, "numpy") else np.asarray(t)
    pre.append({"h": h, "w": w, "mode": mode, "seed_order": len(pre), "shape": list(a.shape), "dtype": str(a.dtype),
                "sha256": hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(), "mean": float(a.mean()), "first": a.ravel()[:8].tolist()})
out["preprocess_image"] = pre

# ---- SAMPNetScorer post-processing ------------------------------------------------------------------------------------
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
tv = sys.modules["torchvision"]
if not hasattr(tv, "transforms"):
    tv.transforms = types.ModuleType("torchvision.transforms"); sys.modules["torchvision.transforms"] = tv.transforms
    tv.models = types.ModuleType("torchvision.models"); sys.modules["torchvision.models"] = tv.models
from models import samp_net as ref_samp                # noqa: E402
g = torch.Generator().manual_seed(5)
pw = torch.randn(6, 8, generator=g) * 2
at = torch.sigmoid(torch.randn(6, 6, generator=g))
sd = torch.softmax(torch.randn(6, 5, generator=g) * 1.5, dim=1)
sd[0] = torch.tensor([0, 0, 0, 0, 1.0]); sd[1] = torch.tensor([1.0, 0, 0, 0, 0])        # the clamps
sc = ref_samp.SAMPNetScorer.__new__(ref_samp.SAMPNetScorer)
sc.preprocess = lambda img: img
sc.saliency_detector = types.SimpleNamespace(detect=lambda x: None)
sc.model = lambda x, sal: (pw, at, sd)
dicts = sc.score_batch([torch.zeros(1, 3, 2, 2) for _ in range(6)])
out["samp_postprocess"] = {"pattern_logits": pw.tolist(), "attributes": at.tolist(), "score_dist": sd.tolist(), "dicts": dicts}

# ---- CLIPTagger ------------------------------------------------------------------------------------------------------
from models.tagger import CLIPTagger                    # noqa: E402
vocab = {"portrait": ["a person", "a face", "a portrait"], "landscape": ["a mountain", "a valley"], "painting": ["a painting"],
         "night": ["the night sky", "stars"], "food": ["a meal"], "statue": ["a statue", "a sculpture"]}
cfg = types.SimpleNamespace(get_tag_vocabulary=lambda: vocab, get_art_tags=lambda: {"painting", "statue"})
tg = CLIPTagger(clip_model=None, device="cpu", config=cfg)
names = [t for t, d in vocab.items() for _ in d]
rng = np.random.default_rng(23)
text = rng.standard_normal((len(names), 768)).astype(np.float32)
text /= np.linalg.norm(text, axis=1, keepdims=True)
tg.tag_names = names
tg.text_embeddings = torch.from_numpy(text)
cases = []
for k in range(8):
    mix = rng.uniform(0, 1, len(names)).astype(np.float32) * (rng.uniform(0, 1, len(names)) > 0.5)
    emb = (mix @ text + 0.05 * rng.standard_normal(768)).astype(np.float32)
    emb /= np.linalg.norm(emb)
    b = emb.tobytes()
    cases.append({"embedding_seed_order": k,
                  "tags_default": tg.get_tags_from_embedding(b), "tags_t22_m5": tg.get_tags_from_embedding(b, threshold=0.22, max_tags=5),
                  "tags_t05_m3": tg.get_tags_from_embedding(b, threshold=0.05, max_tags=3), "with_scores": tg.get_tags_with_scores(b, threshold=0.1),
                  "is_artwork": bool(tg.is_artwork(b, threshold=0.2)), "embedding": emb.tolist()})
out["tagger"] = {"vocabulary": vocab, "art_tags": ["painting", "statue"], "text_embeddings": text.tolist(), "cases": cases,
                 "none_bytes": tg.get_tags_from_embedding(None)}

# ---- FaceAnalyzer EAR ----------------------------------------------------------------------------------------------------
from analyzers.face import FaceAnalyzer                 # noqa: E402
ears = []
for k in range(6):
    lm = rng.uniform(0, 200, (106, 2)).astype(np.float32)
    if k == 5:
        lm[:] = 0
    ears.append({"landmarks": lm.tolist(), "left": float(FaceAnalyzer.calculate_ear(lm, FaceAnalyzer.LEFT_EYE_INDICES)),
                 "right": float(FaceAnalyzer.calculate_ear(lm, FaceAnalyzer.RIGHT_EYE_INDICES)), "avg": float(FaceAnalyzer.compute_avg_ear(lm))})
out["ear"] = ears

# ---- ModelManager sizing / pass packing (models/model_manager.py:631-648, 715-814; GPU-mode branches) ---------------------------
from models.model_manager import ModelManager          # noqa: E402
mm = ModelManager(types.SimpleNamespace(get_model_config=lambda: {"vram_profile": "16gb", "profiles": {"16gb": {"description": "x"}}}))
hot = ["topiq", "clip", "samp_net", "insightface"]
packs = []
for models in (hot, ["clip", "topiq"], ["insightface", "samp_net", "topiq", "clip"], ["samp_net"], ["clip", "clip_aesthetic", "topiq", "samp_net", "insightface"]):
    for vram in (3.0, 5.0, 7.0, 9.0, 11.0, 24.0, 288.0):
        packs.append({"models": models, "vram": vram, "passes": mm.group_passes_by_vram(list(models), vram)})
out["model_manager"] = {"packs": packs, "vram_gb": {m: mm.get_model_vram(m) for m in hot + ["clip_aesthetic"]},
                        "profiles": {str(v): ModelManager.get_recommended_profile(v) for v in (0, 5.9, 6, 13.9, 14, 19.9, 20, 288)},
                        "quality": {str(v): mm.select_quality_model(v) for v in (1.0, 1.9, 2.0, 3.5, 4.0, 288.0)}}

# ---- CompositionAnalyzer.get_placement_data (analyzers/composition.py:111-187; pure arithmetic on the face box) ---------------
from analyzers.composition import CompositionAnalyzer  # noqa: E402
plc = []
for k in range(12):
    w_, h_ = int(rng.integers(200, 2000)), int(rng.integers(200, 2000))
    x1, y1 = rng.uniform(0, w_ * 0.8), rng.uniform(0, h_ * 0.8)
    bbox = [int(x1), int(y1), int(x1 + rng.uniform(5, w_ * 0.2)), int(y1 + rng.uniform(5, h_ * 0.2))]
    cfg_w = None if k % 3 else types.SimpleNamespace(get_composition_weights=lambda: {"power_point_weight": 3.0, "line_weight": 0.5})
    plc.append({"bbox": bbox, "w": w_, "h": h_, "weights": [3.0, 0.5] if cfg_w else [2.0, 1.0],
                "out": CompositionAnalyzer.get_placement_data(np.array(bbox), w_, h_, cfg_w)})
plc.append({"bbox": None, "w": 640, "h": 480, "weights": [2.0, 1.0], "out": CompositionAnalyzer.get_placement_data(None, 640, 480, None)})
out["placement"] = plc
from utils.detection import detect_silhouette          # noqa: E402
sil = []
for hs in (0, 1):
    for tags in (None, "", "silhouette", "portrait,night", "silhouette,group", "landscape", "silhouette,landscape"):
        for fc in (0, 2):
            sil.append({"hist": hs, "tags": tags, "faces": fc, "out": detect_silhouette({"is_silhouette": hs}, tags, fc)})
out["silhouette"] = sil

# ---- CompositionAnalyzer.detect_leading_lines segment scoring + integrate_leading_lines (analyzers/composition.py:231-283) ----------
# cv2 is absent: its four calls in that function are mocked on the stub module - the three image operations pass their input
# through and HoughLinesP hands back the segment array prescribed here, so what is pinned is the scoring that follows them.
import analyzers.composition as ref_comp               # noqa: E402
cvm = ref_comp.cv2
cvm.COLOR_BGR2GRAY = 6
cvm.cvtColor = lambda img, code: img[..., 0]
cvm.GaussianBlur = lambda g, k, s: g
cvm.Canny = lambda g, lo, hi: g
ll = []
for k in range(10):
    h_, w_ = int(rng.integers(100, 1500)), int(rng.integers(100, 1500))
    nl = [0, 1, 2, 5, 17, 40, 3, 8, 1, 60][k]
    segs = None
    if nl:
        segs = np.stack([rng.integers(0, w_, nl), rng.integers(0, h_, nl), rng.integers(0, w_, nl), rng.integers(0, h_, nl)], 1).astype(np.int32)
        if k % 2:
            segs[0, 2] = segs[0, 0]                     # a vertical segment (the x2 == x1 branch)
        if k == 5:
            segs[1] = [10, 10, 10 + 100, 10 + 100]       # exactly 45 degrees
    cvm.HoughLinesP = lambda e, rho, theta, thr, minLineLength=0, maxLineGap=0, _s=segs: None if _s is None else _s[:, None, :].copy()
    res = CompositionAnalyzer.detect_leading_lines(np.zeros((h_, w_, 3), np.uint8))
    ll.append({"h": h_, "w": w_, "lines": None if segs is None else segs.tolist(),
               "out": {"leading_lines_score": float(res["leading_lines_score"]), "line_count": int(res["line_count"])}})
out["leading_lines_scoring"] = ll
out["integrate_leading_lines"] = [{"base": b, "lines": l, "faces": f, "out": float(CompositionAnalyzer.integrate_leading_lines(b, l, f))}
                                  for b in (0.0, 4.5, 9.2, 10.0) for l in (0.0, 3.3, 10.0, 25.0) for f in (False, True)]

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_golden.json")
json.dump(out, open(path, "w"))
print("wrote", path, os.path.getsize(path), "bytes")

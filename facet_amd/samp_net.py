"""Drop-in for reference models/samp_net.py `SAMPNetScorer`, backed by libfacet_engine.so.

Mirrors the reference wrapper's interface (:798-1043): SAMPNetScorer(model_path=None, device='cuda');
.ensure_loaded(); .score(PIL | BGR ndarray | path) -> dict; .score_batch(list) -> list[dict]; attrs .model and
.saliency_detector.model (objects with .cpu()/.to(), poked by ModelManager, model_manager.py:309-315).
Result dict keys, rounding and the pattern-name list are the reference's (:23-32, :957-989). Missing weights ->
warning + seeded synthetic init (the reference warns and runs with random init, :898-900).
"""
import os

import numpy as np

from ._lib import Engine, FE_MODEL_SAMP, FE_MODEL_U2NETP
from .weights import synthetic_state_dict

COMPOSITION_PATTERNS = ['global', 'horizontal', 'vertical', 'triangular', 'surround', 'quarter', 'cross',
                        'rule_of_thirds']


def _softmax(v):
    v = np.asarray(v, np.float32)
    e = np.exp(v - v.max())
    return e / e.sum()


def postprocess(pw_logits, attributes, score_dist):
    """One image: reference samp_net.py:957-989."""
    pw = _softmax(pw_logits)
    sd = np.asarray(score_dist, np.float32)
    idx = int(np.argmax(pw))
    raw = float(np.sum(np.array([1, 2, 3, 4, 5]) * sd))
    comp = max(0.0, min(10.0, (raw - 1) / 4.0 * 10.0))
    return {
        'comp_score': round(comp, 2), 'raw_score': round(raw, 2), 'pattern': COMPOSITION_PATTERNS[idx],
        'pattern_index': idx, 'pattern_weights': {COMPOSITION_PATTERNS[i]: float(pw[i]) for i in range(8)},
        'score_distribution': sd.tolist(), 'attributes': np.asarray(attributes, np.float32).tolist(),
        'power_point_score': round(comp / 2, 2),
    }


class _Handle:
    def __init__(self, owner, model_id):
        self._o, self._id = owner, model_id

    def cpu(self):
        self._o._offload(self._id)
        return self

    def to(self, device):
        (self._o._offload if str(device) == 'cpu' else self._o._resident)(self._id)
        return self

    def eval(self):
        return self


class _Saliency:
    def __init__(self, owner):
        self.model = _Handle(owner, FE_MODEL_U2NETP)
        self._o = owner

    def ensure_loaded(self):
        self._o._resident(FE_MODEL_U2NETP)


def _load_sd(path):
    from .pyiqa_scorer import load_checkpoint
    sd = load_checkpoint(path)
    for k in ('model_state_dict', 'state_dict'):
        if k in sd:
            sd = sd[k]
    return sd


class SAMPNetScorer:
    def __init__(self, model_path=None, device='cuda', engine=None, u2netp_path='pretrained_models/u2netp.pth',
                 synthetic_seed=7):
        self.device = device
        self.model_path = model_path or 'pretrained_models/samp_net.pth'
        self._u2_path = u2netp_path
        self._seed = synthetic_seed
        self._engine = engine if engine is not None else Engine(0)
        self._sd = {}
        for mid, path, name in ((FE_MODEL_SAMP, self.model_path, 'samp_net'), (FE_MODEL_U2NETP, self._u2_path, 'u2netp')):
            if path and os.path.exists(path):
                self._sd[mid] = _load_sd(path)
            else:
                print(f"Warning: Could not load {name} weights from {path}; using seeded synthetic init "
                      "(scores may not be accurate)")
                self._sd[mid] = synthetic_state_dict(name, self._seed)
        self.model = _Handle(self, FE_MODEL_SAMP)
        self.saliency_detector = _Saliency(self)
        self._resident(FE_MODEL_SAMP)

    def _resident(self, mid):
        if not self._engine.loaded(mid):
            self._engine.load_weights(mid, self._sd[mid])

    def _offload(self, mid):
        if self._engine.loaded(mid):
            self._engine.unload(mid)

    def ensure_loaded(self):
        self.saliency_detector.ensure_loaded()

    @staticmethod
    def _to_array(image):
        """-> (uint8 HWC array, is_bgr). PIL / path -> RGB; ndarray is assumed BGR like the reference (:916-921)."""
        from PIL import Image
        if isinstance(image, str):
            image = Image.open(image).convert('RGB')
        if isinstance(image, np.ndarray):
            if image.ndim == 3 and image.shape[2] == 3:
                return np.ascontiguousarray(image, np.uint8), True
            return np.asarray(Image.fromarray(image).convert('RGB'), np.uint8), False
        if not isinstance(image, Image.Image):
            raise ValueError(f"Unsupported image type: {type(image)}")
        if image.mode != 'RGB':
            image = image.convert('RGB')
        return np.asarray(image, np.uint8), False

    def score(self, image):
        return self.score_batch([image])[0]

    def score_batch(self, images):
        self._resident(FE_MODEL_SAMP)
        self._resident(FE_MODEL_U2NETP)
        arrs = [self._to_array(im) for im in images]
        out = [None] * len(images)
        groups = {}
        for i, (a, bgr) in enumerate(arrs):
            groups.setdefault((a.shape, bgr), []).append(i)
        for (shape, bgr), idxs in groups.items():
            pw, at, sd = self._engine.samp_score_images(np.stack([arrs[i][0] for i in idxs]), bgr=bgr)
            for j, i in enumerate(idxs):
                out[i] = postprocess(pw[j], at[j], sd[j])
        return out

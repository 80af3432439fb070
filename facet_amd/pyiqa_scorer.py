"""Drop-in for reference models/pyiqa_scorer.py `PyIQAScorer`, backed by libfacet_engine.so.

Same constructor, methods, return types and error behaviour as the reference wrapper (:78-255):
  PyIQAScorer(model_name='topiq', device=None); .load(); .unload(); .score_image(PIL)->float in [0,10];
  .score_batch(list[PIL])->list[float] (per-image failure -> 5.0 + printed warning); unknown name -> ValueError;
  attrs .model (with .cpu()/.to()), .model_name, .vram_gb, .description.
Differences: score_batch sends same-sized images to the GPU as ONE batched engine call instead of the reference's
per-image Python loop (:245-253); only 'topiq' is served by the HIP engine.
"""
import os

import numpy as np

from ._lib import Engine, EngineError, FE_MODEL_TOPIQ
from .weights import checkpoint_or_synthetic

PYIQA_MODELS = {
    'topiq': {'pyiqa_id': 'topiq_nr', 'vram_gb': 2, 'lower_better': False, 'score_range': (0, 1),
              'description': 'TOPIQ NR - Best accuracy, ResNet50 backbone'},
}

_MAX_INFERENCE_SIZE = 1024  # reference :135


class _ModelHandle:
    """What ModelManager pokes at: `.cpu()` / `.to(device)` (reference model_manager.py:316-318,341-342)."""

    def __init__(self, scorer):
        self._s = scorer

    def cpu(self):
        self._s._offload()
        return self

    def to(self, device):
        if str(device) != 'cpu':
            self._s._ensure_resident()
        else:
            self._s._offload()
        return self

    def eval(self):
        return self


def load_checkpoint(path):
    """state_dict from a local file with a loader that executes nothing from it."""
    if path.endswith('.safetensors'):
        from safetensors.numpy import load_file
        return load_file(path)
    import torch
    sd = torch.load(path, map_location='cpu', weights_only=True)
    if isinstance(sd, dict) and 'params' in sd:
        sd = sd['params']
    return {k: v.numpy() for k, v in sd.items() if hasattr(v, 'numpy')}


class PyIQAScorer:
    def __init__(self, model_name='topiq', device=None, engine=None, weights_path=None, synthetic_seed=3, synthetic=False,
                 gate_act='gelu', weight_blk_act='gelu'):
        if model_name not in PYIQA_MODELS:
            raise ValueError(f"Unknown model '{model_name}'. Available: {', '.join(PYIQA_MODELS)}")
        self.model_name = model_name
        self.model_info = PYIQA_MODELS[model_name]
        self.device = device or 'cuda'
        self.model = None
        self._loaded = False
        self._engine = engine
        self._own_engine = engine is None
        self._weights_path = weights_path or os.environ.get('FACET_AMD_TOPIQ_WEIGHTS')
        self._seed = synthetic_seed
        self._synthetic = synthetic
        self._acts = (gate_act, weight_blk_act)     # pyiqa GatedConv activations (parameter-free; see Engine.topiq_configure)
        self._state = None

    # -- lifecycle --------------------------------------------------------------------------
    def load(self):
        if self._loaded:
            return
        print(f"Loading {self.model_name} ({self.model_info['pyiqa_id']})...")
        self._state = checkpoint_or_synthetic('topiq', self._weights_path, self._synthetic, self._seed, load_checkpoint)
        if self._engine is None:
            idx = int(str(self.device).split(':')[1]) if ':' in str(self.device) else 0
            self._engine = Engine(idx)
        self._engine.topiq_configure(*self._acts)
        self._engine.load_weights(FE_MODEL_TOPIQ, self._state)
        self.model = _ModelHandle(self)
        self._loaded = True
        print(f"  {self.model_name} loaded on {self.device}")

    def _offload(self):
        if self._engine is not None and self._engine.loaded(FE_MODEL_TOPIQ):
            self._engine.unload(FE_MODEL_TOPIQ)

    def _ensure_resident(self):
        if not self._engine.loaded(FE_MODEL_TOPIQ):
            self._engine.topiq_configure(*self._acts)
            self._engine.load_weights(FE_MODEL_TOPIQ, self._state)

    def unload(self):
        if not self._loaded:
            return
        self._offload()
        self.model = None
        self._loaded = False
        print(f"  {self.model_name} unloaded")

    # -- pre/post (reference :137-195) ---------------------------------------------------------
    def _preprocess_image(self, image):
        """PIL -> uint8 HWC RGB array (the /255 and HWC->CHW of the reference happen on the GPU)."""
        from PIL import Image
        if image.mode != 'RGB':
            image = image.convert('RGB')
        w, h = image.size
        long_edge = max(w, h)
        if long_edge > _MAX_INFERENCE_SIZE:
            scale = _MAX_INFERENCE_SIZE / long_edge
            image = image.resize((int(w * scale), int(h * scale)), Image.LANCZOS)
        return np.asarray(image, dtype=np.uint8)

    def _normalize_score(self, raw_score):
        if hasattr(raw_score, 'item'):
            raw_score = raw_score.item()
        raw_score = float(raw_score)
        lo, hi = self.model_info['score_range']
        raw_score = max(float(lo), min(float(hi), raw_score))
        norm = (raw_score - lo) / (hi - lo) if hi > lo else raw_score
        return max(0.0, min(10.0, float(norm * 10.0)))

    # -- scoring ------------------------------------------------------------------------------
    def score_image(self, image):
        if not self._loaded:
            self.load()
        self._ensure_resident()
        arr = self._preprocess_image(image)
        raw = float(self._engine.topiq_score(arr[None])[0])
        if self.model_info['lower_better']:
            lo, hi = self.model_info['score_range']
            raw = float(hi) - raw + float(lo)
        return self._normalize_score(raw)

    def score_batch(self, images):
        if not self._loaded:
            self.load()
        self._ensure_resident()
        scores = [None] * len(images)
        groups = {}
        for i, img in enumerate(images):
            try:
                arr = self._preprocess_image(img)
                groups.setdefault(arr.shape, []).append((i, arr))
            except Exception as e:  # same default substitution as the reference (:251-253)
                print(f"  Warning: Failed to score image: {e}")
                scores[i] = 5.0
        for shape, items in groups.items():
            try:
                raw = self._engine.topiq_score(np.stack([a for _, a in items]))
                for (i, _), r in zip(items, raw):
                    scores[i] = float(self._normalize_score(float(r)))
            except (EngineError, Exception) as e:
                print(f"  Warning: Failed to score image: {e}")
                for i, _ in items:
                    scores[i] = 5.0
        return scores

    @property
    def vram_gb(self):
        return self.model_info['vram_gb']

    @property
    def description(self):
        return self.model_info['description']

"""Drop-in for the hot-path part of reference models/model_manager.py `ModelManager`.

Keeps the call surface the batch processors use: load_model_only / unload_model / unload_all /
evict_cpu_cache / detect_vram / group_passes_by_vram / select_quality_model / get_active_profile / device /
_cache_hits / _cache_misses. One engine context (one GPU) is shared by every wrapper it hands out. The reference
moves modules to CPU RAM between chunks (:237-268, :299-375); here "cached" means the host keeps the state_dict
and the engine re-packs it on the next load (a cache hit), "evicted" means the host copy is dropped too.
"""
import threading

from ._lib import Engine, EngineError

# runtime estimates the reference uses for pass packing (model_manager.py:652-667)
MODEL_VRAM_GB = {'clip': 4, 'clip_aesthetic': 4, 'samp_net': 2, 'insightface': 2, 'topiq': 2}
HOT_PATH_MODELS = ('topiq', 'clip', 'samp_net', 'insightface')


class ModelManager:
    def __init__(self, config=None, engine=None, device_index=0):
        self.config = config
        self._engine = engine
        self._device_index = device_index
        self.device = f'cuda:{device_index}'
        self.models = {}
        self._cpu_cache = {}
        self._cache_hits = 0
        self._cache_misses = 0
        self._lock = threading.Lock()

    @property
    def engine(self):
        if self._engine is None:
            self._engine = Engine(self._device_index)
        return self._engine

    # -- profile / sizing (reference :600-648) -----------------------------------------------------
    @staticmethod
    def detect_vram():
        try:
            import torch
            if torch.cuda.is_available():
                return torch.cuda.get_device_properties(0).total_memory / (1024 ** 3)
        except Exception:
            pass
        return 0.0

    @staticmethod
    def get_recommended_profile(vram_gb):
        return "24gb" if vram_gb >= 20 else "16gb" if vram_gb >= 14 else "8gb" if vram_gb >= 6 else "legacy"

    def get_active_profile(self):
        prof = None
        if self.config is not None and hasattr(self.config, 'get_model_config'):
            prof = self.config.get_model_config().get('vram_profile')
        return prof if prof and prof != 'auto' else self.get_recommended_profile(self.detect_vram())

    def get_model_vram(self, name):
        return MODEL_VRAM_GB.get(name, 4)

    def select_quality_model(self, available_vram):
        return 'topiq' if available_vram >= 2 else 'clip_aesthetic'

    def group_passes_by_vram(self, models, available_vram):
        """First-fit-decreasing packing of models into passes under (vram - 1 GB), as the reference (:768-814)."""
        capacity = available_vram - 1.0
        bins, usage = [], []
        for m in sorted(models, key=self.get_model_vram, reverse=True):
            need = self.get_model_vram(m)
            for i, u in enumerate(usage):
                if u + need <= capacity:
                    bins[i].append(m)
                    usage[i] += need
                    break
            else:
                bins.append([m])
                usage.append(need)
        return bins

    # -- lifecycle (reference :237-268, :393-437) -------------------------------------------------------
    def load_model_only(self, name):
        with self._lock:
            if name in self.models:
                return self.models[name]
            if name in self._cpu_cache:
                self._cache_hits += 1
                obj = self._cpu_cache.pop(name)
                self._restore(name, obj)
            else:
                self._cache_misses += 1
                try:
                    obj = self._create(name)
                except (EngineError, KeyError, ValueError, FileNotFoundError) as e:
                    print(f"Failed to load {name}: {e}")
                    return None  # caller raises (multi_pass.py:344-348)
            self.models[name] = obj
            return obj

    def _create(self, name):
        cfg = self.config.get_model_config() if (self.config is not None and hasattr(self.config, 'get_model_config')) else {}
        if name == 'topiq':
            from .pyiqa_scorer import PyIQAScorer
            s = PyIQAScorer('topiq', device=self.device, engine=self.engine, weights_path=cfg.get('topiq', {}).get('model_path'))
            s.load()
            return s
        if name == 'samp_net':
            from .samp_net import SAMPNetScorer
            return SAMPNetScorer(model_path=cfg.get('samp_net', {}).get('model_path'), device=self.device, engine=self.engine)
        if name == 'clip':
            from .clip import load_clip
            return load_clip(self.engine, cfg.get('clip', {}).get('model_path'))
        if name == 'insightface':
            # reference _load_insightface (model_manager.py:462-480) returns insightface's FaceAnalysis app (.get(img));
            # FaceEngine is that object on the engine. Model files: <root>/models/buffalo_l/*.onnx, never downloaded.
            from .face import FaceEngine, _load_buffalo_l
            fcfg = cfg.get('insightface', {})
            try:
                models = fcfg.get('models') or _load_buffalo_l(fcfg.get('root', '~/.insightface'))
            except FileNotFoundError as e:
                raise EngineError(str(e))
            return FaceEngine(self.engine, models, det_size=(640, 640))
        raise KeyError(f"model '{name}' is not served by the engine (hot-path models: {HOT_PATH_MODELS})")

    @staticmethod
    def _handles(obj):
        if isinstance(obj, dict):
            return [obj['model']]
        hs = [obj.model] if getattr(obj, 'model', None) is not None else []
        if hasattr(obj, 'saliency_detector'):
            hs.append(obj.saliency_detector.model)
        return hs

    def _restore(self, name, obj):
        for h in self._handles(obj):
            h.to(self.device)

    def unload_model(self, name, cache=True):
        with self._lock:
            obj = self.models.pop(name, None)
            if obj is None:
                return
            for h in self._handles(obj):
                h.cpu()
            if cache:
                self._cpu_cache[name] = obj

    def unload_all(self):
        for n in list(self.models):
            self.unload_model(n)

    def evict_cpu_cache(self):
        with self._lock:
            self._cpu_cache.clear()

    def get_loaded_models(self):
        return list(self.models.keys())

"""facet_amd — MI355X-native image-scoring engine behind rlorenzo/facet's model-layer signatures.

The compute lives in libfacet_engine.so (hand-written HIP for gfx950, C ABI in include/facet_engine.h);
this package is the ctypes binding plus host-side mirrors of the reference's Python wrappers
(PyIQAScorer, SAMPNetScorer, CLIP handle, CLIPTagger, ModelManager).
"""
import threading

from ._lib import Engine, EngineError, load_library, LIB_PATH  # noqa: F401

__all__ = ["Engine", "EngineError", "load_library", "LIB_PATH", "default_engine"]

_default = {}
_default_lock = threading.Lock()


def default_engine(device=0):
    """Process-wide engine per device for call sites that, like the reference's stateless helpers (ImageCache(img),
    TechnicalAnalyzer.get_*), have no place to pass one. Raises EngineError without a gfx950 device - there is no CPU path."""
    with _default_lock:
        if device not in _default:
            _default[device] = Engine(device)
        return _default[device]

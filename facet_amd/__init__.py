"""facet_amd — MI355X-native image-scoring engine behind rlorenzo/facet's model-layer signatures.

The compute lives in libfacet_engine.so (hand-written HIP for gfx950, C ABI in include/facet_engine.h);
this package is the ctypes binding plus host-side mirrors of the reference's Python wrappers
(PyIQAScorer, SAMPNetScorer, CLIP handle, CLIPTagger, ModelManager).
"""
from ._lib import Engine, EngineError, load_library, LIB_PATH  # noqa: F401

__all__ = ["Engine", "EngineError", "load_library", "LIB_PATH"]

"""ctypes binding of libfacet_engine.so (C ABI: include/facet_engine.h).

The product path has no CPU fallback: if the shared library or a gfx950 device is missing every
entry point raises EngineError. Only plain pointers and sizes cross the boundary; numpy is used as
the host buffer container.
"""
import ctypes as C
import importlib.util
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FACET_AMD_LIB") or os.path.join(_HERE, "libfacet_engine.so")   # env: developer A/B builds only

FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_SAMP, FE_MODEL_U2NETP, FE_MODEL_AESTHETIC, FE_MODEL_VLM = range(6)
FE_MODEL_SCRFD, FE_MODEL_ARCFACE = 5, 6
FE_RECORD_FLOATS = 789
FE_GRAPH_FACE_DET, FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC = 0, 1, 2
FE_FACE_FLOATS = 739
FE_STATS_DOUBLES = 264
FILTERS = {"lanczos": 1, "bilinear": 2, "bicubic": 3}
FE_PRECISION_RES32 = 16      # or-ed onto a 2-byte precision: fp32 residual streams (include/facet_engine.h fe_precision)
FE_PRECISION_SPLIT3 = 32     # or-ed onto f16: split-operand GEMMs of the CLIP tower ("f16x3")
PRECISION = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2, "float16": 2, "half": 2,
             "bf16+r32": 1 | FE_PRECISION_RES32, "f16+r32": 2 | FE_PRECISION_RES32, "f16x3": 2 | FE_PRECISION_RES32 | FE_PRECISION_SPLIT3}
PRECISION_NAME = {0: "f32", 1: "bf16", 2: "f16", 1 | FE_PRECISION_RES32: "bf16+r32", 2 | FE_PRECISION_RES32: "f16+r32",
                  2 | FE_PRECISION_RES32 | FE_PRECISION_SPLIT3: "f16x3"}
ACT = {"none": 0, None: 0, "relu": 1, "gelu": 2, "sigmoid": 3, "softplus": 5}


class EngineError(RuntimeError):
    pass


_lib = None
_lib_lock = threading.Lock()

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol declared in include/facet_engine.h is listed here and
# tests/test_abi.py checks the two stay in sync.
SIGNATURES = {
    "fe_create": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "fe_destroy": (None, [C.c_void_p]),
    "fe_last_error": (C.c_char_p, [C.c_void_p]),
    "fe_version": (C.c_char_p, []),
    "fe_sync": (C.c_int, [C.c_void_p]),
    "fe_set_microbatch": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_model_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "fe_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fe_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fe_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fe_timer_start": (C.c_int, [C.c_void_p]),
    "fe_timer_stop": (C.c_int, [C.c_void_p, _f32p]),
    "fe_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_profile_count": (C.c_int, [C.c_void_p]),
    "fe_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), _f32p]),
    "fe_flops_reset": (C.c_int, [C.c_void_p]),
    "fe_flops_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "fe_flops_get_executed": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "fe_flops_get_half": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "fe_weights_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_weights_set": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, _f32p, _i64p, C.c_int]),
    "fe_weights_commit": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_model_unload": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_model_loaded": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_op_conv2d": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int,
                               C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "fe_op_topiq_gate64": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float,
                                     _f32p, _f32p, C.c_int, C.c_int, _f32p]),
    "fe_op_conv3x3_c64": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p,
                                    _f32p, _f32p]),
    "fe_op_maxpool2d": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, _f32p]),
    "fe_op_bilinear": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "fe_op_adaptive_avgpool": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _f32p]),
    "fe_op_layernorm": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_float, _f32p]),
    "fe_set_conv_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_bench_conv": (C.c_int, [C.c_void_p] + [C.c_int] * 12 + [_f32p]),
    "fe_topiq_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "fe_topiq_f32_below": (C.c_int, [C.c_void_p, C.c_longlong]),
    "fe_topiq_feature_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "fe_ensemble_select": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_ensemble_score_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int)]),
    "fe_topiq_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "fe_topiq_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "fe_clip_encode_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _f32p, _f32p, _f32p]),
    "fe_resize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p]),
    "fe_clip_encode_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p]),
    "fe_samp_score_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p,
                                       _f32p, _f32p]),
    "fe_clip_encode_text": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, _f32p]),
    "fe_tag_similarities": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int, C.c_int, _f32p]),
    "fe_vlm_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int)]),
    "fe_vlm_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "fe_vlm_prefill": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), _f32p]),
    "fe_vlm_decode_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), _f32p]),
    "fe_vlm_vision_configure": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "fe_vlm_encode_images": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                                       C.POINTER(C.c_int32), C.c_int, _f32p]),
    "fe_vlm_prefill_images": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                        C.POINTER(C.c_int32), _f32p]),
    "fe_vlm_generate": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "fe_ensemble_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p,
                                    C.POINTER(C.c_int)]),
    "fe_u2netp_saliency": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "fe_samp_forward": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p]),
    "fe_onnx_probe": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _i64p,
                                C.c_char_p, C.c_int]),
    "fe_graph_load": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "fe_graph_unload": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_graph_loaded": (C.c_int, [C.c_void_p, C.c_int]),
    "fe_graph_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _i64p, C.POINTER(C.c_int)]),
    "fe_graph_run": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fe_graph_output_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, _i64p, C.POINTER(C.c_int)]),
    "fe_graph_output_copy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _f32p, C.c_size_t]),
    "fe_face_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                 _f32p, C.POINTER(C.c_int), _f32p]),
    "fe_face_crops_run": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                    C.POINTER(C.c_double), C.c_int, C.c_float, C.c_float, C.c_int, _f32p, C.c_int, C.c_void_p]),
    "fe_face_analyze": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                  C.c_int, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fe_image_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
    "fe_roi_laplacian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_double)]),
    "fe_cv_resize_linear_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "fe_aesthetic_score": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]),
    "fe_swap_rb_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "fe_leading_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
}


def _share_hip_runtime_with_torch():
    """One HIP runtime per process. PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64; the engine links the system
    ROCm's. Loaded in the order engine -> torch, the process ends up with two runtimes and torch reports "No HIP GPUs are
    available" (measured: tools/hip_coexist_probe.py); in the order torch -> engine both share torch's copy and work. Facet runs
    torch models (and RCCL through torch.distributed) next to this engine, so when a torch wheel with a bundled runtime is
    installed its copy is mapped first - without importing torch - which makes the import order irrelevant. FACET_AMD_SYSTEM_HIP=1
    keeps the system runtime."""
    if os.environ.get("FACET_AMD_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass            # no bundled runtime to share: the system one is used


def load_library():
    """dlopen the in-tree engine; raises EngineError (never falls back) when it is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise EngineError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise EngineError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def onnx_probe(onnx_bytes):
    """Host-only parse of an .onnx buffer -> dict(nodes, initializers, outputs, input_dims). Raises EngineError on a bad file."""
    lib = load_library()
    buf = C.create_string_buffer(bytes(onnx_bytes), len(onnx_bytes))
    nn, ni, no = C.c_int(), C.c_int(), C.c_int()
    dims = (C.c_int64 * 4)()
    err = C.create_string_buffer(512)
    if lib.fe_onnx_probe(buf, len(onnx_bytes), C.byref(nn), C.byref(ni), C.byref(no), dims, err, 512) != 0:
        raise EngineError(err.value.decode())
    return {"nodes": nn.value, "initializers": ni.value, "outputs": no.value, "input_dims": list(dims)}


class Engine:
    """One engine context = one GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device=0, arena_bytes=0, precision="f32"):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.fe_create(int(device), int(arena_bytes), C.byref(h))
        if rc != 0:
            raise EngineError("fe_create failed: " + (self.lib.fe_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        if PRECISION[precision]:
            self.set_precision(precision)

    def close(self):
        if getattr(self, "h", None):
            self.lib.fe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise EngineError((self.lib.fe_last_error(self.h) or b"?").decode())

    # -- misc -------------------------------------------------------------------------------
    def sync(self):
        self._ck(self.lib.fe_sync(self.h))

    def set_precision(self, precision):
        """Precision of the models loaded AFTER this call: 'f32' (default, the reference's CPU numerics), 'f16' (what the reference
        runs CLIP in on a GPU), 'bf16' (BASELINE configs[3]); '+r32' keeps the residual streams in fp32 ('f16+r32', 'bf16+r32')."""
        self._ck(self.lib.fe_set_precision(self.h, PRECISION[precision]))

    def model_precision(self, model):
        """'f32' / 'f16' / 'bf16' (+ '+r32') of a loaded model, None when it is not loaded."""
        v = self.lib.fe_model_precision(self.h, int(model))
        return PRECISION_NAME.get(v)

    def set_microbatch(self, n):
        self._ck(self.lib.fe_set_microbatch(self.h, int(n)))

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._ck(self.lib.fe_dev_alloc(self.h, int(nbytes), C.byref(p)))
        return p

    def dev_free(self, p):
        self._ck(self.lib.fe_dev_free(self.h, p))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(self.lib.fe_memcpy_h2d(self.h, dptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        self._ck(self.lib.fe_memcpy_d2h(self.h, arr.ctypes.data_as(C.c_void_p), dptr, arr.nbytes))

    def timer_start(self):
        self._ck(self.lib.fe_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.fe_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._ck(self.lib.fe_profile_enable(self.h, 1 if on else 0))

    def profile_records(self):
        out = []
        n = self.lib.fe_profile_count(self.h)
        buf = C.create_string_buffer(160)
        for i in range(n):
            fl, by, ms = C.c_double(), C.c_double(), C.c_float()
            self._ck(self.lib.fe_profile_get(self.h, i, buf, 160, C.byref(fl), C.byref(by), C.byref(ms)))
            out.append({"name": buf.value.decode(), "flops": fl.value, "bytes": by.value, "ms": ms.value})
        return out

    def flops_executed(self):
        f = C.c_double()
        self._ck(self.lib.fe_flops_get_executed(self.h, C.byref(f)))
        return f.value

    def flops_half(self):
        """The part of flops() issued on the 2-byte (bf16 / fp16) matrix instructions."""
        f = C.c_double()
        self._ck(self.lib.fe_flops_get_half(self.h, C.byref(f)))
        return f.value

    def flops_reset(self):
        self._ck(self.lib.fe_flops_reset(self.h))

    def flops(self):
        v = C.c_double()
        self._ck(self.lib.fe_flops_get(self.h, C.byref(v)))
        return v.value

    # -- weights ----------------------------------------------------------------------------
    def load_weights(self, model, state_dict):
        """state_dict: name -> array-like (numpy or torch CPU tensor), PyTorch checkpoint layout."""
        self._ck(self.lib.fe_weights_begin(self.h, model))
        for name, t in state_dict.items():
            a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
            if a.dtype.kind not in "fiu" or a.ndim > 6:
                continue
            a, ap = _f32(a)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._ck(self.lib.fe_weights_set(self.h, model, name.encode(), ap, shape, a.ndim))
        self._ck(self.lib.fe_weights_commit(self.h, model))

    def unload(self, model):
        self._ck(self.lib.fe_model_unload(self.h, model))

    def loaded(self, model):
        return bool(self.lib.fe_model_loaded(self.h, model))

    # -- ops --------------------------------------------------------------------------------
    def conv2d(self, x, w, scale=None, shift=None, res=None, res_after_act=False, stride=1, pad=0, dil=1, act=None):
        x, xp = _f32(x)
        w, wp = _f32(w)
        n, c, h, ww = x.shape
        cout, cin, kh, kw = w.shape
        assert cin == c
        ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (ww + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        y = np.empty((n, cout, ho, wo), np.float32)
        sp = hp = rp = None
        if scale is not None:
            scale, sp = _f32(scale)
        if shift is not None:
            shift, hp = _f32(shift)
        if res is not None:
            res, rp = _f32(res)
        self._ck(self.lib.fe_op_conv2d(self.h, xp, n, c, h, ww, wp, cout, kh, kw, sp, hp, rp, int(res_after_act),
                                       stride, pad, dil, ACT[act], y.ctypes.data_as(_f32p)))
        return y

    def topiq_gate64(self, x, w0, b0, w2, b2, w4, b4, wx, bx, wblk_act="gelu", gate_act="gelu"):
        """Test hook of the fused gate + 16x16 pool of TOPIQ's 64-channel level (2-byte precisions): x [n, 64, h, w]."""
        x, xp = _f32(x)
        n, c, h, w = x.shape
        assert c == 64
        arrs = [_f32(a) for a in (w0, b0, w2, b2, w4, wx, bx)]
        y = np.empty((n, 64, h // 16, w // 16), np.float32)
        self._ck(self.lib.fe_op_topiq_gate64(self.h, xp, n, h, w, arrs[0][1], arrs[1][1], arrs[2][1], arrs[3][1], arrs[4][1], C.c_float(float(b4)),
                                             arrs[5][1], arrs[6][1], ACT[wblk_act], ACT[gate_act], y.ctypes.data_as(_f32p)))
        return y

    def conv3x3_c64(self, x, w2, scale2=None, shift2=None, act2="relu", w3=None, scale3=None, shift3=None, res=None):
        """Test hook of the halo-tiled 3x3 (64 -> 64) kernel and its chained 1x1 expand + identity + ReLU (2-byte precisions)."""
        x, xp = _f32(x)
        n, c, h, w = x.shape
        assert c == 64
        keep = [_f32(a) if a is not None else (None, None) for a in (w2, scale2, shift2, w3, scale3, shift3, res)]
        y = np.empty((n, 256 if w3 is not None else 64, h, w), np.float32)
        self._ck(self.lib.fe_op_conv3x3_c64(self.h, xp, n, h, w, keep[0][1], keep[1][1], keep[2][1], ACT[act2], keep[3][1], keep[4][1], keep[5][1],
                                            keep[6][1], y.ctypes.data_as(_f32p)))
        return y

    def maxpool2d(self, x, k, stride, pad=0, ceil_mode=False):
        x, xp = _f32(x)
        n, c, h, w = x.shape

        def od(i):
            if ceil_mode:
                o = -(-(i + 2 * pad - k) // stride) + 1
                if (o - 1) * stride >= i + pad:
                    o -= 1
                return o
            return (i + 2 * pad - k) // stride + 1
        y = np.empty((n, c, od(h), od(w)), np.float32)
        self._ck(self.lib.fe_op_maxpool2d(self.h, xp, n, c, h, w, k, stride, pad, int(ceil_mode),
                                          y.ctypes.data_as(_f32p)))
        return y

    def bilinear(self, x, ho, wo):
        x, xp = _f32(x)
        n, c, h, w = x.shape
        y = np.empty((n, c, ho, wo), np.float32)
        self._ck(self.lib.fe_op_bilinear(self.h, xp, n, c, h, w, ho, wo, y.ctypes.data_as(_f32p)))
        return y

    def adaptive_avgpool(self, x, ho, wo):
        x, xp = _f32(x)
        n, c, h, w = x.shape
        y = np.empty((n, c, ho, wo), np.float32)
        self._ck(self.lib.fe_op_adaptive_avgpool(self.h, xp, n, c, h, w, ho, wo, y.ctypes.data_as(_f32p)))
        return y

    def layernorm(self, x, g, b, eps=1e-5):
        x, xp = _f32(x)
        g, gp = _f32(g)
        b, bp = _f32(b)
        rows, d = x.shape
        y = np.empty_like(x)
        self._ck(self.lib.fe_op_layernorm(self.h, xp, rows, d, gp, bp, eps, y.ctypes.data_as(_f32p)))
        return y

    def set_conv_variant(self, v):
        self._ck(self.lib.fe_set_conv_variant(self.h, int(v)))

    def bench_conv(self, n, h, w, cin, cout, k, stride=1, pad=0, res=False, act="relu", variant=0, iters=10):
        ms = C.c_float()
        self._ck(self.lib.fe_bench_conv(self.h, n, h, w, cin, cout, k, stride, pad, int(res), ACT[act], variant, iters,
                                        C.byref(ms)))
        return ms.value

    # -- TOPIQ ------------------------------------------------------------------------------
    @staticmethod
    def _img_ptr(images):
        """images: numpy uint8 [n,h,w,3] (host) or (device_ptr, n, h, w) tuple."""
        if isinstance(images, tuple):
            p, n, h, w = images
            return p, n, h, w, 1, None
        a = np.ascontiguousarray(images, dtype=np.uint8)
        assert a.ndim == 4 and a.shape[3] == 3
        return a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], a.shape[2], 0, a

    def topiq_configure(self, gate_act="gelu", weight_blk_act="gelu"):
        """Activations of pyiqa's GatedConv used by the NEXT load_weights(FE_MODEL_TOPIQ): 'relu' | 'gelu' | 'softplus'."""
        self._ck(self.lib.fe_topiq_configure(self.h, ACT[gate_act], ACT[weight_blk_act]))

    def topiq_f32_below(self, pixels):
        """2-byte TOPIQ: images with fewer than `pixels` pixels are scored on the model's fp32 weights (0 = never, the default)."""
        self._ck(self.lib.fe_topiq_f32_below(self.h, int(pixels)))

    def topiq_feature_shape(self, h, w, level):
        """(channels, height, width) of pyramid level `level` for h x w images, as the engine computes it (conv / pool output
        sizes, after the > 1024 LANCZOS cap)."""
        dims = (C.c_int * 3)()
        if self.lib.fe_topiq_feature_shape(int(h), int(w), int(level), dims) != 0:
            raise EngineError(f"topiq_feature_shape: bad arguments h={h} w={w} level={level}")
        return tuple(dims)

    def topiq_features(self, images, level):
        p, n, h, w, dev, keep = self._img_ptr(images)
        y = np.empty((n,) + self.topiq_feature_shape(h, w, level), np.float32)
        self._ck(self.lib.fe_topiq_features(self.h, p, n, h, w, dev, level, y.ctypes.data_as(_f32p)))
        return y

    def topiq_score(self, images):
        p, n, h, w, dev, keep = self._img_ptr(images)
        y = np.empty((n,), np.float32)
        self._ck(self.lib.fe_topiq_score(self.h, p, n, h, w, dev, y.ctypes.data_as(_f32p)))
        return y

    # -- U2-Net-P + SAMP-Net ------------------------------------------------------------------
    def u2netp_saliency(self, x):
        """x: float32 [n,3,h,w] normalised -> saliency [n,1,h,w]."""
        x, xp = _f32(x)
        n, c, h, w = x.shape
        assert c == 3
        y = np.empty((n, 1, h, w), np.float32)
        self._ck(self.lib.fe_u2netp_saliency(self.h, xp, n, h, w, y.ctypes.data_as(_f32p)))
        return y

    def samp_forward(self, x, want_saliency=False):
        """x: float32 [n,3,224,224] normalised -> (pattern_weights [n,8], attributes [n,6], score_dist [n,5][, sal])."""
        x, xp = _f32(x)
        n = x.shape[0]
        assert x.shape[1:] == (3, 224, 224)
        pw = np.empty((n, 8), np.float32)
        at = np.empty((n, 6), np.float32)
        sd = np.empty((n, 5), np.float32)
        sal = np.empty((n, 1, 224, 224), np.float32) if want_saliency else None
        self._ck(self.lib.fe_samp_forward(self.h, xp, n, pw.ctypes.data_as(_f32p), at.ctypes.data_as(_f32p),
                                          sd.ctypes.data_as(_f32p),
                                          sal.ctypes.data_as(_f32p) if want_saliency else None))
        return (pw, at, sd, sal) if want_saliency else (pw, at, sd)

    # -- CLIP -------------------------------------------------------------------------------
    def clip_encode_image(self, x, normalized=False, aesthetic=False, out_dim=768):
        """x: float32 [n,3,224,224] (host array) or (device_ptr, n) tuple. Returns features [n,768]
        (+ normalised embedding, + raw aesthetic score) like Facet.get_aesthetic_and_quality_batch needs."""
        if isinstance(x, tuple):
            p, n = x
            dev, keep = 1, None
        else:
            keep = np.ascontiguousarray(x, dtype=np.float32)
            n, dev = keep.shape[0], 0
            p = keep.ctypes.data_as(C.c_void_p)
        feat = np.empty((n, out_dim), np.float32)
        emb = np.empty((n, out_dim), np.float32) if normalized else None
        aes = np.empty((n,), np.float32) if aesthetic else None
        self._ck(self.lib.fe_clip_encode_image(self.h, p, n, dev, feat.ctypes.data_as(_f32p),
                                               emb.ctypes.data_as(_f32p) if normalized else None,
                                               aes.ctypes.data_as(_f32p) if aesthetic else None))
        out = [feat]
        if normalized:
            out.append(emb)
        if aesthetic:
            out.append(aes)
        return out[0] if len(out) == 1 else tuple(out)

    # -- preprocessing + image-level entry points ------------------------------------------
    def resize_u8(self, imgs, oh, ow, filter="bilinear"):
        """PIL-exact resize of a uint8 [n,h,w,3] batch -> [n,oh,ow,3]."""
        a = np.ascontiguousarray(imgs, dtype=np.uint8)
        n, h, w, _ = a.shape
        out = np.empty((n, oh, ow, 3), np.uint8)
        self._ck(self.lib.fe_resize_u8(self.h, a.ctypes.data_as(C.c_void_p), n, h, w, oh, ow, FILTERS[filter], 0,
                                       out.ctypes.data_as(C.c_void_p)))
        return out

    def aesthetic_score(self, feats):
        """feats float32 [n,768] -> raw aesthetic_head outputs [n] (before the (x+1)*5 clamp)."""
        a = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, 768)
        out = np.empty((a.shape[0],), np.float32)
        self._ck(self.lib.fe_aesthetic_score(self.h, a.ctypes.data_as(_f32p), a.shape[0], out.ctypes.data_as(_f32p)))
        return out

    def clip_encode_images(self, images, normalized=True, aesthetic=True):
        p, n, h, w, dev, keep = self._img_ptr(images)
        feat = np.empty((n, 768), np.float32)
        emb = np.empty((n, 768), np.float32) if normalized else None
        aes = np.empty((n,), np.float32) if aesthetic else None
        self._ck(self.lib.fe_clip_encode_images(self.h, p, n, h, w, dev, feat.ctypes.data_as(_f32p),
                                                emb.ctypes.data_as(_f32p) if normalized else None,
                                                aes.ctypes.data_as(_f32p) if aesthetic else None))
        return feat, emb, aes

    def samp_score_images(self, images, bgr=False):
        p, n, h, w, dev, keep = self._img_ptr(images)
        pw = np.empty((n, 8), np.float32)
        at = np.empty((n, 6), np.float32)
        sd = np.empty((n, 5), np.float32)
        self._ck(self.lib.fe_samp_score_images(self.h, p, n, h, w, int(bgr), dev, pw.ctypes.data_as(_f32p),
                                               at.ctypes.data_as(_f32p), sd.ctypes.data_as(_f32p)))
        return pw, at, sd

    def ensemble_score(self, images):
        """-> (records float32 [n, 789], models_run bitmask). Layout: include/facet_engine.h fe_ensemble_score."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        rec = np.empty((n, FE_RECORD_FLOATS), np.float32)
        mask = C.c_int(0)
        self._ck(self.lib.fe_ensemble_score(self.h, p, n, h, w, dev, rec.ctypes.data_as(_f32p), C.byref(mask)))
        return rec, mask.value

    def ensemble_select(self, models=7):
        """Which loaded models ensemble_score runs: 1 topiq | 2 clip (+ aesthetic) | 4 samp."""
        self._ck(self.lib.fe_ensemble_select(self.h, int(models)))
        self.ensemble_mask = int(models)

    def ensemble_score_dev(self, images, d_records, ld_records=FE_RECORD_FLOATS):
        """ensemble_score with the [n, ld_records] float32 records left in device memory at `d_records` (an int address or
        c_void_p, e.g. torch_tensor.data_ptr()); returns the models_run bitmask after the engine stream has drained."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        mask = C.c_int(0)
        self._ck(self.lib.fe_ensemble_score_dev(self.h, p, n, h, w, dev, C.c_void_p(int(d_records) if not isinstance(d_records, C.c_void_p)
                                                                                     else d_records.value),
                                                int(ld_records), C.byref(mask)))
        return mask.value

    # -- VLM tagger text decoder (models/vlm_tagger.py) ---------------------------------------------------
    def vlm_configure(self, n_heads=28, n_kv_heads=4, head_dim=128, rope_theta=1e6, rms_eps=1e-6, mrope_section=(16, 24, 24)):
        """Geometry read by the NEXT load_weights(FE_MODEL_VLM, ...) (transformers Qwen2_5_VLTextConfig; defaults = Qwen2.5-VL-7B)."""
        ms = (C.c_int * 3)(*[int(v) for v in mrope_section])
        self._ck(self.lib.fe_vlm_configure(self.h, int(n_heads), int(n_kv_heads), int(head_dim), float(rope_theta), float(rms_eps), ms))

    def vlm_vision_configure(self, n_heads=16, fullatt_block_indexes=(7, 15, 23, 31)):
        """Vision-tower geometry read by the NEXT load_weights(FE_MODEL_VLM, ...) (Qwen2_5_VLVisionConfig; defaults = Qwen2.5-VL-7B)."""
        fa = (C.c_int * max(1, len(fullatt_block_indexes)))(*[int(v) for v in fullatt_block_indexes])
        self._ck(self.lib.fe_vlm_vision_configure(self.h, int(n_heads), fa, len(fullatt_block_indexes)))

    def vlm_encode_images(self, pixel_values, patch_pos_hw, window_index, cu_window_seqlens, cu_seqlens, want_embeds=True):
        """model.visual(pixel_values, grid_thw).pooler_output: float32 [n_patches / 4, hidden] (bf16 values), raster order; the embeddings
        also stay on the device for the next vlm_prefill(..., image_rows=...). Index arrays: facet_amd.vlm_tagger.vision_indices."""
        pv = np.ascontiguousarray(pixel_values, dtype=np.float32)
        n = pv.shape[0]
        pos, pp = self._i32(patch_pos_hw)
        wi, wp = self._i32(window_index)
        cw, cwp = self._i32(cu_window_seqlens)
        cf, cfp = self._i32(cu_seqlens)
        assert pos.shape == (n, 2) and wi.shape == (n // 4,), (pos.shape, wi.shape)
        out = np.empty((n // 4, self.vlm_dims()["hidden"]), np.float32) if want_embeds else None
        self._ck(self.lib.fe_vlm_encode_images(self.h, pv.ctypes.data_as(_f32p), n, pp, wp, cwp, len(cw) - 1, cfp, len(cf) - 1,
                                               out.ctypes.data_as(_f32p) if want_embeds else None))
        return out

    def vlm_dims(self):
        d = (C.c_int * 8)()
        self._ck(self.lib.fe_vlm_dims(self.h, d))
        return dict(zip(("vocab", "hidden", "layers", "heads", "kv_heads", "intermediate", "max_seq", "cur_len"), list(d)))

    @staticmethod
    def _i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return a, a.ctypes.data_as(C.POINTER(C.c_int32))

    def vlm_prefill(self, tokens, position_ids=None, max_seq=None, want_logits=False, image_rows=None):
        """tokens int [n_seq, len]; position_ids int [3, n_seq, len] (None: text-only positions 0..len-1 on all three axes).
        image_rows: flat row indices (sequence * len + position) of the <|image_pad|> tokens, in order - they take the embeddings of
        the last vlm_encode_images.
        -> next token ids [n_seq] (greedy) and, with want_logits, the bf16 logits widened to float32 [n_seq, vocab]."""
        tok, tp = self._i32(tokens)
        n, L = tok.shape
        if position_ids is None:
            position_ids = np.broadcast_to(np.arange(L, dtype=np.int32), (3, n, L))
        pos, pp = self._i32(position_ids)
        assert pos.shape == (3, n, L), pos.shape
        nxt = np.empty(n, np.int32)
        lg = np.empty((n, self.vlm_dims()["vocab"]), np.float32) if want_logits else None
        if image_rows is not None:
            ir, irp = self._i32(image_rows)
            self._ck(self.lib.fe_vlm_prefill_images(self.h, tp, pp, n, L, int(max_seq or min(8192, L + 256)), irp, int(ir.size),
                                                    nxt.ctypes.data_as(C.POINTER(C.c_int32)), lg.ctypes.data_as(_f32p) if want_logits else None))
        else:
            self._ck(self.lib.fe_vlm_prefill(self.h, tp, pp, n, L, int(max_seq or min(8192, L + 256)), nxt.ctypes.data_as(C.POINTER(C.c_int32)),
                                             lg.ctypes.data_as(_f32p) if want_logits else None))
        return (nxt, lg) if want_logits else nxt

    def vlm_decode_step(self, tokens, position_ids, want_logits=False):
        """tokens int [n_seq] (the tokens chosen at the previous step), position_ids int [3, n_seq]."""
        tok, tp = self._i32(tokens)
        pos, pp = self._i32(position_ids)
        n = tok.shape[0]
        assert pos.shape == (3, n), pos.shape
        nxt = np.empty(n, np.int32)
        lg = np.empty((n, self.vlm_dims()["vocab"]), np.float32) if want_logits else None
        self._ck(self.lib.fe_vlm_decode_step(self.h, tp, pp, n, nxt.ctypes.data_as(C.POINTER(C.c_int32)), lg.ctypes.data_as(_f32p) if want_logits else None))
        return (nxt, lg) if want_logits else nxt

    def vlm_generate(self, tokens, max_new_tokens, position_ids=None, eos_token_ids=(), want_logits=False, forced_tokens=None, image_rows=None):
        """Greedy generation (`generate(..., do_sample=False)`, models/vlm_tagger.py:255-259): prefill + max_new_tokens - 1 decode
        steps for all sequences in lockstep; a sequence that emitted an EOS id keeps receiving that id (what generate's padding does).
        New positions continue from max(position_ids) + 1 per sequence. forced_tokens [n_seq, max_new_tokens]: teacher forcing - the
        token FED at each step is taken from there instead of the engine's own choice (parity tests)."""
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        n, L = tok.shape
        if position_ids is None:
            position_ids = np.broadcast_to(np.arange(L, dtype=np.int32), (3, n, L))
        position_ids = np.ascontiguousarray(position_ids, dtype=np.int32)
        nxt_pos = position_ids.max(axis=(0, 2)) + 1            # [n_seq]
        if not want_logits and forced_tokens is None:
            # the product path: prefill, then every decode step on the device (fe_vlm_generate: captured graph, no host round trips)
            first = self.vlm_prefill(tok, position_ids, max_seq=min(8192, L + max_new_tokens), image_rows=image_rows)
            out = np.empty((n, max_new_tokens), np.int32)
            out[:, 0] = first
            if max_new_tokens > 1:
                ft, fp = self._i32(first)
                ps, pp = self._i32(np.broadcast_to(nxt_pos.astype(np.int32), (3, n)))
                steps = np.empty((max_new_tokens - 1, n), np.int32)
                self._ck(self.lib.fe_vlm_generate(self.h, fp, pp, n, max_new_tokens - 1, steps.ctypes.data_as(C.POINTER(C.c_int32))))
                out[:, 1:] = steps.T
            eos = [int(e) for e in eos_token_ids]
            for b in range(n if eos else 0):      # generate() pads a finished sequence with its EOS id
                hit = np.flatnonzero(np.isin(out[b], eos))
                if hit.size:
                    out[b, hit[0]:] = out[b, hit[0]]
            return out
        out = np.zeros((n, max_new_tokens), np.int32)
        logits = []
        r = self.vlm_prefill(tok, position_ids, max_seq=min(8192, L + max_new_tokens), want_logits=want_logits, image_rows=image_rows)
        cur = r[0] if want_logits else r
        done = np.zeros(n, bool)
        eos = set(int(e) for e in eos_token_ids)
        for step in range(max_new_tokens):
            if want_logits:
                logits.append(r[1])
            out[:, step] = cur
            done |= np.isin(cur, list(eos)) if eos else False
            if step + 1 == max_new_tokens or done.all():
                out[:, step + 1:] = cur[:, None] if done.all() else 0
                break
            feed = cur if forced_tokens is None else np.asarray(forced_tokens)[:, step].astype(np.int32)
            r = self.vlm_decode_step(feed, np.broadcast_to(nxt_pos.astype(np.int32), (3, n)), want_logits=want_logits)
            nxt_pos = nxt_pos + 1
            new = r[0] if want_logits else r
            cur = np.where(done, cur, new)
        return (out, np.stack(logits, 1)) if want_logits else out

    # -- ONNX graphs (InsightFace sessions) -------------------------------------------------------------
    def graph_load(self, slot, onnx_bytes):
        buf = C.create_string_buffer(bytes(onnx_bytes), len(onnx_bytes))
        self._ck(self.lib.fe_graph_load(self.h, int(slot), buf, len(onnx_bytes)))

    def graph_unload(self, slot):
        self._ck(self.lib.fe_graph_unload(self.h, int(slot)))

    def graph_loaded(self, slot):
        return bool(self.lib.fe_graph_loaded(self.h, int(slot)))

    def graph_info(self, slot):
        nn, no, fl = C.c_int(), C.c_int(), C.c_int()
        dims = (C.c_int64 * 4)()
        self._ck(self.lib.fe_graph_info(self.h, int(slot), C.byref(nn), C.byref(no), dims, C.byref(fl)))
        return {"nodes": nn.value, "outputs": no.value, "input_dims": list(dims), "has_sub": bool(fl.value & 1),
                "has_mul": bool(fl.value & 2)}

    def graph_run(self, slot, x):
        """x: float32 [n,c,h,w] -> list of numpy outputs in the model's declared order (ONNX layouts)."""
        x, xp = _f32(x)
        n, c, h, w = x.shape
        self._ck(self.lib.fe_graph_run(self.h, int(slot), xp, n, c, h, w, 0))
        outs = []
        for i in range(self.graph_info(slot)["outputs"]):
            dims = (C.c_int64 * 6)()
            rank = C.c_int()
            name = C.create_string_buffer(256)
            self._ck(self.lib.fe_graph_output_info(self.h, int(slot), i, name, 256, dims, C.byref(rank)))
            shape = tuple(dims[k] for k in range(rank.value))
            y = np.empty(shape, np.float32)
            self._ck(self.lib.fe_graph_output_copy(self.h, int(slot), i, y.ctypes.data_as(_f32p), y.size))
            outs.append(y)
        return outs

    # -- face path ------------------------------------------------------------------------------------------
    def face_detect(self, images, det_size=(640, 640), thresh=0.5, max_cand=512):
        """images: BGR uint8 [n,h,w,3] or (device_ptr,n,h,w). -> (cand [n,max_cand,16], counts [n], det_scale)."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        cand = np.zeros((n, max_cand, 16), np.float32)
        counts = np.zeros((n,), np.int32)
        ds = C.c_float()
        self._ck(self.lib.fe_face_detect(self.h, p, n, h, w, dev, int(det_size[0]), int(det_size[1]), float(thresh), int(max_cand),
                                         cand.ctypes.data_as(_f32p), counts.ctypes.data_as(C.POINTER(C.c_int)), C.byref(ds)))
        return cand, counts, ds.value

    def face_crops_run(self, slot, images, img_index, M, size, mean, scale, swap_rb=True, out_dim=0, want_crops=False):
        """M: float64 [m,2,3] forward affine matrices. -> (out [m,out_dim] or None, crops uint8 [m,size,size,3] or None)."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        idx = np.ascontiguousarray(img_index, dtype=np.int32)
        Mm = np.ascontiguousarray(M, dtype=np.float64).reshape(-1, 6)
        m = idx.shape[0]
        assert Mm.shape[0] == m
        out = np.empty((m, out_dim), np.float32) if out_dim else None
        crops = np.empty((m, size, size, 3), np.uint8) if want_crops else None
        self._ck(self.lib.fe_face_crops_run(self.h, int(slot), p, n, h, w, dev, m, idx.ctypes.data_as(C.POINTER(C.c_int)),
                                            Mm.ctypes.data_as(C.POINTER(C.c_double)), int(size), float(mean), float(scale), int(swap_rb),
                                            out.ctypes.data_as(_f32p) if out_dim else None, int(out_dim),
                                            crops.ctypes.data_as(C.c_void_p) if want_crops else None))
        return out, crops

    def face_analyze(self, images, det_size=(640, 640), det_thresh=0.5, nms_thresh=0.4, max_faces=8):
        """-> (faces float32 [n,max_faces,739], counts int32 [n], models_run bitmask); layout: include/facet_engine.h."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        faces = np.empty((n, max_faces, FE_FACE_FLOATS), np.float32)
        counts = np.zeros((n,), np.int32)
        mask = C.c_int(0)
        self._ck(self.lib.fe_face_analyze(self.h, p, n, h, w, dev, int(det_size[0]), int(det_size[1]), float(det_thresh),
                                          float(nms_thresh), int(max_faces), faces.ctypes.data_as(_f32p),
                                          counts.ctypes.data_as(C.POINTER(C.c_int)), C.byref(mask)))
        return faces, counts, mask.value

    def image_stats(self, images, want_gray=False, want_hsv=False):
        """BGR uint8 [n,h,w,3] (or device tuple) -> (stats float64 [n,264], gray uint8 [n,h,w] | None, hsv uint8 [n,h,w,3] | None)."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        stats = np.empty((n, FE_STATS_DOUBLES), np.float64)
        gray = np.empty((n, h, w), np.uint8) if want_gray else None
        hsv = np.empty((n, h, w, 3), np.uint8) if want_hsv else None
        self._ck(self.lib.fe_image_stats(self.h, p, n, h, w, dev, stats.ctypes.data_as(C.POINTER(C.c_double)),
                                         gray.ctypes.data_as(C.c_void_p) if want_gray else None,
                                         hsv.ctypes.data_as(C.c_void_p) if want_hsv else None))
        return stats, gray, hsv

    def roi_laplacian(self, images, img_index, rois):
        """rois int [m,4] (x1,y1,x2,y2 exclusive, clipped) -> float64 [m,4]: sum lap, sum lap^2, sum gray, pixel count."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        idx = np.ascontiguousarray(img_index, dtype=np.int32)
        r = np.ascontiguousarray(rois, dtype=np.int32).reshape(-1, 4)
        out = np.zeros((idx.shape[0], 4), np.float64)
        self._ck(self.lib.fe_roi_laplacian(self.h, p, n, h, w, dev, idx.shape[0], idx.ctypes.data_as(C.POINTER(C.c_int)),
                                           r.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def swap_rb(self, src, pixels, dst_device):
        """dst_device (device pointer) <- src (host uint8 array or device pointer) with R and B exchanged, `pixels` 3-byte pixels."""
        if isinstance(src, np.ndarray):
            a = np.ascontiguousarray(src, dtype=np.uint8)
            assert a.size == pixels * 3
            self._ck(self.lib.fe_swap_rb_u8(self.h, a.ctypes.data_as(C.c_void_p), 0, pixels, dst_device))
        else:
            self._ck(self.lib.fe_swap_rb_u8(self.h, src, 1, pixels, dst_device))

    def leading_lines(self, images, canny_low=50, canny_high=150, threshold=80, min_line_length=None, max_line_gap=20, max_lines=2048,
                      want_edges=False):
        """BGR uint8 batch -> list of int32 [k,4] segment arrays (x1,y1,x2,y2; what cv2.HoughLinesP(cv2.Canny(cv2.GaussianBlur(gray,
        (5,5), 0), low, high), 1, pi/180, threshold, minLineLength, maxLineGap) returns per image, in the order found), and the
        Canny edge images uint8 [n,h,w] when want_edges. min_line_length defaults to int(min(h, w) * 0.15) (composition.py:219)."""
        p, n, h, w, dev, keep = self._img_ptr(images)
        if min_line_length is None:
            min_line_length = int(min(h, w) * 0.15)
        edges = np.empty((n, h, w), np.uint8) if want_edges else None
        while True:
            lines = np.zeros((n, max_lines, 4), np.int32)
            counts = np.zeros(n, np.int32)
            self._ck(self.lib.fe_leading_lines(self.h, p, n, h, w, dev, canny_low, canny_high, threshold, min_line_length, max_line_gap, max_lines,
                                               lines.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                               edges.ctypes.data_as(C.c_void_p) if want_edges else None))
            if int(counts.max()) <= max_lines:
                break
            max_lines = int(counts.max())          # rare: more segments than room - run again with enough
        out = [lines[i, :counts[i]].copy() for i in range(n)]
        return (out, edges) if want_edges else out

    def cv_resize_linear(self, imgs, oh, ow):
        a = np.ascontiguousarray(imgs, dtype=np.uint8)
        n, h, w, _ = a.shape
        out = np.empty((n, oh, ow, 3), np.uint8)
        self._ck(self.lib.fe_cv_resize_linear_u8(self.h, a.ctypes.data_as(C.c_void_p), n, h, w, oh, ow, out.ctypes.data_as(C.c_void_p)))
        return out

    def tag_similarities(self, emb, text):
        """emb [n,d], text [T,d] (L2-normalised rows) -> cosine similarities [n,T] computed on the GPU."""
        emb, ep = _f32(emb)
        text, tp = _f32(text)
        n, d = emb.shape
        T = text.shape[0]
        out = np.empty((n, T), np.float32)
        self._ck(self.lib.fe_tag_similarities(self.h, ep, n, tp, T, d, out.ctypes.data_as(_f32p)))
        return out

    def clip_encode_text(self, tokens, out_dim=768):
        """tokens: int [n, 77] -> un-normalised text features [n, 768]."""
        tk = np.ascontiguousarray(tokens, dtype=np.int32)
        n, L = tk.shape
        out = np.empty((n, out_dim), np.float32)
        self._ck(self.lib.fe_clip_encode_text(self.h, tk.ctypes.data_as(C.POINTER(C.c_int32)), n, L, out.ctypes.data_as(_f32p)))
        return out

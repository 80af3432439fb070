"""Probe: can torch.cuda and the engine share one process in either initialisation order? usage: hip_coexist_probe.py engine_first|preload|torch_first"""
import ctypes, importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "preload":
    d = os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib")
    ctypes.CDLL(os.path.join(d, "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
if mode == "torch_first":
    import torch
    print("torch sees", torch.cuda.is_available(), torch.cuda.device_count())
from facet_amd import Engine
e = Engine(0, arena_bytes=1 << 30)
print("engine ok", e is not None)
import torch
print(mode, "-> torch.cuda.is_available():", torch.cuda.is_available())
if torch.cuda.is_available():
    x = torch.ones(4, device="cuda"); print("torch sum", float(x.sum()))
import numpy as np
from facet_amd.weights import synthetic_images
print("engine stats ok", e.image_stats(synthetic_images(1, 1, 32, 32))[0].shape)

"""CLIP ViT-L/14 tower rate per precision at bench-sized batches (256 crops resident, tower chunks as the engine picks them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ["FACET_AMD_SYNTHETIC"] = "1"
import numpy as np
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_CLIP, FE_MODEL_AESTHETIC
from facet_amd.weights import synthetic_state_dict
x = np.random.default_rng(0).normal(0, 1, (256, 3, 224, 224)).astype(np.float32)
sd, sa = synthetic_state_dict("clip", 3), synthetic_state_dict("aesthetic", 3)
for prec in sys.argv[1:] or ["f32", "f16", "f16+r32", "f16x3"]:
    e = Engine(0, arena_bytes=24 << 30, precision=prec)
    e.load_weights(FE_MODEL_CLIP, sd); e.load_weights(FE_MODEL_AESTHETIC, sa)
    e.clip_encode_image(x, normalized=True, aesthetic=True)
    e.flops_reset(); e.timer_start(); e.clip_encode_image(x, normalized=True, aesthetic=True); ms = e.timer_stop()
    print(f"{prec:8s} {256 / ms * 1e3:8.1f} crops/s  {ms / 256:.3f} ms/crop  executed {e.flops_executed() / ms / 1e9:7.1f} TFLOP/s  (algorithmic {e.flops() / ms / 1e9:6.1f})", flush=True)
    e.close()

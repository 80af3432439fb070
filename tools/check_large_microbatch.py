"""bf16 TOPIQ at micro-batch 128 (512x512x64 maps of 4.3 GB: issued per image group) against micro-batch 32 - needs a 200 GB arena, so it is a tool, not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ["FACET_AMD_SYNTHETIC"] = "1"
import numpy as np
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_TOPIQ
from facet_amd.weights import synthetic_state_dict, synthetic_images
e = Engine(0, arena_bytes=200 << 30, precision="bf16")
e.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 5))
imgs = synthetic_images(21, 130, 1024, 1024)
d = e.dev_alloc(imgs.nbytes); e.h2d(d, imgs)
e.set_microbatch(32); a = e.topiq_score((d, 130, 1024, 1024))
e.set_microbatch(128); b = e.topiq_score((d, 130, 1024, 1024))
print("max rel diff mb128 vs mb32 (bf16):", float(np.abs(a - b).max() / np.abs(a).max()), a[:3], b[:3])
e.close()

# fp32 at micro-batch 64 (66 images: the last group is ragged): the timing that used to sit in tests/test_fullsize_gpu.py as an assert.
# Through the register-staged fallback those layers once took, this forward ran at 340 images/s; on the split launches ~400.
e = Engine(0, arena_bytes=136 << 30)
e.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 5))
imgs = synthetic_images(21, 66, 1024, 1024)
d = e.dev_alloc(imgs.nbytes); e.h2d(d, imgs)
e.set_microbatch(64)
e.topiq_score((d, 66, 1024, 1024))
e.timer_start(); e.topiq_score((d, 66, 1024, 1024)); ms = e.timer_stop()
print(f"[fp32 micro-batch 64] 66 images in {ms:.0f} ms = {66 / ms * 1e3:.0f} images/s")
e.close()

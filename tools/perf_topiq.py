"""Per-kernel timing of the TOPIQ forward on synthetic 1024x1024 batches (developer tool)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_TOPIQ
from facet_amd.weights import synthetic_state_dict, synthetic_images

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
level_only = len(sys.argv) > 4 and sys.argv[4] == "levels"
prec = "bf16" if "bf16" in sys.argv else ("f16" if "f16" in sys.argv else "f32")
eng = Engine(0, arena_bytes=(4 + 2 * mb) << 30, precision=prec)
eng.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 3))
imgs = synthetic_images(2, n, hw, hw)
d = eng.dev_alloc(imgs.nbytes); eng.h2d(d, imgs)
eng.set_microbatch(mb)
run = (lambda: eng.topiq_features((d, n, hw, hw), 4)) if level_only else (lambda: eng.topiq_score((d, n, hw, hw)))
run()
eng.flops_reset()
eng.timer_start(); run(); ms = eng.timer_stop()
fl = eng.flops()
print(f"n={n} mb={mb} {hw}x{hw}: {ms:.1f} ms  {n/ms*1e3:.1f} img/s  {fl/ms/1e9:.1f} TFLOP/s  ({fl/n/1e9:.1f} GFLOP/img)")
eng.set_microbatch(mb); eng.profile_enable(True)
eng.topiq_features((d, mb, hw, hw), 4) if level_only else eng.topiq_score((d, mb, hw, hw))
recs = eng.profile_records(); eng.profile_enable(False)
agg = {}
for r in recs:
    a = agg.setdefault(r["name"], [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]; a[3] += r["bytes"]
tot = sum(a[1] for a in agg.values())
print(f"profiled conv time per microbatch: {tot:.2f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1]:8.3f} ms {100*a[1]/tot:5.1f}%  x{a[0]:<3d} {a[2]/a[1]/1e9:7.1f} TF/s {a[3]/a[1]/1e6:7.0f} GB/s  {k}")

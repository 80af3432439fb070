import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_CLIP, FE_MODEL_AESTHETIC
from facet_amd.weights import synthetic_state_dict
n, mb = 127, 32
prec = next((a for a in sys.argv[1:] if a in ('f32', 'bf16', 'f16', 'f16+r32', 'bf16+r32', 'f16x3')), 'f32')
eng = Engine(0, arena_bytes=40 << 30, precision=prec)
eng.set_microbatch(mb)
x = np.random.default_rng(0).standard_normal((n, 3, 224, 224), dtype=np.float32)
eng.load_weights(FE_MODEL_CLIP, synthetic_state_dict("clip", 9))
d = eng.dev_alloc(x.nbytes); eng.h2d(d, x)
eng.clip_encode_image((d, n))
eng.timer_start(); eng.clip_encode_image((d, n)); ms = eng.timer_stop()
print(f'{prec}: {n} images in {ms:.2f} ms = {n / ms * 1e3:.1f} img/s')
eng.profile_enable(True)
eng.clip_encode_image((d, n))
recs = eng.profile_records(); eng.profile_enable(False)
agg = {}
for r in recs:
    a = agg.setdefault(r["name"], [0, 0.0, 0.0]); a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
tot = sum(a[1] for a in agg.values())
print(f"profiled GEMM time per {n} images: {tot:.2f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1]:8.3f} ms {100*a[1]/tot:5.1f}%  x{a[0]:<3d} {a[2]/a[1]/1e9:7.1f} TF/s  {k}")

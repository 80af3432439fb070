import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.weights import synthetic_state_dict
n = mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = Engine(0, arena_bytes=60 << 30)
eng.set_microbatch(mb)
x = np.random.default_rng(0).standard_normal((n, 3, 224, 224), dtype=np.float32)
eng.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", 7))
eng.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", 7))
eng.samp_forward(x)
eng.profile_enable(True)
eng.samp_forward(x)
recs = eng.profile_records(); eng.profile_enable(False)
agg = {}
for r in recs:
    a = agg.setdefault(r["name"], [0, 0.0, 0.0]); a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
tot = sum(a[1] for a in agg.values())
print(f"profiled conv time per {n} images: {tot:.2f} ms, {len(recs)} launches")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{a[1]:8.3f} ms {100*a[1]/tot:5.1f}%  x{a[0]:<3d} {a[2]/a[1]/1e9:7.1f} TF/s  {k}")

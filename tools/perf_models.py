"""Throughput of CLIP and SAMP paths on synthetic inputs (developer tool)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.weights import synthetic_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = Engine(0, arena_bytes=24 << 30)
eng.set_microbatch(mb)
x = np.random.default_rng(0).standard_normal((n, 3, 224, 224), dtype=np.float32)
eng.load_weights(FE_MODEL_CLIP, synthetic_state_dict("clip", 9))
eng.load_weights(FE_MODEL_AESTHETIC, synthetic_state_dict("aesthetic", 9))
d = eng.dev_alloc(x.nbytes); eng.h2d(d, x)
eng.clip_encode_image((d, n), normalized=True, aesthetic=True)
eng.flops_reset(); eng.timer_start(); eng.clip_encode_image((d, n), normalized=True, aesthetic=True); ms = eng.timer_stop()
print(f"CLIP n={n} mb={mb}: {ms:.1f} ms {n/ms*1e3:.1f} img/s {eng.flops()/ms/1e9:.1f} TF/s ({eng.flops()/n/1e9:.1f} GFLOP/img)")
eng.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", 7))
eng.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", 7))
eng.samp_forward(x)
eng.flops_reset(); t=time.perf_counter(); eng.samp_forward(x); dt=time.perf_counter()-t
print(f"SAMP+U2NETP n={n} mb={mb}: {dt*1e3:.1f} ms (incl H2D) {n/dt:.1f} img/s {eng.flops()/dt/1e12:.1f} TF/s ({eng.flops()/n/1e9:.1f} GFLOP/img)")

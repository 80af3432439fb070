"""Throughput of the VLM tagger's text decoder at Qwen2.5-VL-7B's geometry (BASELINE configs[4] shape; slice 1 = decoder only).

  python tools/perf_vlm.py [--layers 28] [--prompt 512] [--new 32] [--batches 1,8,32]

Seeded synthetic weights in the 7B geometry (hidden 3584, 28 q heads over 4 KV heads of 128, intermediate 18944, vocab 152064); with
--layers < 28 the per-layer work is measured on that many layers and the lm_head once, and both the measured and the 28-layer
extrapolation are printed. Prefill is matrix-core bound (2 * parameters * tokens FLOPs); decode is HBM bound (every weight byte once per
step, whatever the batch) - the decode line reports GB/s of weight traffic against the 8 TB/s peak.
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facet_amd import Engine
from facet_amd._lib import FE_MODEL_VLM
from facet_amd.weights import synthetic_state_dict, qwen2_5_vl_text_spec

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--prompt", type=int, default=512)
ap.add_argument("--new", type=int, default=16)
ap.add_argument("--batches", default="1,8,32")
ap.add_argument("--vision", type=int, default=0, help="also time the vision tower: this many images of 1036x1036 pixels (74x74 patches) per call, full depth 32")
a = ap.parse_args()
H, NH, NKV, INTER, V = 3584, 28, 4, 18944, 152064
t0 = time.time()
from facet_amd.weights import qwen2_5_vl_vision_spec
spec = qwen2_5_vl_text_spec(hidden=H, layers=a.layers, heads=NH, kv_heads=NKV, inter=INTER, vocab=V)
if a.vision:
    spec = spec + qwen2_5_vl_vision_spec()      # 32 blocks of 1280 (16 heads of 80), intermediate 3420, merger to 3584
sd = synthetic_state_dict(None, 3, spec=spec)
print(f"weights drawn in {time.time() - t0:.0f} s", flush=True)
e = Engine(0, arena_bytes=40 << 30)
e.vlm_configure(NH, NKV, 128, 1e6, 1e-6, (16, 24, 24))
t0 = time.time()
e.load_weights(FE_MODEL_VLM, sd)
del sd
print(f"committed in {time.time() - t0:.0f} s", flush=True)
if a.vision:
    from facet_amd.vlm_tagger import vision_indices
    g = [[1, 74, 74]] * a.vision
    idx = vision_indices(g)
    n = 74 * 74 * a.vision
    pv = np.random.default_rng(0).normal(0, 1, (n, 1176)).astype(np.float32)
    e.vlm_encode_images(pv, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"], want_embeds=False)
    e.flops_reset(); e.timer_start()
    e.vlm_encode_images(pv, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"], want_embeds=False)
    ms = e.timer_stop()
    print(f"vision tower: {a.vision} images of 74x74 patches ({n // 4 // a.vision} image tokens each) in {ms:.2f} ms = {a.vision / ms * 1e3:.1f} images/s, "
          f"{e.flops() / ms / 1e9:.1f} TFLOP/s (projections; host patches uploaded inside the call)", flush=True)
layer_params = H * (NH + 2 * NKV) * 128 + NH * 128 * H + 3 * H * INTER
head_params = V * H
for B in [int(b) for b in a.batches.split(",")]:
    L = a.prompt
    p = np.random.default_rng(B).integers(0, V, (B, L)).astype(np.int32)
    e.vlm_prefill(p, max_seq=L + a.new + 8)
    e.timer_start(); nxt = e.vlm_prefill(p, max_seq=L + a.new + 8); t_pre = e.timer_stop()
    pos = np.full((3, B), L, np.int32)
    e.vlm_decode_step(nxt, pos)
    t0 = time.perf_counter()
    for s in range(a.new):
        nxt = e.vlm_decode_step(nxt, pos + 1 + s)
    t_dec = (time.perf_counter() - t0) / a.new * 1e3
    # the product path: all decode steps on the device (fe_vlm_generate): one captured graph replayed for <= 2 sequences, stream launches above
    e.vlm_prefill(p, max_seq=L + a.new + 8)
    t0 = time.perf_counter(); e.vlm_generate(p, a.new + 1); t_all = (time.perf_counter() - t0) * 1e3
    t_graph = (t_all - t_pre) / a.new
    fl_pre = 2.0 * (a.layers * layer_params) * B * L + 2.0 * head_params * B + 4.0 * B * NH * L * (L + 1) / 2 * 128 * a.layers
    wbytes = 2.0 * (a.layers * layer_params + head_params)
    full_pre = t_pre * 28 / a.layers
    full_dec = (t_dec - 0) * (28 * layer_params + head_params) / (a.layers * layer_params + head_params)
    print(f"B={B:3d} L={L}: prefill {t_pre:8.2f} ms = {B * L / t_pre * 1e3:9.0f} tok/s, {fl_pre / t_pre / 1e9:7.1f} TFLOP/s ({a.layers} layers; x28/{a.layers}: {full_pre:.1f} ms) | "
          f"decode {t_dec:6.3f} ms/step = {B / t_dec * 1e3:7.0f} tok/s, weights {wbytes / t_dec / 1e6:6.0f} GB/s = {wbytes / t_dec / 1e6 / 8000:.2f} of HBM peak "
          f"(28 layers: ~{full_dec:.2f} ms/step) | device-resident loop (fe_vlm_generate) {t_graph:6.3f} ms/step = {wbytes / t_graph / 1e6:6.0f} GB/s = {wbytes / t_graph / 1e6 / 8000:.2f} of peak", flush=True)
e.close()

"""GPU: the one-launch form of TOPIQ's GatedConv + 16x16 average pool on the 64-channel pyramid level (facet_amd/csrc/kernels_gate.hip)
against torch's fp32 arithmetic of the same layer chain (pyiqa CFANet GatedConv as models/pyiqa_scorer.py runs it; oracle/topiq.py), and
the whole TOPIQ model with and without it.

Tolerances: the fused kernel keeps wa / wb / the gate in the model's 2-byte type exactly where the four-launch form stores them (so
do the torch references below, which round at the same points) and uses the tanh form of GELU (<= 4.8e-4 from erf): pooled outputs
within 2e-3 * max|ref| for fp16, 1.2e-2 for bf16. With ReLU and small-integer weights every intermediate up to the gate's logit is
an exact integer, so an indexing mistake in the halo / tap / fragment orders shows as a wrong value far outside the gate's rounding.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["f16", "bf16"])
def eng2(request):
    from facet_amd import Engine
    e = Engine(0, arena_bytes=8 << 30, precision=request.param)
    e._prec_name = request.param
    yield e
    e.close()


def _round(t, prec):
    return t.half().float() if prec == "f16" else t.bfloat16().float()


def _ref(x, w0, b0, w2, b2, w4, b4, wx, bx, act_w, act_g, prec):
    rd = lambda t: _round(t, prec)
    x = rd(torch.from_numpy(x))
    w0, w2, w4, wx = (rd(torch.from_numpy(a)) for a in (w0, w2, w4, wx))
    b0, b2, bx = (torch.from_numpy(a) for a in (b0, b2, bx))
    aw = {"gelu": F.gelu, "relu": F.relu, "softplus": F.softplus}[act_w]
    ag = {"gelu": F.gelu, "relu": F.relu, "softplus": F.softplus}[act_g]
    wa = rd(aw(F.conv2d(x, w0.view(64, 64, 1, 1), b0)))
    wb = rd(aw(F.conv2d(wa, w2, b2, padding=1)))
    wc = rd(torch.sigmoid(F.conv2d(wb, w4, torch.tensor([b4], dtype=torch.float32), padding=1)))
    gated = ag(F.conv2d(x, wx.view(64, 64, 1, 1), bx)) * wc
    return F.avg_pool2d(gated, 16).numpy()


def _weights(rng, scale=1.0):
    w0 = rng.normal(0, scale / 8, (64, 64)).astype(np.float32)
    w2 = rng.normal(0, scale / 24, (64, 64, 3, 3)).astype(np.float32)
    w4 = rng.normal(0, scale / 12, (1, 64, 3, 3)).astype(np.float32)
    wx = rng.normal(0, scale / 8, (64, 64)).astype(np.float32)
    b0, b2, bx = (rng.normal(0, 0.2, 64).astype(np.float32) for _ in range(3))
    return w0, b0, w2, b2, w4, float(rng.normal(0, 0.2)), wx, bx


@pytest.mark.parametrize("shape", [(1, 16, 16), (2, 32, 48), (1, 64, 16), (3, 48, 80)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("acts", [("gelu", "gelu"), ("relu", "softplus")], ids=lambda a: "+".join(a))
def test_fused_gate_matches_torch(eng2, shape, acts):
    n, h, w = shape
    prec = eng2._prec_name
    rng = np.random.default_rng(n * 1000 + h * 10 + w)
    x = rng.normal(0, 1, (n, 64, h, w)).astype(np.float32)
    ws = _weights(rng)
    ref = _ref(x, *ws, acts[0], acts[1], prec)
    got = eng2.topiq_gate64(x, *ws, wblk_act=acts[0], gate_act=acts[1])
    assert got.shape == ref.shape
    tol = (2e-3 if prec == "f16" else 1.2e-2) * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol, (float(np.abs(got - ref).max()), float(np.abs(ref).max()))


def test_fused_gate_integer_data_catches_index_errors(eng2):
    """ReLU + weights in {-1, 0, 1} (sparse) + small-integer inputs: wa, wb and the gate's logit are exact integers in either 2-byte
    type; every channel / tap / halo pixel carries a distinct pattern, so a permuted fragment or a shifted tap changes logits by whole
    units. Window borders, image borders (zero padding of BOTH 3x3 convolutions) and interior windows are all present at 48 x 64."""
    prec = eng2._prec_name
    rng = np.random.default_rng(11)
    n, h, w = 2, 48, 64
    x = rng.integers(-2, 3, (n, 64, h, w)).astype(np.float32)
    sparse = lambda shape, p: (rng.integers(-1, 2, shape) * (rng.random(shape) < p)).astype(np.float32)
    w0 = sparse((64, 64), 0.12)
    w2 = sparse((64, 64, 3, 3), 0.02)
    w4 = sparse((1, 64, 3, 3), 0.05)
    wx = sparse((64, 64), 0.12)
    b0 = rng.integers(-1, 2, 64).astype(np.float32)
    b2 = rng.integers(-1, 2, 64).astype(np.float32)
    bx = rng.integers(-1, 2, 64).astype(np.float32)
    ref = _ref(x, w0, b0, w2, b2, w4, 0.0, wx, bx, "relu", "relu", prec)
    # the integer intermediates stay below the exact range of the narrower type (bf16: 256)
    xt = torch.from_numpy(x)
    wa = F.relu(F.conv2d(xt, torch.from_numpy(w0).view(64, 64, 1, 1), torch.from_numpy(b0)))
    wb = F.relu(F.conv2d(wa, torch.from_numpy(w2), torch.from_numpy(b2), padding=1))
    assert wa.max() < 256 and wb.max() < 256
    got = eng2.topiq_gate64(x, w0, b0, w2, b2, w4, 0.0, wx, bx, wblk_act="relu", gate_act="relu")
    # only the sigmoid's rounding to the 2-byte type and the final store separate the two
    tol = (1.5e-3 if prec == "f16" else 1e-2) * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol, (float(np.abs(got - ref).max()), float(np.abs(ref).max()))


def test_topiq_scores_with_and_without_the_fused_gate():
    """The whole fp16 TOPIQ model on 256 x 256 and 320 x 224 images (level 0 = 128 x 128 -> 8 x 8, and 160 x 112 -> 10 x 7): the fused
    level and the four-launch form agree within the fp16 rounding of their intermediates, and both stay within the 1e-3 gate of the
    fp32 engine's score (the oracle comparisons of tests/test_precision_policy_gpu.py run on the fused form too: it is the default)."""
    from facet_amd import Engine
    from facet_amd._lib import FE_MODEL_TOPIQ
    from facet_amd.weights import synthetic_images, synthetic_state_dict
    sd = synthetic_state_dict("topiq", 13)
    batches = [synthetic_images(5, 2, 256, 256), synthetic_images(6, 2, 224, 320)]
    out = {}
    for name, prec, env in (("f32", "f32", None), ("fused", "f16", None), ("unfused", "f16", "1")):
        os.environ.pop("FE_NO_FUSED_GATE", None)
        if env is not None:
            os.environ["FE_NO_FUSED_GATE"] = env
        try:
            e = Engine(0, arena_bytes=8 << 30, precision=prec)
            e.load_weights(FE_MODEL_TOPIQ, sd)
            out[name] = np.concatenate([np.asarray(e.topiq_score(b), np.float64) for b in batches])
            e.close()
        finally:
            os.environ.pop("FE_NO_FUSED_GATE", None)
    print("[fused gate]", out)
    rel = lambda a, b: (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max()
    assert rel(out["fused"], out["unfused"]) < 1e-3, out
    assert rel(out["fused"], out["f32"]) < 1e-3, out


def test_fused_kernels_at_baseline_size_against_the_unfused_forms_and_fp32():
    """BASELINE's image size: two 1024 x 1024 images through fp16 TOPIQ with every fused 64-channel kernel on (level-0 gate + pool on the
    512 x 512 map, the three chained bottleneck tails and the halo-tiled gate 3x3 on 256 x 256) and with all of them off
    (FE_NO_FUSED_GATE, FE_NO_FUSED_C64), and through the fp32 engine: within 1e-3 of each other and of fp32."""
    from facet_amd import Engine
    from facet_amd._lib import FE_MODEL_TOPIQ
    from facet_amd.weights import synthetic_images, synthetic_state_dict
    sd = synthetic_state_dict("topiq", 13)
    imgs = synthetic_images(17, 2, 1024, 1024)
    out = {}
    for name, prec, off in (("f32", "f32", False), ("fused", "f16", False), ("unfused", "f16", True)):
        for k in ("FE_NO_FUSED_GATE", "FE_NO_FUSED_C64"):
            os.environ.pop(k, None)
            if off:
                os.environ[k] = "1"
        try:
            e = Engine(0, arena_bytes=16 << 30, precision=prec)
            e.load_weights(FE_MODEL_TOPIQ, sd)
            e.set_microbatch(2)
            out[name] = np.asarray(e.topiq_score(imgs), np.float64)
            e.close()
        finally:
            os.environ.pop("FE_NO_FUSED_GATE", None)
            os.environ.pop("FE_NO_FUSED_C64", None)
    print("[fused kernels @1024]", out)
    rel = lambda a, b: (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max()
    assert rel(out["fused"], out["unfused"]) < 1e-3, out
    assert rel(out["fused"], out["f32"]) < 1e-3, out

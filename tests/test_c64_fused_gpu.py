"""GPU: the halo-tiled 3x3 (64 -> 64) kernel of the 2-byte models and its chained form - conv2 + bn2 + relu + conv3 + bn3 + identity + relu
of a ResNet-50 layer1 bottleneck in one launch (facet_amd/csrc/kernels_c64.hip) - against torch's fp32 arithmetic on the same 2-byte
rounded inputs, and the whole TOPIQ model with and without it.

Tolerances: products of 2-byte values are exact and accumulated in fp32; the plain form rounds its result once (2^-12 relative for
fp16, 2^-9 for bf16); the chained form rounds the 64-channel intermediate to the 2-byte type exactly where the two-launch form stores
it (the torch reference below rounds at the same point) and the result once more. Integer data is exact end to end.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(1, 16, 14), (2, 16, 28), (1, 17, 15), (1, 1, 1), (3, 25, 33), (1, 40, 64), (2, 7, 100)]


@pytest.fixture(scope="module", params=["f16", "bf16"])
def eng2(request):
    from facet_amd import Engine
    e = Engine(0, arena_bytes=8 << 30, precision=request.param)
    e._prec_name = request.param
    yield e
    e.close()


def _rd(a, prec):
    t = torch.from_numpy(np.asarray(a, np.float32))
    return t.half().float() if prec == "f16" else t.bfloat16().float()


def _eps(prec):
    return 2.0 ** -11 if prec == "f16" else 2.0 ** -8


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("act", ["relu", "gelu", None])
def test_plain_3x3_matches_torch(eng2, shape, act):
    n, h, w = shape
    prec = eng2._prec_name
    rng = np.random.default_rng(h * 131 + w)
    x = _rd(rng.normal(0, 1, (n, 64, h, w)), prec)
    w2 = _rd(rng.normal(0, 1 / 24, (64, 64, 3, 3)), prec)
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    shift = rng.normal(0, 0.2, 64).astype(np.float32)
    ref = F.conv2d(x, w2, padding=1) * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1)
    ref = {None: lambda v: v, "relu": F.relu, "gelu": F.gelu}[act](ref).numpy()
    got = eng2.conv3x3_c64(x.numpy(), w2.numpy(), scale, shift, act)
    assert got.shape == ref.shape
    tol = _eps(prec) * np.abs(ref) + (6e-4 if act == "gelu" else 2e-4) * np.abs(ref).max()      # tanh-form GELU: <= 4.8e-4 from erf
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outside tolerance, worst {np.abs(got - ref).max():.3e}"


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_chained_bottleneck_tail_matches_torch(eng2, shape):
    n, h, w = shape
    prec = eng2._prec_name
    rng = np.random.default_rng(h * 17 + w + 5)
    x = _rd(rng.normal(0, 1, (n, 64, h, w)), prec)
    w2 = _rd(rng.normal(0, 1 / 24, (64, 64, 3, 3)), prec)
    w3 = _rd(rng.normal(0, 1 / 8, (256, 64)), prec)
    res = _rd(rng.normal(0, 1, (n, 256, h, w)), prec)
    s2, s3 = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.uniform(0.5, 1.5, 256).astype(np.float32)
    h2, h3 = rng.normal(0, 0.2, 64).astype(np.float32), rng.normal(0, 0.2, 256).astype(np.float32)
    v = lambda a: torch.from_numpy(a).view(1, -1, 1, 1)
    mid = _rd(F.relu(F.conv2d(x, w2, padding=1) * v(s2) + v(h2)).numpy(), prec)
    ref = F.relu(F.conv2d(mid, w3.view(256, 64, 1, 1)) * v(s3) + v(h3) + res).numpy()
    got = eng2.conv3x3_c64(x.numpy(), w2.numpy(), s2, h2, "relu", w3.numpy(), s3, h3, res.numpy())
    assert got.shape == ref.shape
    # a mid value that lands on the other side of a rounding boundary moves the result by one 2-byte ulp of mid times |w3|
    tol = _eps(prec) * np.abs(ref) + 6 * _eps(prec) * np.abs(ref).max() / 16
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outside tolerance, worst {np.abs(got - ref).max():.3e} (max|ref| {np.abs(ref).max():.3e})"


def test_integer_data_is_exact_in_both_forms(eng2):
    """Weights in {-1, 0, 1}, inputs in {-2..2}: every product and sum is an exact small integer in either 2-byte type, so any error in
    the patch / tap / fragment / K-permutation orders shows as a wrong integer. 37 x 45: interior tiles, partial tiles on both edges."""
    rng = np.random.default_rng(3)
    n, h, w = 2, 37, 45
    x = rng.integers(-2, 3, (n, 64, h, w)).astype(np.float32)
    sp = lambda shape, p: (rng.integers(-1, 2, shape) * (rng.random(shape) < p)).astype(np.float32)
    w2 = sp((64, 64, 3, 3), 0.03)
    w3 = sp((256, 64), 0.1)
    res = rng.integers(-3, 4, (n, 256, h, w)).astype(np.float32)
    xt = torch.from_numpy(x)
    c2 = F.conv2d(xt, torch.from_numpy(w2), padding=1)
    assert c2.abs().max() < 128
    got = eng2.conv3x3_c64(x, w2, act2=None)
    assert np.array_equal(got, c2.numpy())
    mid = F.relu(c2)
    ref = F.relu(F.conv2d(mid, torch.from_numpy(w3).view(256, 64, 1, 1)) + torch.from_numpy(res))
    assert ref.abs().max() < 256
    got = eng2.conv3x3_c64(x, w2, None, None, "relu", w3, None, None, res)
    assert np.array_equal(got, ref.numpy())


def test_topiq_scores_with_and_without_the_fused_bottleneck_tails():
    """Whole fp16 TOPIQ on 264 x 328 images (layer1 at 66 x 82: partial tiles): scores with the chained kernel, with the three-launch
    form (FE_NO_FUSED_C64) and of the fp32 engine agree within the 1e-3 gate."""
    from facet_amd import Engine
    from facet_amd._lib import FE_MODEL_TOPIQ
    from facet_amd.weights import synthetic_images, synthetic_state_dict
    sd = synthetic_state_dict("topiq", 13)
    imgs = synthetic_images(9, 3, 264, 328)
    out = {}
    for name, prec, env in (("f32", "f32", None), ("fused", "f16", None), ("unfused", "f16", "1")):
        os.environ.pop("FE_NO_FUSED_C64", None)
        if env is not None:
            os.environ["FE_NO_FUSED_C64"] = env
        try:
            e = Engine(0, arena_bytes=8 << 30, precision=prec)
            e.load_weights(FE_MODEL_TOPIQ, sd)
            out[name] = np.asarray(e.topiq_score(imgs), np.float64)
            e.close()
        finally:
            os.environ.pop("FE_NO_FUSED_C64", None)
    print("[fused c64]", out)
    rel = lambda a, b: (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max()
    assert rel(out["fused"], out["unfused"]) < 1e-3, out
    assert rel(out["fused"], out["f32"]) < 1e-3, out

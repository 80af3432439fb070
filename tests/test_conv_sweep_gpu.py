"""GPU: seeded random sweep of the two contraction kernels (fe_op_conv2d under fp32 and bf16 precision) against torch's fp32 convolution.
The hand-picked cases of test_ops_gpu.py / test_bf16_gpu.py name the tile and epilogue forms; this sweep walks shapes nobody picked:
ragged M and Cout against every tile size, channel counts at the packing boundaries, strides, dilations, every activation, residuals."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "sigmoid": torch.sigmoid, "softplus": F.softplus}


def _cases(seed, count, cin_step):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        k = int(rng.choice([1, 1, 3]))
        cin = int(rng.integers(1, 13 if k == 3 else 40)) * cin_step
        cout = int(rng.choice([int(rng.integers(1, 40)) * 8, int(rng.integers(1, 90)) * 8, int(rng.integers(8, 300))]))
        stride = int(rng.choice([1, 1, 2]))
        dil = int(rng.choice([1, 1, 1, 2])) if k == 3 else 1
        pad = dil if k == 3 else 0
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(5, 40)), int(rng.integers(5, 40))
        if n * h * w * cin * k * k > 6e6 or n * h * w * cout > 3e6:
            continue
        act = list(ACTS)[int(rng.integers(0, len(ACTS)))]
        out.append((n, cin, h, w, cout, k, stride, pad, dil, act, bool(rng.integers(0, 2))))
    return out


def _run(engine, case, rounded):
    n, cin, h, w, cout, k, stride, pad, dil, act, with_res = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))      # stable across processes (str hashes are salted)
    rd = (lambda a: torch.from_numpy(np.asarray(a, np.float32)).bfloat16().float()) if rounded else (lambda a: torch.from_numpy(np.asarray(a, np.float32)))
    x = rd(rng.normal(0, 1, (n, cin, h, w)))
    wt = rd(rng.normal(0, 1.0 / np.sqrt(cin * k * k), (cout, cin, k, k)))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.2, cout).astype(np.float32)
    ref = F.conv2d(x, wt, stride=stride, padding=pad, dilation=dil) * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1)
    res = None
    if with_res:
        res = rd(rng.normal(0, 1, tuple(ref.shape)))
        ref = ref + res
    ref = ACTS[act](ref).numpy()
    got = engine.conv2d(x.numpy(), wt.numpy(), scale=scale, shift=shift, res=None if res is None else res.numpy(), stride=stride, pad=pad, dil=dil, act=act)
    assert got.shape == ref.shape
    return got, ref


@pytest.mark.parametrize("case", _cases(11, 40, 16), ids=lambda c: "x".join(str(v) for v in c))
def test_fp32_contraction_sweep(engine, case):
    got, ref = _run(engine, case, rounded=False)
    err = np.abs(got - ref).max()
    assert err <= 2e-4 * max(1.0, np.abs(ref).max()), f"max err {err:.3e} (max|ref| {np.abs(ref).max():.3e})"


@pytest.fixture(scope="module")
def eng16():
    from facet_amd import Engine
    e = Engine(0, arena_bytes=8 << 30, precision="bf16")
    yield e
    e.close()


@pytest.mark.parametrize("case", _cases(12, 48, 16), ids=lambda c: "x".join(str(v) for v in c))
def test_bf16_contraction_sweep(eng16, case):
    """Same tolerance as tests/test_bf16_gpu.py: exact products, fp32 sums, one rounding of the result to bf16."""
    got, ref = _run(eng16, case, rounded=True)
    tol = 2.0 ** -8 * np.abs(ref) + 1e-3 * np.abs(ref).max()
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outside tolerance, worst {np.abs(got - ref).max():.3e} (max|ref| {np.abs(ref).max():.3e})"

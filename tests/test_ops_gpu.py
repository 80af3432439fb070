"""GPU parity of the single HIP ops (through the C ABI) against plain torch-CPU fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-4  # fp32 MFMA = k-ordered fmaf chain; torch CPU sums in a different order


def _close(a, b, rtol=RTOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max() / scale
    assert err < rtol, f"max rel-to-max err {err:.3e}"


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil
    (2, 64, 56, 56, 64, 1, 1, 0, 1),      # resnet 1x1 (fast path, BN=64 tile)
    (2, 64, 56, 56, 256, 1, 1, 0, 1),     # 1x1 expand (128x128 tile)
    (1, 64, 40, 40, 64, 3, 1, 1, 1),      # 3x3
    (2, 128, 28, 28, 128, 3, 2, 1, 1),    # 3x3 stride 2
    (1, 3, 64, 64, 64, 7, 2, 3, 1),       # stem: Cin 3 -> padded 4, generic path, K=196
    (1, 256, 14, 14, 512, 1, 2, 0, 1),    # downsample 1x1 stride 2
    (1, 16, 31, 29, 16, 3, 1, 2, 2),      # u2netp dilated, ragged spatial, Cout 16 tile
    (1, 32, 17, 17, 16, 3, 1, 8, 8),      # dilation 8 bigger than half the map
    (1, 64, 20, 20, 1, 3, 1, 1, 1),       # side conv Cout=1
    (1, 6, 20, 20, 1, 1, 1, 0, 1),        # outconv Cin 6 -> padded 8
    (3, 24, 9, 9, 40, 3, 1, 1, 1),        # Cin%16 != 0 generic path, Cout not a tile multiple
    (1, 1296, 2, 1, 1024, (2, 1), 1, 0, 1),  # SAMP pattern conv as full-window contraction
    (1, 2048, 1, 1, 1000, 1, 1, 0, 1),    # M = 1 (pure GEMV shape)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(engine, case):
    n, cin, h, w, cout, k, stride, pad, dil = case
    kh, kw = (k, k) if isinstance(k, int) else k
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / np.sqrt(cin * kh * kw)
    ref = F.conv2d(x, wt, None, stride, pad, dil)
    if kh != kw:
        pytest.skip("C ABI conv op takes square kernels; non-square covered by SAMP model test")
    got = engine.conv2d(x.numpy(), wt.numpy(), stride=stride, pad=pad, dil=dil)
    assert got.shape == tuple(ref.shape)
    _close(got, ref.numpy())


def test_conv_epilogue_bn_relu_residual(engine):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 24, 24, generator=g)
    wt = torch.randn(256, 64, 1, 1, generator=g) / 8
    scale = torch.rand(256, generator=g) + 0.5
    shift = torch.randn(256, generator=g)
    res = torch.randn(2, 256, 24, 24, generator=g)
    y = F.conv2d(x, wt) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    _close(engine.conv2d(x.numpy(), wt.numpy(), scale.numpy(), shift.numpy(), res.numpy(), act="relu"),
           F.relu(y + res).numpy())
    _close(engine.conv2d(x.numpy(), wt.numpy(), scale.numpy(), shift.numpy(), res.numpy(), res_after_act=True, act="relu"),
           (F.relu(y) + res).numpy())
    _close(engine.conv2d(x.numpy(), wt.numpy(), scale.numpy(), shift.numpy(), act="gelu"), F.gelu(y).numpy())
    _close(engine.conv2d(x.numpy(), wt.numpy(), scale.numpy(), shift.numpy(), act="sigmoid"), torch.sigmoid(y).numpy())


def test_conv_exact_integers(engine):
    """A = small integers, asymmetric weights: catches any row/col or k-permutation slip exactly."""
    x = (torch.arange(2 * 16 * 6 * 5) % 7 - 3).float().view(2, 16, 6, 5)
    wt = ((torch.arange(48 * 16 * 9) * 5) % 11 - 5).float().view(48, 16, 3, 3)
    ref = F.conv2d(x, wt, None, 1, 1)
    got = engine.conv2d(x.numpy(), wt.numpy(), stride=1, pad=1)
    assert np.array_equal(got, ref.numpy())


@pytest.mark.parametrize("shape,k,s,p,ceil", [((2, 64, 37, 41), 3, 2, 1, False), ((1, 16, 15, 15), 2, 2, 0, True),
                                              ((1, 16, 14, 14), 2, 2, 0, True), ((1, 1, 224, 224), 3, 2, 1, False),
                                              ((2, 3, 7, 9), 2, 2, 0, True), ((1, 8, 9, 10), 3, 1, 1, False), ((1, 4, 10, 10), 3, 2, 0, True),
                                              ((3, 128, 20, 24), 3, 2, 1, False)])
def test_maxpool(engine, shape, k, s, p, ceil):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    ref = F.max_pool2d(x, k, s, p, ceil_mode=ceil)
    got = engine.maxpool2d(x.numpy(), k, s, p, ceil)
    assert got.shape == tuple(ref.shape)
    assert np.array_equal(got, ref.numpy())


@pytest.mark.parametrize("shape,out", [((2, 16, 7, 7), (14, 14)), ((1, 16, 4, 4), (7, 7)), ((1, 1, 56, 56), (7, 7)),
                                       ((1, 3, 5, 9), (11, 4)), ((1, 1, 7, 7), (224, 224))])
def test_bilinear(engine, shape, out):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(2))
    ref = F.interpolate(x, size=out, mode="bilinear", align_corners=False)
    _close(engine.bilinear(x.numpy(), *out), ref.numpy(), 1e-5)


@pytest.mark.parametrize("shape,out", [((2, 64, 64, 64), (32, 32)), ((1, 8, 7, 7), (4, 4)), ((1, 8, 7, 7), (3, 3)),
                                       ((1, 1, 7, 7), (8, 8)), ((1, 5, 10, 6), (1, 1))])
def test_adaptive_avgpool(engine, shape, out):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    ref = F.adaptive_avg_pool2d(x, out)
    _close(engine.adaptive_avgpool(x.numpy(), *out), ref.numpy(), 1e-5)


def test_layernorm(engine):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(300, 1024, generator=g) * 3 + 1
    w = torch.rand(1024, generator=g) + 0.5
    b = torch.randn(1024, generator=g)
    _close(engine.layernorm(x.numpy(), w.numpy(), b.numpy(), 1e-5), F.layer_norm(x, (1024,), w, b, 1e-5).numpy(), 1e-5)


@pytest.mark.parametrize("variant", [1, 2, 4, 7, 11, 12, 13, 14, 17, 18, 21, 22])
def test_conv_tile_variants_agree(engine, variant):
    """Every tile variant of the contraction kernel (register-staged, LDS-DMA, loader-wave) gives the same answer,
    including ragged M/N edges, padding taps, stride 2 and a K that needs the zero-padded tail slab."""
    g = torch.Generator().manual_seed(77)
    cases = [(2, 48, 19, 23, 80, 3, 1, 1), (1, 64, 30, 30, 200, 1, 1, 0), (1, 32, 33, 31, 64, 3, 2, 1),
             (3, 16, 9, 9, 40, 3, 1, 1)]
    try:
        for n, cin, h, w, cout, k, s, p in cases:
            x = torch.randn(n, cin, h, w, generator=g)
            wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
            res = torch.randn(n, cout, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1, generator=g)
            ref = F.relu(F.conv2d(x, wt, None, s, p) + res)
            engine.set_conv_variant(variant)
            got = engine.conv2d(x.numpy(), wt.numpy(), res=res.numpy(), stride=s, pad=p, act="relu")
            _close(got, ref.numpy())
    finally:
        engine.set_conv_variant(0)


@pytest.mark.parametrize("cout,hw,act", [(64, (45, 77), "relu"), (32, (64, 64), None), (64, (7, 9), "relu"), (64, (130, 33), None)])
def test_stem7x7_kernel_matches_torch(engine, cout, hw, act):
    """7x7 / stride 2 / pad 3 / Cin 3 takes the LDS-resident stem kernel (kernels_stem.hip); ragged tiles on every border.
    Tolerance 2e-4 of max-abs like the other conv tests."""
    rng = np.random.default_rng(cout + hw[0])
    x = rng.standard_normal((2, 3, hw[0], hw[1])).astype(np.float32)
    w = (rng.standard_normal((cout, 3, 7, 7)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    got = engine.conv2d(x, w, scale=sc, shift=sh, stride=2, pad=3, act=act)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=2, padding=3) * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1)
    if act == "relu":
        ref = F.relu(ref)
    ref = ref.numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("cout,stride,hw", [(64, 1, (40, 50)), (32, 2, (61, 37)), (64, 2, (16, 16)), (32, 1, (9, 70))])
def test_stem3x3_kernel_matches_torch(engine, cout, stride, hw):
    """3x3 / pad 1 / Cin 3 first layers (face graphs) take the same LDS-resident stem kernel."""
    rng = np.random.default_rng(cout * stride + hw[1])
    x = rng.standard_normal((2, 3, hw[0], hw[1])).astype(np.float32)
    w = (rng.standard_normal((cout, 3, 3, 3)) * 0.2).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    got = engine.conv2d(x, w, shift=sh, stride=stride, pad=1, act="relu")
    ref = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(sh), stride=stride, padding=1)).numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("cin,cout,hw,act", [(256, 256, (14, 14), "relu"), (512, 384, (13, 17), None), (256, 64, (1, 5), "relu"), (320, 256, (32, 32), "relu"),
                                             (128, 128, (30, 41), "relu"), (128, 256, (6, 3), None)])
def test_winograd_path_matches_torch(engine, cin, cout, hw, act):
    """3x3 / stride 1 / pad 1 with Cin >= 128 runs as Winograd F(4x4,3x3): transforms + 36 batched GEMMs in one launch (odd sizes:
    partial tiles on both borders). Same 2e-4-of-max tolerance as the direct kernels."""
    rng = np.random.default_rng(cin + hw[0])
    x = rng.standard_normal((2, cin, hw[0], hw[1])).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * (1.0 / np.sqrt(9 * cin))).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    got = engine.conv2d(x, w, scale=sc, shift=sh, stride=1, pad=1, act=act)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1) * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1)
    if act == "relu":
        ref = F.relu(ref)
    ref = ref.numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("res_after", [False, True])
def test_winograd_with_residual(engine, res_after):
    """Residual add before / after the activation inside the Winograd output transform (ResNet basic block, IResNet block)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 128, 11, 9)).astype(np.float32)
    w = (rng.standard_normal((160, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    r = rng.standard_normal((2, 160, 11, 9)).astype(np.float32)
    sh = rng.standard_normal(160).astype(np.float32)
    got = engine.conv2d(x, w, shift=sh, res=r, res_after_act=res_after, stride=1, pad=1, act="relu")
    y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(sh), padding=1)
    ref = (F.relu(y) + torch.from_numpy(r)) if res_after else F.relu(y + torch.from_numpy(r))
    assert np.abs(got - ref.numpy()).max() <= 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,cin,cout,act,res", [(8, 25088, 512, None, False), (3, 4096, 96, "relu", True), (130, 8192, 64, None, False),
                                                (1, 6144, 1000, "relu", False)])
def test_split_k_long_contraction(engine, m, cin, cout, act, res):
    """Few rows x long K (ArcFace's 25088 -> 512 embedding layer): K is cut into slices run as one batched launch and summed in a
    fixed order (engine.hip conv_forward). Checked against float64, with scale / shift / residual / ReLU in the reduce pass."""
    rng = np.random.default_rng(cin + m)
    x = rng.standard_normal((m, cin, 1, 1)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    r = rng.standard_normal((m, cout, 1, 1)).astype(np.float32) if res else None
    got = engine.conv2d(x, w, scale=sc, shift=sh, res=r, act=act)
    ref = (x[:, :, 0, 0].astype(np.float64) @ w[:, :, 0, 0].astype(np.float64).T) * sc + sh
    if res:
        ref = ref + r[:, :, 0, 0]
    if act == "relu":
        ref = np.maximum(ref, 0)
    assert got.shape == (m, cout, 1, 1) and np.abs(got[:, :, 0, 0] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    a = engine.conv2d(x, w, scale=sc, shift=sh, res=r, act=act)
    assert np.array_equal(a, got)                          # fixed summation order: run-to-run identical

"""GPU: the bf16 path (BASELINE.json configs[3]: TOPIQ + SAMP-Net + CLIP ViT-L/14 in bf16) against the fp32 oracle.

Tolerances, stated once here and in DESIGN.md:
  * one contraction (fe_op_conv2d under bf16 precision) against torch's fp32 convolution of the SAME bf16-rounded inputs: the bf16
    matrix cores multiply exactly and accumulate in fp32, so the only differences are summation order and the single rounding of
    the result to bf16 (2^-9 relative): |diff| <= 2^-8 * |ref| + 1e-3 * max|ref|.
  * whole models against the fp32 oracle fed the fp32 weights: bf16 carries 8 significant bits per stored activation, through
    ~50 (ResNet-50 + CFANet) to ~100 (ViT-L/14) layers. Measured on the seeded checkpoints: TOPIQ MOS within 2e-2 relative,
    CLIP embedding cosine >= 0.999, SAMP logits / attributes / distribution within 3e-2 absolute of values of order 1.
    north_star's 1e-3 is stated for fp32 and is NOT claimed for bf16.
"""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from facet_amd._lib import (FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP, FE_RECORD_FLOATS)
from facet_amd.weights import synthetic_state_dict, synthetic_images

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng16():
    from facet_amd import Engine
    e = Engine(0, arena_bytes=16 << 30, precision="bf16")
    yield e
    e.close()


def _r16(a):
    return torch.from_numpy(np.asarray(a, np.float32)).bfloat16().float()


CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil, act, res
    (2, 64, 20, 24, 256, 1, 1, 0, 1, None, False),        # 1x1, K = 64 (one K-step)
    (1, 256, 17, 19, 64, 1, 1, 0, 1, "relu", True),       # 1x1 with residual, ragged M
    (2, 64, 18, 22, 64, 3, 1, 1, 1, "relu", False),       # 3x3, channel block 32, K = 576
    (1, 16, 33, 31, 16, 3, 1, 1, 1, "relu", False),       # U2-Net-P mid layers: Cin = 16 -> two (block, tap) units per slab, 9 units (odd)
    (1, 48, 15, 15, 32, 3, 1, 1, 1, None, False),         # Cin = 48: three 16-blocks, 27 units
    (1, 32, 20, 20, 16, 3, 1, 2, 2, "relu", True),        # dilation 2 + residual, Cout = 16
    (2, 128, 16, 16, 128, 3, 2, 1, 1, "relu", False),     # stride 2
    (1, 40, 9, 13, 24, 1, 1, 0, 1, "gelu", False),        # 1x1 with Cin % 32 != 0: the chunk past Cin is zero filled
    (1, 64, 12, 12, 20, 3, 1, 1, 1, "sigmoid", False),    # Cout % 8 != 0: scalar epilogue
    (1, 512, 7, 7, 1024, 1, 1, 0, 1, None, False),        # wide N, small M
    (3, 64, 28, 28, 64, 3, 1, 1, 1, "softplus", False),
    # wide wave tiles (K >= 256, Cout > 64: transposed accumulators, register epilogue, LDS-staged 16-byte stores), every epilogue form
    (2, 256, 17, 19, 192, 1, 1, 0, 1, None, False),       # ragged M (646 rows), Cout = 1.5 tile columns
    (2, 256, 17, 19, 256, 1, 1, 0, 1, "relu", True),      # residual before the activation
    (1, 512, 23, 11, 384, 1, 1, 0, 1, "gelu", False),
    (1, 1024, 16, 20, 128, 1, 1, 0, 1, None, True),       # long K, residual, no activation (ViT projection form)
    (1, 256, 14, 14, 512, 1, 1, 0, 1, "sigmoid", False),
    (1, 256, 14, 14, 136, 1, 1, 0, 1, "softplus", True),  # parameter-driven form; Cout % 8 == 0 but not a multiple of 32
    (1, 256, 20, 20, 256, 3, 1, 1, 1, "relu", False),     # 3x3, K = 2304
    (5, 320, 32, 32, 256, 1, 1, 0, 1, "gelu", True),      # 5120 rows: 256-row tiles, several per column
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_bf16_contraction_matches_fp32_on_rounded_inputs(eng16, case):
    n, cin, h, w, cout, k, stride, pad, dil, act, with_res = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))      # stable across processes (str hashes are salted)
    x = _r16(rng.normal(0, 1, (n, cin, h, w)))
    wt = _r16(rng.normal(0, 1.0 / np.sqrt(cin * k * k), (cout, cin, k, k)))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.2, cout).astype(np.float32)
    ref = F.conv2d(x, wt, stride=stride, padding=pad, dilation=dil) * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1)
    res = None
    if with_res:
        res = _r16(rng.normal(0, 1, tuple(ref.shape)))
        ref = ref + res
    ref = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "sigmoid": torch.sigmoid, "softplus": F.softplus}[act](ref).numpy()
    got = eng16.conv2d(x.numpy(), wt.numpy(), scale=scale, shift=shift, res=None if res is None else res.numpy(), stride=stride, pad=pad, dil=dil, act=act)
    assert got.shape == ref.shape
    tol = 2.0 ** -8 * np.abs(ref) + 1e-3 * np.abs(ref).max()
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outside tolerance, worst {np.abs(got - ref).max():.3e} (max|ref| {np.abs(ref).max():.3e})"


def test_bf16_integer_data_is_exact(eng16):
    """Small integers are exact in bf16 and their sums exact in fp32: any indexing mistake (tap order, channel blocks, swizzle,
    fragment maps) shows up as a wrong integer, not as 'noise'. Asymmetric weights, non-square image, both channel-block sizes."""
    rng = np.random.default_rng(5)
    for cin, cout in ((32, 40), (16, 8), (96, 64), (64, 192)):       # the last one: K = 576 on the wide tiles
        x = rng.integers(-1, 2, (2, cin, 11, 14)).astype(np.float32)
        w = rng.integers(-1, 2, (cout, cin, 3, 3)).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy()
        assert np.abs(ref).max() < 256                    # results representable exactly in bf16
        got = eng16.conv2d(x, w, pad=1)
        assert np.array_equal(got, ref), (cin, cout, float(np.abs(got - ref).max()))


def _load(e, names, seed):
    ids = {"topiq": FE_MODEL_TOPIQ, "clip": FE_MODEL_CLIP, "aesthetic": FE_MODEL_AESTHETIC, "u2netp": FE_MODEL_U2NETP, "samp_net": FE_MODEL_SAMP}
    sd = {}
    for nme in names:
        sd[nme] = synthetic_state_dict(nme, seed)
        e.load_weights(ids[nme], sd[nme])
    return sd


def test_model_precision_is_reported(eng16):
    _load(eng16, ["aesthetic"], 3)
    assert eng16.model_precision(FE_MODEL_AESTHETIC) == "f32" and eng16.model_precision(FE_MODEL_SAMP) in (None, "bf16")


def test_clip_bf16_embedding_against_fp32_oracle(eng16):
    from oracle.clip_vit import CLIPImage, aesthetic_head
    sd = _load(eng16, ["clip", "aesthetic"], 13)
    assert eng16.model_precision(FE_MODEL_CLIP) == "bf16"
    x = np.random.default_rng(1).normal(0, 1, (3, 3, 224, 224)).astype(np.float32)
    feat, emb, aes = eng16.clip_encode_image(x, normalized=True, aesthetic=True)
    net = CLIPImage().eval(); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd["clip"].items()})
    head = aesthetic_head().eval(); head.load_state_dict({k: torch.from_numpy(v) for k, v in sd["aesthetic"].items()})
    with torch.no_grad():
        f = net.encode_image(torch.from_numpy(x))
        e = F.normalize(f, dim=-1).numpy()
        a = head(f).flatten().numpy()
    cos = (emb * e).sum(1)
    print("[bf16 clip] cosine to fp32 oracle", cos, "aesthetic raw", aes, a)
    assert cos.min() > 0.999
    assert np.abs(aes - a).max() < 3e-2 * max(1.0, np.abs(a).max())
    assert np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-5        # normalisation itself is fp32


@pytest.mark.parametrize("hw,n", [((160, 192), 3), ((256, 256), 2)])
def test_topiq_bf16_score_against_fp32_oracle(eng16, hw, n):
    from oracle.topiq import CFANet
    sd = _load(eng16, ["topiq"], 3)
    imgs = synthetic_images(2, n, *hw)
    eng16.set_microbatch(2)
    got = eng16.topiq_score(imgs)
    net = CFANet().eval(); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd["topiq"].items()})
    with torch.no_grad():
        ref = net(torch.from_numpy(imgs.astype(np.float32) / 255.0).permute(0, 3, 1, 2)).flatten().numpy()
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    print("[bf16 topiq]", got, ref, rel)
    assert rel.max() < 2e-2


@pytest.mark.parametrize("hw", [(128, 160), (97, 139)])      # the odd size: ragged stem tiles (8 x 32 outputs), ragged pooling, ragged M everywhere
def test_topiq_bf16_pyramid_levels(eng16, hw):
    from oracle.resnet import ResNet50Features
    sd = _load(eng16, ["topiq"], 3)
    imgs = synthetic_images(1, 2, *hw)
    net = ResNet50Features().eval()
    net.load_state_dict({k[len("semantic_model."):]: torch.from_numpy(v) for k, v in sd["topiq"].items() if k.startswith("semantic_model.")})
    m, s = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    with torch.no_grad():
        ref = net((torch.from_numpy(imgs.astype(np.float32) / 255.0).permute(0, 3, 1, 2) - m) / s)
    for level in range(5):
        got = eng16.topiq_features(imgs, level)
        r = ref[level].numpy()
        err = np.abs(got - r).max() / np.abs(r).max()
        print(f"[bf16 resnet50] level {level}: max err / max = {err:.3e}")
        assert got.shape == r.shape and err < 3e-2


def test_samp_bf16_against_reference_golden(eng16):
    """The SAMP-Net / U2-Net-P golden vectors come from the REFERENCE's own classes (tests/golden/make_samp_golden.py): the bf16 path
    is held to them directly."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "samp_golden.npz"))
    seed = int(g["seed_w"])
    eng16.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", seed))
    eng16.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", seed))
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(g["seed_x"]))).numpy()
    pw, at, sdist, sal = eng16.samp_forward(x, want_saliency=True)
    print("[bf16 samp vs reference golden] sal", float(np.abs(sal[:, 0, ::8, ::8] - g["saliency_ds"]).max()), "pw", float(np.abs(pw - g["pattern_weights"]).max()),
          "attr", float(np.abs(at - g["attributes"]).max()), "dist", float(np.abs(sdist - g["score_dist"]).max()))
    assert np.abs(sal[:, 0, ::8, ::8] - g["saliency_ds"]).max() < 3e-2
    assert np.abs(pw - g["pattern_weights"]).max() < 3e-2 * max(1.0, np.abs(g["pattern_weights"]).max())
    assert np.array_equal(pw.argmax(1), g["pattern_weights"].argmax(1))
    assert np.abs(at - g["attributes"]).max() < 3e-2 and np.abs(sdist - g["score_dist"]).max() < 3e-2


def test_samp_bf16_against_fp32_oracle(eng16):
    from oracle.sampnet import U2NETP, SAMPNet
    sd = _load(eng16, ["u2netp", "samp_net"], 13)
    x = np.random.default_rng(2).normal(0, 1, (3, 3, 224, 224)).astype(np.float32)
    pw, at, sdist, sal = eng16.samp_forward(x, want_saliency=True)
    u2 = U2NETP().eval(); u2.load_state_dict({k: torch.from_numpy(v) for k, v in sd["u2netp"].items()})
    sn = SAMPNet().eval(); sn.load_state_dict({k: torch.from_numpy(v) for k, v in sd["samp_net"].items()})
    with torch.no_grad():
        xt = torch.from_numpy(x)
        s_ref = u2(xt)
        p_ref, a_ref, d_ref = sn(xt, s_ref)
    s_ref = s_ref[0] if isinstance(s_ref, (tuple, list)) else s_ref
    print("[bf16 samp] sal err", float(np.abs(sal - s_ref.numpy().reshape(sal.shape)).max()), "pw err", float(np.abs(pw - p_ref.numpy()).max()),
          "attr err", float(np.abs(at - a_ref.numpy()).max()), "dist err", float(np.abs(sdist - d_ref.numpy()).max()))
    assert np.abs(sal - s_ref.numpy().reshape(sal.shape)).max() < 3e-2
    assert np.abs(pw - p_ref.numpy()).max() < 3e-2 * max(1.0, np.abs(p_ref.numpy()).max())
    assert np.abs(at - a_ref.numpy()).max() < 3e-2 and np.abs(sdist - d_ref.numpy()).max() < 3e-2
    assert (pw.argmax(1) == p_ref.numpy().argmax(1)).all()


def test_ensemble_bf16_records_against_fp32_engine(eng16, engine):
    """configs[3] end to end: fe_ensemble_score of a bf16 context against the fp32 context on the same weights and images, incl. one
    1024x1024 image (the fp32 context itself is held to the oracle by tests/test_ensemble_gpu.py and test_fullsize_gpu.py)."""
    names = ["topiq", "clip", "aesthetic", "u2netp", "samp_net"]
    _load(eng16, names, 13)
    _load(engine, names, 13)
    for hw, n in (((288, 352), 3), ((1024, 1024), 1)):
        imgs = synthetic_images(8, n, *hw)
        eng16.set_microbatch(2); engine.set_microbatch(2)
        r16, m16 = eng16.ensemble_score(imgs)
        r32, m32 = engine.ensemble_score(imgs)
        assert m16 == m32 == 7 and r16.shape == (n, FE_RECORD_FLOATS)
        rel_t = np.abs(r16[:, 0] - r32[:, 0]) / np.maximum(np.abs(r32[:, 0]), 1e-3)
        cos = (r16[:, 21:] * r32[:, 21:]).sum(1)
        print(f"[bf16 ensemble {hw}] topiq rel {rel_t}, aesthetic {r16[:, 1]} vs {r32[:, 1]}, clip cosine {cos}, "
              f"samp max diff {np.abs(r16[:, 2:21] - r32[:, 2:21]).max():.3e}")
        assert rel_t.max() < 2e-2
        assert np.abs(r16[:, 1] - r32[:, 1]).max() < 3e-2 * max(1.0, np.abs(r32[:, 1]).max())
        assert cos.min() > 0.999
        assert np.abs(r16[:, 2:10] - r32[:, 2:10]).max() < 3e-2 * max(1.0, np.abs(r32[:, 2:10]).max())
        assert np.abs(r16[:, 10:21] - r32[:, 10:21]).max() < 3e-2


def test_clip_in_bf16_beside_fp32_models_in_one_context(eng16, engine):
    """The reference's own GPU precisions (processing/scorer.py:513-516 halves CLIP only): precision is a property of a model's
    committed weights, so one context can hold CLIP in bf16 beside TOPIQ / SAMP-Net in fp32. Every field of the record must then be
    exactly what the context of that field's precision produces on its own."""
    from facet_amd import Engine
    names = ["topiq", "clip", "aesthetic", "u2netp", "samp_net"]
    _load(eng16, names, 13)
    _load(engine, names, 13)
    mixed = Engine(0, arena_bytes=8 << 30)
    try:
        _load(mixed, ["topiq", "u2netp", "samp_net"], 13)
        mixed.set_precision("bf16")
        _load(mixed, ["clip", "aesthetic"], 13)
        mixed.set_precision("f32")
        assert mixed.model_precision(FE_MODEL_CLIP) == "bf16" and mixed.model_precision(FE_MODEL_TOPIQ) == "f32"
        assert mixed.model_precision(FE_MODEL_SAMP) == "f32" and mixed.model_precision(FE_MODEL_U2NETP) == "f32"
        imgs = synthetic_images(9, 3, 224, 320)
        for e in (mixed, eng16, engine):
            e.set_microbatch(2)
        rm, mm = mixed.ensemble_score(imgs)
        r16, _ = eng16.ensemble_score(imgs)
        r32, _ = engine.ensemble_score(imgs)
        assert mm == 7
        assert np.array_equal(rm[:, 0], r32[:, 0]) and np.array_equal(rm[:, 2:21], r32[:, 2:21])      # TOPIQ, SAMP-Net: the fp32 path
        assert np.array_equal(rm[:, 1], r16[:, 1]) and np.array_equal(rm[:, 21:], r16[:, 21:])        # aesthetic, embedding: the bf16 tower
    finally:
        mixed.close()


@pytest.mark.parametrize("hw", [(97, 131), (33, 500), (64, 64)])
def test_topiq_bf16_arbitrary_sizes(eng16, hw):
    """The edge cases of tests/test_edge_cases_gpu.py on the bf16 path: sizes that are not multiples of 32 (ceil-mode pyramid, ragged
    adaptive pooling windows, odd token grids, ragged stem tiles), against the fp32 oracle at the bf16 tolerance."""
    from oracle.topiq import CFANet
    sd = _load(eng16, ["topiq"], 3)["topiq"]
    net = CFANet().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    imgs = synthetic_images(21, 2, *hw)
    with torch.no_grad():
        ref = net(torch.from_numpy(imgs.astype(np.float32) / 255).permute(0, 3, 1, 2)).flatten().numpy()
    eng16.set_microbatch(8)
    got = eng16.topiq_score(imgs)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    print(f"[bf16 topiq {hw}] {got} vs {ref}: rel {rel}")
    assert rel.max() < 2e-2


def test_bf16_single_image_microbatch_and_long_edge_cap(eng16):
    """Micro-batch larger than the batch / of one image give the same scores up to the row-count-dependent tile choices; an image with a
    long edge > 1024 is LANCZOS-reduced on the GPU exactly as PyIQAScorer._preprocess_image does with PIL on the host
    (models/pyiqa_scorer.py:131-153) - that stage is precision-independent, so engine(big) == engine(PIL-resized) bit for bit."""
    from PIL import Image
    _load(eng16, ["topiq"], 3)
    imgs = synthetic_images(22, 3, 96, 96)
    eng16.set_microbatch(64)
    a = eng16.topiq_score(imgs)
    eng16.set_microbatch(1)
    b = eng16.topiq_score(imgs)
    one = eng16.topiq_score(imgs[1:2])
    assert np.allclose(a, b, rtol=2e-2) and np.allclose(one, a[1:2], rtol=2e-2)
    big = synthetic_images(31, 2, 700, 1400)
    scale = 1024 / 1400
    small = np.stack([np.asarray(Image.fromarray(x).resize((int(1400 * scale), int(700 * scale)), Image.LANCZOS)) for x in big])
    eng16.set_microbatch(2)
    assert np.array_equal(eng16.topiq_score(big), eng16.topiq_score(small))

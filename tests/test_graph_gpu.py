"""ONNX-subset graph runtime (facet_amd/csrc/onnx_graph.hip) against the torch-CPU ONNX oracle (oracle/onnx_ref.py).

The graphs are seeded stand-ins of the buffalo_l architectures (facet_amd/synthetic_onnx.py); parity is UNPINNED against
onnxruntime (absent offline) - what is compared is ONNX operator semantics, engine vs oracle, on identical .onnx bytes.
Tolerance: 1e-3 of the output's max magnitude (fp32 both sides, different summation orders over K up to 25088).
"""
import numpy as np
import pytest

from standins import onnx_writer as W
from standins import synthetic_onnx as S
from oracle import onnx_ref

pytestmark = pytest.mark.gpu


def _check(engine, data, x, slot=3, tol=1e-3):
    engine.graph_load(slot, data)
    got = engine.graph_run(slot, x)
    want = onnx_ref.run(data, x)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.shape == w.shape
        scale = max(float(np.abs(w).max()), 1e-6)
        assert float(np.abs(g - w).max()) <= tol * scale, (g.shape, float(np.abs(g - w).max()), scale)
    engine.graph_unload(slot)
    return got


@pytest.mark.parametrize("explicit_bn", [False, True])
def test_arcface_small(engine, explicit_bn):
    data, info = S.arcface_iresnet(layers=(1, 2, 1, 1), seed=3, explicit_bn=explicit_bn)
    x = np.random.default_rng(1).uniform(-1, 1, (3, 3, 112, 112)).astype(np.float32)
    _check(engine, data, x)


def test_arcface_r50_full(engine):
    data, info = S.arcface_iresnet(seed=5)
    assert abs(info["macs"] / 1e9 - 6.3) < 0.2      # ArcFace-R50 @112 ~ 6.3 GMAC
    x = np.random.default_rng(2).uniform(-1, 1, (2, 3, 112, 112)).astype(np.float32)
    (emb,) = _check(engine, data, x)
    assert emb.shape == (2, 512)


def test_scrfd_like(engine):
    data, info = S.scrfd_like(seed=7, size=160)
    x = np.random.default_rng(3).uniform(-1, 1, (1, 3, 160, 160)).astype(np.float32)
    outs = _check(engine, data, x)
    assert [o.shape for o in outs] == [(800, 1), (200, 1), (50, 1), (800, 4), (200, 4), (50, 4), (800, 10), (200, 10), (50, 10)]
    assert all(0.0 <= float(o.min()) and float(o.max()) <= 1.0 for o in outs[:3])


def test_landmark_like(engine):
    data, info = S.landmark_like(seed=9)
    x = np.random.default_rng(4).uniform(0, 255, (2, 3, 192, 192)).astype(np.float32)
    (pts,) = _check(engine, data, x)
    assert pts.shape == (2, 212)
    e = engine
    e.graph_load(3, data)
    inf = e.graph_info(3)
    assert inf["has_sub"] and inf["has_mul"] and inf["input_dims"][1:] == [3, 192, 192]
    e.graph_unload(3)


def test_misc_ops(engine):
    """Operators outside the three big graphs: Concat, GlobalAveragePool, LeakyRelu, linear Resize, MatMul, Softmax, and the
    Shape/Gather/Concat/Reshape arithmetic torch exporters emit for dynamic views."""
    g = W.GraphBuilder(21)
    a = g.conv("x", 3, 24, 3, 2)
    b = g.op("LeakyRelu", [g.conv("x", 3, 10, 3, 2)], alpha=0.1)
    c = g.op("Concat", [a, b], axis=1)                         # 34 channels -> padded layout with a gap
    c = g.relu(g.bn(c, 34))
    up = g.op("Resize", [c, g.const(np.zeros(0, np.float32)), g.const(np.asarray([1, 1, 2, 2], np.float32))], mode="linear",
              coordinate_transformation_mode="pytorch_half_pixel")
    d = g.conv(up, 34, 16, 1, 1, p=0)
    pooled = g.op("GlobalAveragePool", [d])
    shp = g.op("Shape", [pooled])
    n0 = g.op("Gather", [shp, g.const(np.asarray(0, np.int64))], axis=0)
    n0 = g.op("Unsqueeze", [n0], axes=[0])
    tgt = g.op("Concat", [n0, g.const(np.asarray([-1], np.int64))], axis=0)
    flat = g.op("Reshape", [pooled, tgt])
    mm = g.op("MatMul", [flat, g.const(np.random.default_rng(5).standard_normal((16, 7)).astype(np.float32))])
    sm = g.op("Softmax", [mm], axis=1)
    data = g.build([("x", ["N", 3, 40, 56])], [(sm, ["N", 7]), (d, ["N", 16, 40, 56])])
    x = np.random.default_rng(6).uniform(-1, 1, (2, 3, 40, 56)).astype(np.float32)
    outs = _check(engine, data, x)
    assert np.allclose(outs[0].sum(1), 1.0, atol=1e-5)


def test_unsupported_operator_is_reported(engine):
    from facet_amd._lib import EngineError
    g = W.GraphBuilder(1)
    y = g.op("Erf", [g.conv("x", 3, 8, 3, 1)])
    data = g.build([("x", [1, 3, 16, 16])], [(y, [1, 8, 16, 16])])
    engine.graph_load(3, data)
    with pytest.raises(EngineError, match="Erf"):
        engine.graph_run(3, np.zeros((1, 3, 16, 16), np.float32))
    engine.graph_unload(3)
    with pytest.raises(EngineError):
        engine.graph_run(3, np.zeros((1, 3, 16, 16), np.float32))


def test_tensor_encodings_and_legacy_input_lists(engine):
    """The same network written with raw_data, with typed repeated fields, with float16 weights and with every initializer also
    listed as a graph input (IR < 4) must load and give the same answer (fp16: the answer of the fp16-rounded weights)."""
    def build(dtype, **kw):
        g = W.GraphBuilder(33)
        y = g.relu(g.bn(g.conv("x", 3, 20, 3, 2, bias=True), 20))
        y = g.conv(y, 20, 32, 3, 1)
        y = g.op("Flatten", [g.op("GlobalAveragePool", [y])], axis=1)
        y = g.gemm(y, 32, 10)
        if dtype is not None:
            g.init = {k: (v.astype(dtype) if v.dtype == np.float32 else v) for k, v in g.init.items()}
        return g.build([("x", ["N", 3, 32, 48])], [(y, ["N", 10])], **kw)
    x = np.random.default_rng(8).uniform(-1, 1, (2, 3, 32, 48)).astype(np.float32)
    base = _check(engine, build(None), x)[0]
    for data in (build(None, encoding="typed"), build(None, list_initializers_as_inputs=True), build(np.float64, encoding="typed"),
                 build(np.float64)):
        assert np.allclose(_check(engine, data, x)[0], base, rtol=1e-5, atol=1e-6)
    half = _check(engine, build(np.float16), x)[0]
    half_typed = _check(engine, build(np.float16, encoding="typed"), x)[0]
    assert np.allclose(half, half_typed, rtol=1e-6, atol=1e-7) and np.abs(half - base).max() < 0.05 * np.abs(base).max()


def test_winograd_inside_graphs_with_padded_channels(engine):
    """3x3 convs of a graph whose channel counts are not multiples of 16 (150 -> physical 160, 200 -> 208): the Winograd path must
    work on the physical (padded) layout; with a residual Add + Relu and a PRelu behind them."""
    g = W.GraphBuilder(44)
    a = g.relu(g.conv("x", 3, 150, 3, 1))
    b = g.conv(a, 150, 150, 3, 1)                      # Cin 150 (pad 160): Winograd, residual + relu fused into the output transform
    c = g.relu(g.add(b, a))
    d = g.prelu(g.conv(c, 150, 200, 3, 1), 200)        # PRelu epilogue
    e = g.conv(d, 200, 24, 3, 1)                       # physical Cin 208: not a multiple of 32 -> direct kernel
    data = g.build([("x", ["N", 3, 21, 30])], [(e, ["N", 24, 21, 30]), (c, ["N", 150, 21, 30])])
    x = np.random.default_rng(2).uniform(-1, 1, (2, 3, 21, 30)).astype(np.float32)
    _check(engine, data, x)


def _torch_exported():
    import glob
    import os
    return sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "torch_onnx_*.npz")))


@pytest.mark.parametrize("path", _torch_exported(), ids=lambda p: p.split("torch_onnx_")[-1][:-4])
def test_graphs_from_torchs_exporter_match_torch(engine, path):
    """Models serialised by PyTorch's own exporter (not by facet_amd/onnx_writer.py), checked against the outputs TORCH computed
    (tests/golden/make_torch_onnx_golden.py): IResNet-style blocks with un-fused BatchNormalization / PRelu / Flatten / Gemm /
    1-D BatchNormalization, an FPN detector with Resize / MaxPool / Transpose / Reshape / Mul-by-scalar heads in opsets 11 and 13 (also exported with dynamic
    axes: Shape / Gather / Unsqueeze / Concat / Slice / Cast chains feeding Resize, run at two input sizes),
    a depthwise-separable landmark net with GlobalAveragePool. Other batch sizes than the exported one run too."""
    z = np.load(path)
    blob, x = z["onnx"].tobytes(), z["x"]
    engine.graph_load(3, blob)
    got = engine.graph_run(3, x)
    want = [z[f"y{i}"] for i in range(len(got))]
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert float(np.abs(g - w).max()) <= 2e-4 * max(1.0, float(np.abs(w).max())), (path, g.shape, float(np.abs(g - w).max()))
    if "x2" in z.files:                       # dynamic-shape export: Shape / Gather / Concat / Slice chains folded for another input size
        got2 = engine.graph_run(3, z["x2"])
        for i, g in enumerate(got2):
            w = z[f"z{i}"]
            assert g.shape == w.shape and float(np.abs(g - w).max()) <= 2e-4 * max(1.0, float(np.abs(w).max())), (i, g.shape)
    if "det" not in path:                     # the detector's Reshape(-1, C) folds the batch into the anchors; the others are per-row
        xb = np.concatenate([x, x[::-1] * 0.5], 0)
        gb = engine.graph_run(3, xb)
        assert gb[0].shape[0] == xb.shape[0] and float(np.abs(gb[0][:x.shape[0]] - got[0]).max()) <= 1e-5 * max(1.0, float(np.abs(got[0]).max()))
    engine.graph_unload(3)


def test_full_size_iresnet50_exported_by_torch(engine):
    """The real ArcFace-R50 structure (IResNet-50: 3-4-14-3 blocks, 25088 -> 512 embedding layer, 1-D BatchNormalization), built in
    torch with seeded weights, serialised on the spot by PyTorch's exporter and run on the engine: embeddings equal torch's.
    (w600k_r50.onnx itself is such a PyTorch export; the file is not available offline.)"""
    import importlib.util
    import io
    import os
    import torch
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_torch_onnx_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)                                # also installs the exporter bypass for the absent `onnx` package
    nn = torch.nn

    class IResNet50(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
            blocks, cin = [], 64
            for cout, n in ((64, 3), (128, 4), (256, 14), (512, 3)):
                for b in range(n):
                    blocks.append(mk.IBlock(cin, cout, 2 if b == 0 else 1))
                    cin = cout
            self.body = nn.Sequential(*blocks)
            self.bn2, self.fc, self.features = nn.BatchNorm2d(512), nn.Linear(512 * 7 * 7, 512), nn.BatchNorm1d(512)

        def forward(self, x):
            return self.features(self.fc(torch.flatten(self.bn2(self.body(self.stem(x))), 1)))

    torch.manual_seed(4)
    g = torch.Generator().manual_seed(5)
    m = IResNet50().eval()
    with torch.no_grad():
        mk.randomise(m, g)
        x = torch.rand((3, 3, 112, 112), generator=g) * 2 - 1
        ref = m(x).numpy()
        f = io.BytesIO()
        torch.onnx.export(m, (x,), f, dynamo=False, opset_version=11, input_names=["input.1"], output_names=["683"])
    blob = f.getvalue()
    assert len(blob) > 150e6                                     # ~43.6 M parameters
    engine.graph_load(3, blob)
    info = engine.graph_info(3)
    (emb,) = engine.graph_run(3, x.numpy())
    engine.graph_unload(3)
    assert emb.shape == (3, 512) and np.isfinite(emb).all()
    assert float(np.abs(emb - ref).max()) <= 1e-3 * max(1.0, float(np.abs(ref).max())), float(np.abs(emb - ref).max())
    cos = (emb * ref).sum(1) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(ref, axis=1))
    assert (cos > 0.999999).all() and info is not None

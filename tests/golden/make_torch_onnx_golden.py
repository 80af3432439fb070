"""Generates tests/golden/torch_onnx_*.npz: small torch modules shaped like the three buffalo_l networks, serialised to .onnx by
PyTorch's OWN TorchScript exporter (its C++ protobuf writer, i.e. an encoder this repo did not write: raw_data tensors, Constant
nodes, Resize with empty roi / scales inputs, Reshape shapes as int64 initialisers, Gemm with transB, BatchNormalization
attributes ...), together with the input and the outputs torch computes for it.

Why: the product's ONNX reader / graph runtime and the oracle's evaluator are otherwise only exercised on files written by
facet_amd/onnx_writer.py. These fixtures hold both to a mainstream exporter's encoding and to torch's numerics
(tests/test_onnx_host.py for reader + oracle on the CPU, tests/test_graph_gpu.py for the engine).

The `onnx` Python package is absent; the exporter only needs it to splice onnxscript functions into the finished bytes, which
these modules do not use - that one helper is bypassed below. Run in the build container:

    python tests/golden/make_torch_onnx_golden.py
"""
import io
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

warnings.filterwarnings("ignore")
onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes   # no onnxscript functions to add


class IBlock(nn.Module):
    """IResNet block (ArcFace): BN -> conv3x3 -> BN -> PReLU -> conv3x3(stride) -> BN, plus a conv1x1+BN shortcut when needed."""
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.bn1, self.conv1, self.bn2 = nn.BatchNorm2d(cin), nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.prelu, self.conv2, self.bn3 = nn.PReLU(cout), nn.Conv2d(cout, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout)
        self.down = None if (stride == 1 and cin == cout) else nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.bn3(self.conv2(self.prelu(self.bn2(self.conv1(self.bn1(x))))))
        return y + (x if self.down is None else self.down(x))


class ArcLike(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 8, 3, 1, 1, bias=False), nn.BatchNorm2d(8), nn.PReLU(8))
        self.body = nn.Sequential(IBlock(8, 8, 2), IBlock(8, 16, 2), IBlock(16, 16, 1), IBlock(16, 32, 2), IBlock(32, 48, 2))
        self.bn2, self.fc, self.feat = nn.BatchNorm2d(48), nn.Linear(48 * 4 * 4, 40), nn.BatchNorm1d(40)

    def forward(self, x):
        x = self.bn2(self.body(self.stem(x)))
        return self.feat(self.fc(torch.flatten(x, 1)))


class DetLike(nn.Module):
    """SCRFD-style: residual backbone, top-down FPN with nearest upsampling, shared heads, outputs flattened to [anchors, C]."""
    def __init__(self):
        super().__init__()
        cbr = lambda i, o, s: nn.Sequential(nn.Conv2d(i, o, 3, s, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU())   # noqa: E731
        self.stem = nn.Sequential(cbr(3, 16, 2), cbr(16, 16, 1), nn.MaxPool2d(2, 2))
        self.s8, self.s16, self.s32 = cbr(16, 32, 2), cbr(32, 48, 2), cbr(48, 64, 2)
        self.r8 = nn.Sequential(nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32))
        self.lat = nn.ModuleList([nn.Conv2d(c, 24, 1) for c in (32, 48, 64)])
        self.smooth = nn.ModuleList([nn.Conv2d(24, 24, 3, 1, 1) for _ in range(3)])
        self.tower = nn.Sequential(nn.Conv2d(24, 24, 3, 1, 1), nn.ReLU(), nn.Conv2d(24, 24, 3, 1, 1), nn.ReLU())
        self.cls, self.box, self.kps = nn.Conv2d(24, 2, 3, 1, 1), nn.Conv2d(24, 8, 3, 1, 1), nn.Conv2d(24, 20, 3, 1, 1)
        self.scales = nn.Parameter(torch.tensor([1.1, 0.9, 1.3]))

    def forward(self, x):
        c8 = self.s8(self.stem(x))
        c8 = F.relu(c8 + self.r8(c8))
        c16 = self.s16(c8)
        c32 = self.s32(c16)
        p32 = self.lat[2](c32)
        p16 = self.lat[1](c16) + F.interpolate(p32, scale_factor=2.0, mode="nearest")
        p8 = self.lat[0](c8) + F.interpolate(p16, scale_factor=2.0, mode="nearest")
        feats = [self.smooth[0](p8), self.smooth[1](p16), self.smooth[2](p32)]
        scores, boxes, kps = [], [], []
        for i, f in enumerate(feats):
            t = self.tower(f)
            scores.append(self.cls(t).permute(0, 2, 3, 1).reshape(-1, 1).sigmoid())
            boxes.append((self.box(t) * self.scales[i]).permute(0, 2, 3, 1).reshape(-1, 4))
            kps.append(self.kps(t).permute(0, 2, 3, 1).reshape(-1, 10))
        return tuple(scores + boxes + kps)


class DetDynamic(DetLike):
    """The same detector written the way mmdet-style code is: upsampling to the size of the lateral map (`size=x.shape[-2:]`), exported
    with dynamic batch / height / width - the file then carries Shape / Gather / Unsqueeze / Concat / Slice / Cast chains in front
    of every Resize, which a runtime has to fold on the host for the actual input size."""
    def forward(self, x):
        c8 = self.s8(self.stem(x))
        c8 = F.relu(c8 + self.r8(c8))
        c16 = self.s16(c8)
        c32 = self.s32(c16)
        p32 = self.lat[2](c32)
        p16 = self.lat[1](c16) + F.interpolate(p32, size=c16.shape[-2:], mode="nearest")
        p8 = self.lat[0](c8) + F.interpolate(p16, size=c8.shape[-2:], mode="nearest")
        outs = []
        for i, f in enumerate([self.smooth[0](p8), self.smooth[1](p16), self.smooth[2](p32)]):
            t = self.tower(f)
            outs.append(self.cls(t).permute(0, 2, 3, 1).reshape(-1, 1).sigmoid())
            outs.append((self.box(t) * self.scales[i]).permute(0, 2, 3, 1).reshape(-1, 4))
        return tuple(outs)


class LmkLike(nn.Module):
    """MobileNet-style landmark regressor: depthwise-separable blocks with PReLU, global average pooling, FC."""
    def __init__(self):
        super().__init__()
        def dw(i, o, s):
            return nn.Sequential(nn.Conv2d(i, i, 3, s, 1, groups=i, bias=False), nn.BatchNorm2d(i), nn.PReLU(i),
                                 nn.Conv2d(i, o, 1, bias=False), nn.BatchNorm2d(o), nn.PReLU(o))
        self.stem = nn.Sequential(nn.Conv2d(3, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.PReLU(16))
        self.body = nn.Sequential(dw(16, 32, 1), dw(32, 32, 2), dw(32, 64, 2), dw(64, 64, 1), dw(64, 128, 2))
        self.fc = nn.Linear(128, 212)

    def forward(self, x):
        x = self.body(self.stem(x))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


def randomise(m, g):
    """Non-trivial BatchNorm statistics / PReLU slopes (fresh modules have mean 0, var 1, slope 0.25 everywhere)."""
    for mod in m.modules():
        if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.2)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
        elif isinstance(mod, nn.PReLU):
            mod.weight.data.copy_(torch.rand(mod.num_parameters, generator=g) * 0.4)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    cases = [("arc", ArcLike, (2, 3, 64, 64), 11), ("det", DetLike, (1, 3, 64, 96), 11),
             ("det", DetLike, (1, 3, 64, 96), 13), ("lmk", LmkLike, (2, 3, 48, 48), 11), ("detdyn", DetDynamic, (1, 3, 64, 96), 11)]
    for name, cls, shape, opset in cases:
        torch.manual_seed({"arc": 1, "det": 2, "lmk": 3, "detdyn": 2}[name])
        g = torch.Generator().manual_seed(11)
        m = cls().eval()
        with torch.no_grad():
            randomise(m, g)
            x = torch.rand(shape, generator=g) * 2 - 1
            ref = m(x)
            ref = ref if isinstance(ref, tuple) else (ref,)
            f = io.BytesIO()
            dyn = {"input": {0: "n", 2: "h", 3: "w"}} if name == "detdyn" else None
            torch.onnx.export(m, (x,), f, dynamo=False, opset_version=opset, input_names=["input"],
                              output_names=[f"out{i}" for i in range(len(ref))], dynamic_axes=dyn)
            extra = {}
            if dyn:                                    # a second input of another size through the same file
                x2 = torch.rand((1, 3, 96, 128), generator=g) * 2 - 1
                extra = {"x2": x2.numpy(), **{f"z{i}": r.numpy() for i, r in enumerate(m(x2))}}
        blob = np.frombuffer(f.getvalue(), np.uint8)
        path = os.path.join(here, f"torch_onnx_{name}_opset{opset}.npz")
        np.savez_compressed(path, onnx=blob, x=x.numpy(), **{f"y{i}": r.numpy() for i, r in enumerate(ref)}, **extra)
        print(f"{os.path.basename(path)}: model {blob.size} B, file {os.path.getsize(path)} B, outputs {[tuple(r.shape) for r in ref]}")


if __name__ == "__main__":
    main()

"""Seeded stand-ins for the three buffalo_l ONNX models the reference loads through insightface (analyzers/face.py:30-38).

The real files (det_10g.onnx = SCRFD-10GF with keypoints, 2d106det.onnx = 106-point landmark regressor,
w600k_r50.onnx = ArcFace IResNet-50) are downloaded by insightface at run time and are not available offline, so tests
and benchmarks build graphs of the same public architectures [DEP-KNOWLEDGE: insightface model zoo / arcface_torch
iresnet.py / mmdet SCRFD configs] with seeded weights, written as genuine .onnx bytes by `standins.onnx_writer`. The
engine treats them exactly like the real files; with real files on disk, pass those bytes instead.

Every builder returns (onnx_bytes, info) where info carries the input size, the MACs per image and layout facts.
"""
import numpy as np

from .onnx_writer import GraphBuilder


def arcface_iresnet(layers=(3, 4, 14, 3), seed=11, size=112, explicit_bn=False, emb=512):
    """ArcFace backbone (IResNet): conv-bn-prelu stem, IBasicBlocks (bn, conv3x3, bn, prelu, conv3x3 stride, bn, +identity),
    bn - flatten - fc - bn1d. `explicit_bn=False` mirrors torch.onnx.export's eval-mode Conv+BN folding (the conv carries a
    bias); `explicit_bn=True` keeps Conv -> BatchNormalization nodes so the engine's own folding is exercised."""
    g = GraphBuilder(seed)
    x = "input.1"
    hw = size
    macs = 0

    def conv_bn(x, cin, cout, k, s, hw_out, gain=1.0):
        nonlocal macs
        macs += hw_out * hw_out * cout * cin * k * k
        if explicit_bn:
            return g.bn(g.conv(x, cin, cout, k, s, bias=False, gain=gain), cout)
        return g.conv(x, cin, cout, k, s, bias=True, gain=gain)

    x = g.prelu(conv_bn(x, 3, 64, 3, 1, hw), 64)
    inpl = 64
    for planes, nblk in zip((64, 128, 256, 512), layers):
        for b in range(nblk):
            stride = 2 if b == 0 else 1
            out = g.bn(x, inpl)
            out = g.prelu(conv_bn(out, inpl, planes, 3, 1, hw), planes)
            hw_o = hw // stride
            out = conv_bn(out, planes, planes, 3, stride, hw_o, gain=0.5)
            identity = conv_bn(x, inpl, planes, 1, stride, hw_o) if b == 0 else x
            x = g.add(out, identity)
            hw, inpl = hw_o, planes
    x = g.bn(x, 512)
    x = g.op("Flatten", [x], axis=1)
    x = g.gemm(x, 512 * hw * hw, emb)
    macs += 512 * hw * hw * emb
    x = g.bn(x, emb)
    data = g.build([("input.1", ["N", 3, size, size])], [(x, ["N", emb])], producer="pytorch")
    return data, {"input": (3, size, size), "macs": macs, "outputs": [x], "mean": 127.5, "std": 127.5}


def scrfd_like(seed=12, size=640, stem=28, widths=(56, 88, 88, 224), blocks=(3, 4, 2, 3), neck=56, feat=80, kps=5, cls_gain=0.01, cls_bias=-0.3):
    """SCRFD-style detector: ResNet-V1e-like backbone (3-conv stem, BasicBlocks, avg-pool shortcuts), PAFPN neck on /8 /16 /32,
    per-stride heads of 3 conv+relu then cls (2 anchors x 1, sigmoid), bbox (2 x 4, learned scale) and keypoint (2 x 10)
    3x3 convs, each flattened by Transpose(0,2,3,1) + Reshape like the exported det_10g.onnx. Outputs are ordered
    score_8, score_16, score_32, bbox_8, .., kps_8, .. as insightface's SCRFD wrapper expects (fmc = 3, use_kps)."""
    g = GraphBuilder(seed)
    macs = 0
    hw = size

    def conv(x, cin, cout, k, s, hw_out, relu=True, gain=1.0, bias_shift=0.0):
        nonlocal macs
        macs += hw_out * hw_out * cout * cin * k * k
        y = g.conv(x, cin, cout, k, s, bias=True, gain=gain, bias_shift=bias_shift)
        return g.relu(y) if relu else y

    x = "input.1"
    hw //= 2
    x = conv(x, 3, stem, 3, 2, hw)
    x = conv(x, stem, stem, 3, 1, hw)
    x = conv(x, stem, widths[0], 3, 1, hw)
    x = g.op("MaxPool", [x], kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1])
    hw //= 2
    feats = []
    inpl = widths[0]
    for si, (planes, nblk) in enumerate(zip(widths, blocks)):
        for b in range(nblk):
            stride = 2 if (b == 0 and si > 0) else 1
            hw_o = hw // stride
            out = conv(x, inpl, planes, 3, stride, hw_o)
            out = conv(out, planes, planes, 3, 1, hw_o, relu=False, gain=0.5)
            identity = x
            if stride != 1 or inpl != planes:
                if stride != 1:
                    identity = g.op("AveragePool", [identity], kernel_shape=[2, 2], strides=[2, 2], pads=[0, 0, 0, 0], ceil_mode=1,
                                    count_include_pad=0)
                identity = conv(identity, inpl, planes, 1, 1, hw_o, relu=False)
            x = g.relu(g.add(out, identity))
            hw, inpl = hw_o, planes
        feats.append((x, planes, hw))
    c3, c4, c5 = feats[1], feats[2], feats[3]
    lat = [conv(f, c, neck, 1, 1, s, relu=False) for f, c, s in (c3, c4, c5)]
    sizes = [c3[2], c4[2], c5[2]]
    roi = g.const(np.zeros((0,), np.float32), "roi")
    sc2 = g.const(np.asarray([1, 1, 2, 2], np.float32), "scales")
    for i in (2, 1):
        up = g.op("Resize", [lat[i], roi, sc2], mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")
        lat[i - 1] = g.add(lat[i - 1], up)
    fpn = [conv(lat[i], neck, neck, 3, 1, sizes[i], relu=False) for i in range(3)]
    for i in (0, 1):
        down = conv(fpn[i], neck, neck, 3, 2, sizes[i + 1], relu=False)
        fpn[i + 1] = g.add(fpn[i + 1], down)
    outs = [fpn[0]] + [conv(fpn[i], neck, neck, 3, 1, sizes[i], relu=False) for i in (1, 2)]
    scores, boxes, kpss = [], [], []
    for i, stride in enumerate((8, 16, 32)):
        t = outs[i]
        cin = neck
        for _ in range(3):
            t = conv(t, cin, feat, 3, 1, sizes[i])
            cin = feat
        cls = conv(t, feat, 2, 3, 1, sizes[i], relu=False, gain=cls_gain, bias_shift=cls_bias)   # small logits: few anchors pass 0.5, no saturated ties
        reg = conv(t, feat, 8, 3, 1, sizes[i], relu=False, gain=0.3)
        reg = g.op("Mul", [reg, g.const(np.asarray(1.0 + 0.1 * i, np.float32), "scale")])
        kp = conv(t, feat, 4 * kps, 3, 1, sizes[i], relu=False, gain=0.3)
        for src, width, dst in ((cls, 1, scores), (reg, 4, boxes), (kp, 2 * kps, kpss)):
            v = g.op("Transpose", [src], perm=[0, 2, 3, 1])
            v = g.op("Reshape", [v, g.const(np.asarray([-1, width], np.int64), "shape")])
            if dst is scores:
                v = g.op("Sigmoid", [v])
            dst.append((v, width, sizes[i]))
    out_list = scores + boxes + kpss
    data = g.build([("input.1", [1, 3, size, size])], [(v, [s * s * 2, wd]) for v, wd, s in out_list], producer="pytorch")
    return data, {"input": (3, size, size), "macs": macs, "outputs": [v for v, _, _ in out_list], "strides": (8, 16, 32), "anchors": 2}


def landmark_like(seed=13, size=192, points=106):
    """2d106det-style regressor: in-graph (x - 127.5) * 0.0078125 (named like MXNet's _minusscalar/_mulscalar, which is how
    insightface concludes mean 0 / std 1), MobileNet-v1-style depthwise-separable stack with explicit BatchNormalization and
    PRelu nodes (MXNet exports do not fold them), fully-connected to 2*points outputs."""
    g = GraphBuilder(seed)
    macs = 0
    x = g.op("Sub", ["data", g.const(np.asarray(127.5, np.float32), "c")], name="_minusscalar0")
    x = g.op("Mul", [x, g.const(np.asarray(0.0078125, np.float32), "c")], name="_mulscalar0")
    hw = size // 2

    def cbp(x, cin, cout, k, s, hw_out, group=1):
        nonlocal macs
        macs += hw_out * hw_out * cout * (cin // group) * k * k
        return g.prelu(g.bn(g.conv(x, cin, cout, k, s, bias=False, group=group), cout), cout)

    x = cbp(x, 3, 16, 3, 2, hw)
    cin = 16
    for cout, s in ((32, 1), (64, 2), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (256, 1), (512, 2), (512, 1), (512, 2)):
        hw_o = hw // s
        x = cbp(x, cin, cin, 3, s, hw_o, group=cin)
        x = cbp(x, cin, cout, 1, 1, hw_o)
        hw, cin = hw_o, cout
    x = g.op("Flatten", [x], axis=1)
    x = g.gemm(x, cin * hw * hw, 2 * points)
    macs += cin * hw * hw * 2 * points
    data = g.build([("data", [1, 3, size, size])], [(x, [1, 2 * points])], producer="mxnet-like")
    return data, {"input": (3, size, size), "macs": macs, "outputs": [x], "mean": 0.0, "std": 1.0}

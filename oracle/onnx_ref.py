"""TEST INFRASTRUCTURE ONLY - CPU oracle for the engine's ONNX-subset graph runtime (facet_amd/csrc/onnx_graph.hip).

Reads an .onnx file with its own small protobuf reader (independent of the C++ one under test) and evaluates the graph node by
node with torch CPU ops in NCHW fp32, following the ONNX operator specifications (the semantics onnxruntime implements for the
reference's InsightFace sessions, analyzers/face.py:30-38,99). Parity status: UNPINNED against onnxruntime itself - neither
onnxruntime nor the buffalo_l model files exist offline; what this pins is operator semantics as published in the ONNX spec.
"""
import struct

import numpy as np
import torch
import torch.nn.functional as F


# ---- protobuf reader ---------------------------------------------------------------------------------------------
def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key = 0
        sh = 0
        while True:
            b = buf[i]
            i += 1
            key |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        f, wt = key >> 3, key & 7
        if wt == 0:
            v = 0
            sh = 0
            while True:
                b = buf[i]
                i += 1
                v |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            yield f, wt, v
        elif wt == 1:
            yield f, wt, buf[i:i + 8]
            i += 8
        elif wt == 2:
            ln = 0
            sh = 0
            while True:
                b = buf[i]
                i += 1
                ln |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            yield f, wt, buf[i:i + ln]
            i += ln
        elif wt == 5:
            yield f, wt, buf[i:i + 4]
            i += 4
        else:
            raise ValueError(f"wire type {wt}")


def _s64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(wt, v):
    if wt == 0:
        return [_s64(v)]
    return [_s64(x) for _, _, x in _fields_varints(v)]


def _fields_varints(buf):
    i, n = 0, len(buf)
    while i < n:
        v = 0
        sh = 0
        while True:
            b = buf[i]
            i += 1
            v |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        yield 0, 0, v


_DT = {1: "<f4", 2: "u1", 3: "i1", 6: "<i4", 7: "<i8", 9: "u1", 10: "<f2", 11: "<f8"}


def _tensor(buf):
    dims, dtype, name, raw = [], 1, "", None
    fl, il, dl = [], [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            dims += _packed_ints(wt, v)
        elif f == 2:
            dtype = v
        elif f == 4:
            fl += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f in (5, 7):
            il += _packed_ints(wt, v)
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 10:
            dl += list(struct.unpack(f"<{len(v) // 8}d", v)) if wt == 2 else [struct.unpack("<d", v)[0]]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=_DT[dtype]).reshape(dims)
    elif dtype == 1:
        arr = np.asarray(fl, np.float32).reshape(dims)
    elif dtype == 10:
        arr = np.asarray(il, np.uint16).view(np.float16).reshape(dims)
    elif dtype == 11:
        arr = np.asarray(dl, np.float64).reshape(dims)
    else:
        arr = np.asarray(il, np.int64).reshape(dims)
    return name, np.array(arr)


def _attr(buf):
    name, a = "", {}
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            a["f"] = struct.unpack("<f", v)[0]
        elif f == 3:
            a["i"] = _s64(v)
        elif f == 4:
            a["s"] = bytes(v).decode()
        elif f == 5:
            a["t"] = _tensor(v)[1]
        elif f == 7:
            a.setdefault("floats", [])
            a["floats"] += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            a.setdefault("ints", [])
            a["ints"] += _packed_ints(wt, v)
        elif f == 20:
            a["type"] = v
    t = a.get("type")
    val = {1: a.get("f", 0.0), 2: a.get("i", 0), 3: a.get("s", ""), 4: a.get("t"), 6: a.get("floats", []), 7: a.get("ints", [])}.get(t)
    if t is None:   # writers that omit the type tag
        val = next((a[k] for k in ("ints", "floats", "t", "s", "i", "f") if k in a), None)
    return name, val


def parse(onnx_bytes):
    graph = None
    opset = 0
    for f, wt, v in _fields(memoryview(onnx_bytes)):
        if f == 7:
            graph = v
        elif f == 8:
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode()
                elif f2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                opset = ver
    nodes, init, inputs, outputs = [], {}, [], []
    for f, wt, v in _fields(graph):
        if f == 1:
            n = {"in": [], "out": [], "attr": {}, "op": "", "name": ""}
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    n["in"].append(bytes(v2).decode())
                elif f2 == 2:
                    n["out"].append(bytes(v2).decode())
                elif f2 == 3:
                    n["name"] = bytes(v2).decode()
                elif f2 == 4:
                    n["op"] = bytes(v2).decode()
                elif f2 == 5:
                    k, val = _attr(v2)
                    n["attr"][k] = val
            nodes.append(n)
        elif f == 5:
            name, arr = _tensor(v)
            init[name] = arr
        elif f in (11, 12):
            name = ""
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    name = bytes(v2).decode()
            (inputs if f == 11 else outputs).append(name)
    inputs = [n for n in inputs if n not in init]
    return {"nodes": nodes, "init": init, "inputs": inputs, "outputs": outputs, "opset": opset}


# ---- evaluation ------------------------------------------------------------------------------------------------------
def _t(v):
    if isinstance(v, torch.Tensor):
        return v
    a = np.asarray(v)
    return torch.from_numpy(a.astype(np.float32) if a.dtype.kind == "f" else a.astype(np.int64))


def run(model, x):
    """model: parse() result; x: float32 array/tensor [N,C,H,W]. Returns the list of outputs as numpy arrays."""
    if isinstance(model, (bytes, bytearray, memoryview)):
        model = parse(model)
    env = {k: _t(v) for k, v in model["init"].items()}
    env[model["inputs"][0]] = _t(x).float()
    opset = model["opset"]
    with torch.no_grad():
        for n in model["nodes"]:
            op, a = n["op"], n["attr"]
            i = [env[s] if s else None for s in n["in"]]
            if op == "Conv":
                pads = a.get("pads", [0, 0, 0, 0])
                assert pads[0] == pads[2] and pads[1] == pads[3]
                y = F.conv2d(i[0], i[1].float(), i[2].float() if len(i) > 2 and i[2] is not None else None, stride=tuple(a.get("strides", [1, 1])),
                             padding=(pads[0], pads[1]), dilation=tuple(a.get("dilations", [1, 1])), groups=a.get("group", 1))
            elif op == "BatchNormalization":
                y = F.batch_norm(i[0], i[3].float(), i[4].float(), i[1].float(), i[2].float(), False, 0.0, a.get("epsilon", 1e-5))
            elif op == "Relu":
                y = F.relu(i[0])
            elif op == "PRelu":
                s = i[1].float()
                y = torch.where(i[0] > 0, i[0], i[0] * (s if s.dim() != 1 or i[0].dim() < 3 else s.view(-1, 1, 1)))
            elif op == "LeakyRelu":
                y = F.leaky_relu(i[0], a.get("alpha", 0.01))
            elif op == "Sigmoid":
                y = torch.sigmoid(i[0])
            elif op in ("Add", "Sub", "Mul", "Div"):
                p, q = i[0], i[1]
                if op == "Div" and not p.is_floating_point() and not q.is_floating_point():
                    y = torch.div(p, q, rounding_mode="trunc")
                else:
                    y = {"Add": torch.add, "Sub": torch.sub, "Mul": torch.mul, "Div": torch.div}[op](p, q)
            elif op in ("MaxPool", "AveragePool"):
                k, s, p = a["kernel_shape"], a.get("strides", [1, 1]), a.get("pads", [0, 0, 0, 0])
                cm = bool(a.get("ceil_mode", 0))
                if op == "MaxPool":
                    y = F.max_pool2d(i[0], tuple(k), tuple(s), (p[0], p[1]), ceil_mode=cm)
                else:
                    y = F.avg_pool2d(i[0], tuple(k), tuple(s), (p[0], p[1]), ceil_mode=cm, count_include_pad=bool(a.get("count_include_pad", 0)))
            elif op == "GlobalAveragePool":
                y = i[0].mean(dim=(2, 3), keepdim=True)
            elif op in ("Resize", "Upsample"):
                old = op == "Upsample" or len(i) == 2
                scales = i[1] if old else (i[2] if len(i) > 2 and i[2] is not None and i[2].numel() else None)
                sizes = i[3] if (not old and len(i) > 3 and i[3] is not None) else None
                if sizes is not None:
                    oh, ow = int(sizes[2]), int(sizes[3])
                else:
                    oh, ow = int(np.floor(i[0].shape[2] * float(scales[2]))), int(np.floor(i[0].shape[3] * float(scales[3])))
                mode = a.get("mode", "nearest")
                if mode == "nearest":     # asymmetric + floor
                    ys = torch.floor(torch.arange(oh) * (i[0].shape[2] / oh)).long().clamp(max=i[0].shape[2] - 1)
                    xs = torch.floor(torch.arange(ow) * (i[0].shape[3] / ow)).long().clamp(max=i[0].shape[3] - 1)
                    y = i[0][:, :, ys][:, :, :, xs]
                else:
                    y = F.interpolate(i[0], size=(oh, ow), mode="bilinear", align_corners=False)
            elif op == "Concat":
                y = torch.cat(i, dim=a.get("axis", 1))
            elif op == "Flatten":
                y = i[0].flatten(a.get("axis", 1))
            elif op == "Gemm":
                w = i[1].float()
                y = i[0] @ (w.t() if a.get("transB", 0) else w)
                if len(i) > 2 and i[2] is not None:
                    y = y + i[2].float()
            elif op == "MatMul":
                y = i[0] @ i[1].float()
            elif op == "Transpose":
                perm = a.get("perm") or list(range(i[0].dim() - 1, -1, -1))
                y = i[0].permute(*perm).contiguous()
            elif op == "Reshape":
                shape = [int(v) for v in (i[1].tolist() if len(i) > 1 else a["shape"])]
                shape = [i[0].shape[k] if d == 0 else d for k, d in enumerate(shape)]
                y = i[0].reshape(shape)
            elif op == "Squeeze":
                axes = a.get("axes") or (i[1].tolist() if len(i) > 1 else None)
                y = i[0]
                for ax in sorted(axes or [k for k, d in enumerate(y.shape) if d == 1], reverse=True):
                    y = y.squeeze(ax)
            elif op == "Unsqueeze":
                axes = a.get("axes") or i[1].tolist()
                y = i[0]
                for ax in sorted(axes):
                    y = y.unsqueeze(ax)
            elif op == "Softmax":
                y = torch.softmax(i[0], dim=a.get("axis", -1 if opset >= 13 else 1))
            elif op in ("Identity", "Dropout"):
                y = i[0]
            elif op == "Constant":
                y = _t(a["value"])
            elif op == "Shape":
                y = torch.tensor(list(i[0].shape), dtype=torch.int64)
            elif op == "Gather":
                y = torch.index_select(i[0], a.get("axis", 0), i[1].reshape(-1).long()).reshape(
                    list(i[0].shape[:a.get("axis", 0)]) + list(i[1].shape) + list(i[0].shape[a.get("axis", 0) + 1:]))
            elif op == "Cast":
                y = i[0].long() if a["to"] in (2, 3, 6, 7, 9) else i[0].float()
            elif op == "Floor":
                y = torch.floor(i[0])
            elif op == "Ceil":
                y = torch.ceil(i[0])
            elif op == "Slice":
                st, en = (int(i[1][0]), int(i[2][0])) if len(i) > 2 else (a["starts"][0], a["ends"][0])
                y = i[0][st:en]
            else:
                raise NotImplementedError(op)
            env[n["out"][0]] = y
    return [env[o].numpy() for o in model["outputs"]]

"""Minimal ONNX (protobuf wire format) writer: just enough of onnx.ModelProto to emit the graphs the engine's runtime reads.

There is no `onnx` package in this environment, and the reference's face models (buffalo_l: det_10g.onnx, 2d106det.onnx,
w600k_r50.onnx, fetched by insightface at analyzers/face.py:30-38) are not on disk; `standins.synthetic_onnx` uses this
writer to build seeded stand-ins of the same architectures for tests and benchmarks. Field numbers follow the public
onnx.proto3 schema; tensors are written as little-endian raw_data, like torch.onnx.export does.
"""
import struct

import numpy as np

FLOAT, UINT8, INT8, INT32, INT64, BOOL, FLOAT16, DOUBLE = 1, 2, 3, 6, 7, 9, 10, 11
_NP2ONNX = {np.dtype(np.float32): FLOAT, np.dtype(np.int64): INT64, np.dtype(np.int32): INT32, np.dtype(np.float16): FLOAT16,
            np.dtype(np.float64): DOUBLE, np.dtype(np.uint8): UINT8, np.dtype(np.int8): INT8, np.dtype(np.bool_): BOOL}


def _varint(n):
    n &= (1 << 64) - 1          # negative int64 -> two's complement, 10 bytes
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wt):
    return _varint((field << 3) | wt)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _str(field, s):
    return _ld(field, s.encode("utf-8"))


def _int(field, v):
    return _key(field, 0) + _varint(int(v))


def tensor_proto(name, arr, encoding="raw"):
    """encoding: "raw" (raw_data, what torch.onnx.export writes) or "typed" (float_data / int64_data / int32_data /
    double_data repeated fields, what some converters write; float16 goes into int32_data as bit patterns)."""
    arr = np.asarray(arr)
    shape = arr.shape                      # ascontiguousarray would turn a 0-d scalar into shape (1,)
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in _NP2ONNX:
        raise TypeError(f"unsupported dtype {arr.dtype}")
    out = b"".join(_int(1, d) for d in shape)
    out += _int(2, _NP2ONNX[arr.dtype])
    out += _str(8, name)
    if encoding == "raw":
        out += _ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    elif arr.dtype == np.float32:
        out += _ld(4, arr.astype("<f4").tobytes())                                   # packed float_data
    elif arr.dtype == np.float64:
        out += _ld(10, arr.astype("<f8").tobytes())                                  # packed double_data
    elif arr.dtype == np.int64:
        out += _ld(7, b"".join(_varint(int(v)) for v in arr.ravel()))                 # packed int64_data
    elif arr.dtype == np.float16:
        out += _ld(5, b"".join(_varint(int(v)) for v in arr.view(np.uint16).ravel()))
    else:
        out += _ld(5, b"".join(_varint(int(v)) for v in arr.ravel()))                 # int32_data (also int8/uint8/bool)
    return out


def attribute(name, value):
    out = _str(1, name)
    if isinstance(value, bool):
        value = int(value)
    if isinstance(value, (int, np.integer)):
        out += _int(3, value) + _int(20, 2)
    elif isinstance(value, (float, np.floating)):
        out += _key(2, 5) + struct.pack("<f", float(value)) + _int(20, 1)
    elif isinstance(value, str):
        out += _ld(4, value.encode("utf-8")) + _int(20, 3)
    elif isinstance(value, np.ndarray):
        out += _ld(5, tensor_proto("", value)) + _int(20, 4)
    elif isinstance(value, (list, tuple)) and all(isinstance(v, (int, np.integer)) for v in value):
        out += _ld(8, b"".join(_varint(int(v)) for v in value)) + _int(20, 7)      # packed ints
    elif isinstance(value, (list, tuple)):
        out += _ld(7, b"".join(struct.pack("<f", float(v)) for v in value)) + _int(20, 6)
    else:
        raise TypeError(f"unsupported attribute value {value!r}")
    return out


def node(op, inputs, outputs, name="", **attrs):
    out = b"".join(_str(1, s) for s in inputs) + b"".join(_str(2, s) for s in outputs)
    out += _str(3, name) + _str(4, op)
    out += b"".join(_ld(5, attribute(k, v)) for k, v in attrs.items())
    return out


def value_info(name, dims, elem_type=FLOAT):
    shape = b""
    for d in dims:
        dim = _str(2, d) if isinstance(d, str) else _int(1, d)
        shape += _ld(1, dim)
    tensor_type = _int(1, elem_type) + _ld(2, shape)
    return _str(1, name) + _ld(2, _ld(1, tensor_type))


def model(nodes, initializers, inputs, outputs, opset=11, producer="standins.onnx_writer", graph_name="g", encoding="raw",
          list_initializers_as_inputs=False):
    """nodes: list of node() payloads; initializers: {name: ndarray}; inputs/outputs: [(name, dims)].
    list_initializers_as_inputs: IR < 4 files also declare every initializer as a graph input."""
    g = b"".join(_ld(1, n) for n in nodes) + _str(2, graph_name)
    g += b"".join(_ld(5, tensor_proto(k, v, encoding)) for k, v in initializers.items())
    if list_initializers_as_inputs:
        inputs = list(inputs) + [(k, list(np.asarray(v).shape)) for k, v in initializers.items()]
    g += b"".join(_ld(11, value_info(n, d)) for n, d in inputs)
    g += b"".join(_ld(12, value_info(n, d)) for n, d in outputs)
    m = _int(1, 6) + _str(2, producer) + _ld(7, g) + _ld(8, _str(1, "") + _int(2, opset))
    return m


class GraphBuilder:
    """Tiny convenience layer: keeps the node list, the initializers and fresh value names."""

    def __init__(self, seed=0, opset=11):
        self.nodes, self.init, self.opset = [], {}, opset
        self.rng = np.random.default_rng(seed)
        self._n = 0
        self.macs = 0     # multiply-accumulates per image, filled by conv/gemm with spatial sizes the caller passes

    def fresh(self, stem="v"):
        self._n += 1
        return f"{stem}_{self._n}"

    def const(self, arr, stem="c"):
        name = self.fresh(stem)
        self.init[name] = np.asarray(arr)
        return name

    def op(self, op, inputs, n_out=1, name=None, **attrs):
        outs = [self.fresh(op.lower())] if n_out == 1 else [self.fresh(op.lower()) for _ in range(n_out)]
        self.nodes.append(node(op, inputs, outs, name or f"{op}_{len(self.nodes)}", **attrs))
        return outs[0] if n_out == 1 else outs

    # -- layers with seeded parameters ------------------------------------------------------------------------
    def conv(self, x, cin, cout, k=3, s=1, p=None, bias=True, group=1, dilation=1, gain=1.0, bias_shift=0.0):
        p = (k // 2) * dilation if p is None else p
        fan_in = (cin // group) * k * k
        w = (self.rng.standard_normal((cout, cin // group, k, k)) * (gain * np.sqrt(2.0 / fan_in))).astype(np.float32)
        ins = [x, self.const(w, "w")]
        if bias:
            ins.append(self.const((self.rng.standard_normal(cout) * 0.05 + bias_shift).astype(np.float32), "b"))
        return self.op("Conv", ins, kernel_shape=[k, k], strides=[s, s], pads=[p, p, p, p], dilations=[dilation, dilation],
                       group=group)

    def bn(self, x, c, eps=1e-5):
        r = self.rng
        ps = [self.const(r.uniform(0.6, 1.2, c).astype(np.float32), "g"), self.const((r.standard_normal(c) * 0.1).astype(np.float32), "beta"),
              self.const((r.standard_normal(c) * 0.1).astype(np.float32), "mu"), self.const(r.uniform(0.5, 1.5, c).astype(np.float32), "var")]
        return self.op("BatchNormalization", [x] + ps, epsilon=float(eps), momentum=0.9)

    def prelu(self, x, c):
        return self.op("PRelu", [x, self.const(self.rng.uniform(0.05, 0.35, (c, 1, 1)).astype(np.float32), "slope")])

    def relu(self, x):
        return self.op("Relu", [x])

    def add(self, a, b):
        return self.op("Add", [a, b])

    def gemm(self, x, cin, cout, bias=True, trans_b=True):
        w = (self.rng.standard_normal((cout, cin) if trans_b else (cin, cout)) * np.sqrt(1.0 / cin)).astype(np.float32)
        ins = [x, self.const(w, "w")]
        if bias:
            ins.append(self.const((self.rng.standard_normal(cout) * 0.05).astype(np.float32), "b"))
        return self.op("Gemm", ins, alpha=1.0, beta=1.0, transB=int(trans_b))

    def build(self, inputs, outputs, producer="standins.onnx_writer", **kw):
        return model(self.nodes, self.init, inputs, outputs, opset=self.opset, producer=producer, **kw)

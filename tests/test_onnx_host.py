"""CPU-side checks of the ONNX path: the writer, the C++ wire reader (fe_onnx_probe needs no GPU) and the oracle's own
reader must agree on the same bytes; malformed files are rejected with a message, not a crash."""
import numpy as np
import pytest

from standins import onnx_writer as W
from standins import synthetic_onnx as S
from facet_amd._lib import EngineError, onnx_probe
from oracle import onnx_ref


def test_probe_matches_oracle_reader():
    for fn, kw in ((S.arcface_iresnet, dict(layers=(1, 1, 1, 1))), (S.scrfd_like, dict(size=64)), (S.landmark_like, {})):
        data, info = fn(**kw)
        p = onnx_probe(data)
        m = onnx_ref.parse(data)
        assert p["nodes"] == len(m["nodes"]) and p["initializers"] == len(m["init"]) and p["outputs"] == len(m["outputs"])
        assert p["input_dims"][1:] == list(info["input"])


def test_writer_roundtrip_values_and_attrs():
    g = W.GraphBuilder(0)
    w = np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2) - 7.5
    n = g.op("Conv", ["x", g.const(w), g.const(np.asarray([-3, 2**40], np.int64))], kernel_shape=[2, 2], strides=[1, 1],
             pads=[0, 0, 0, 0], alpha=0.25, mode="nearest", neg=-5)
    data = g.build([("x", ["N", 3, 8, 8])], [(n, ["N", 2, 7, 7])])
    m = onnx_ref.parse(data)
    assert m["opset"] == 11 and m["inputs"] == ["x"]
    (node,) = m["nodes"]
    assert node["op"] == "Conv" and node["attr"]["kernel_shape"] == [2, 2] and node["attr"]["neg"] == -5
    assert abs(node["attr"]["alpha"] - 0.25) < 1e-7 and node["attr"]["mode"] == "nearest"
    assert np.array_equal(m["init"][node["in"][1]], w)
    assert m["init"][node["in"][2]].tolist() == [-3, 2**40]
    assert onnx_probe(data)["input_dims"] == [-1, 3, 8, 8]


def test_malformed_files_are_rejected():
    data, _ = S.landmark_like()
    for bad in (b"", b"\x00" * 16, data[: len(data) // 2], b"not an onnx file at all"):
        with pytest.raises(EngineError):
            onnx_probe(bad if bad else b"\x00")


def test_oracle_arcface_embedding_is_deterministic():
    data, _ = S.arcface_iresnet(layers=(1, 1, 1, 1), seed=4)
    x = np.random.default_rng(0).uniform(-1, 1, (1, 3, 112, 112)).astype(np.float32)
    a = onnx_ref.run(data, x)[0]
    b = onnx_ref.run(S.arcface_iresnet(layers=(1, 1, 1, 1), seed=4)[0], x)[0]
    assert a.shape == (1, 512) and np.array_equal(a, b) and np.isfinite(a).all()


def test_alternative_encodings_parse_on_host():
    g = W.GraphBuilder(2)
    y = g.gemm(g.op("Flatten", [g.conv("x", 3, 8, 3, 1)], axis=1), 8 * 6 * 6, 4)
    for kw in (dict(encoding="typed"), dict(list_initializers_as_inputs=True)):
        data = g.build([("x", [1, 3, 6, 6])], [(y, [1, 4])], **kw)
        p = onnx_probe(data)
        m = onnx_ref.parse(data)
        assert p["nodes"] == 3 and p["initializers"] == len(m["init"]) == 4 and p["input_dims"] == [1, 3, 6, 6] and m["inputs"] == ["x"]
        for k, v in g.init.items():
            assert np.array_equal(m["init"][k], v)


def test_corrupted_bytes_never_crash_the_reader():
    """Bit flips / truncations of a valid file: the host parser must answer (ok or EngineError), never fault or balloon."""
    g = W.GraphBuilder(5)
    y = g.relu(g.bn(g.conv("x", 3, 8, 3, 1), 8))
    data = bytearray(g.build([("x", [1, 3, 8, 8])], [(y, [1, 8, 8, 8])]))
    rng = np.random.default_rng(0)
    outcomes = {"ok": 0, "rejected": 0}
    for trial in range(300):
        bad = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        if trial % 5 == 0:
            bad = bad[: int(rng.integers(1, len(bad)))]
        try:
            onnx_probe(bytes(bad))
            outcomes["ok"] += 1
        except EngineError:
            outcomes["rejected"] += 1
    assert outcomes["ok"] + outcomes["rejected"] == 300 and outcomes["rejected"] > 0


def test_reader_under_sanitizers_on_mutated_files(tmp_path):
    """The same wire reader (facet_amd/csrc/onnx_parse.cpp is plain C++) built with AddressSanitizer + UBSan: valid files of the three
    stand-in architectures in both encodings parse with the counts the writer put in, and 600 mutated files (byte flips, inserted
    and deleted runs, truncations, spliced length fields) end in "ok" or "error" - never in a sanitizer report."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "facet_amd", "csrc")
    exe = str(tmp_path / "onnx_harness")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", src,
                    os.path.join(root, "tests", "native", "onnx_harness.cpp"), os.path.join(src, "onnx_parse.cpp"), "-o", exe], check=True)
    seeds = {"det": S.scrfd_like(seed=1, size=64)[0], "lmk": S.landmark_like(seed=2)[0], "rec": S.arcface_iresnet(layers=(1, 1, 1, 1), seed=3)[0]}
    g = W.GraphBuilder(5)
    y = g.relu(g.bn(g.conv("x", 3, 8, 3, 1), 8))
    seeds["typed"] = g.build([("x", [1, 3, 8, 8])], [(y, [1, 8, 8, 8])], encoding="typed")
    files = []
    for name, blob in seeds.items():
        p = str(tmp_path / f"{name}.onnx")
        open(p, "wb").write(blob)
        files.append(p)
    out = subprocess.run([exe] + files, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    for line, (name, blob) in zip(out.stdout.splitlines(), seeds.items()):
        m = onnx_ref.parse(blob)
        assert line.split()[:3] == ["ok", str(len(m["nodes"])), str(len(m["init"]))], (name, line)
    rng = np.random.default_rng(1)
    small = [bytearray(seeds["typed"]), bytearray(S.landmark_like(seed=2)[0])]
    files = []
    for t in range(600):
        b = bytearray(small[t % 2])
        kind = t % 6
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 2:
            i = int(rng.integers(0, len(b)))
            b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        elif kind == 3:
            i = int(rng.integers(0, len(b) - 1))
            del b[i:i + int(rng.integers(1, 64))]
        elif kind == 4:                       # a huge varint where a length or a dim may sit
            i = int(rng.integers(0, len(b)))
            b[i:i + 1] = b"\xff\xff\xff\xff\xff\xff\xff\xff\x7f"
        else:
            i, j = sorted(int(v) for v in rng.integers(0, len(b), 2))
            b = b[:i] + b[j:] + b[i:j]
        p = str(tmp_path / f"m{t}.onnx")
        open(p, "wb").write(bytes(b))
        files.append(p)
    out = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 600 and all(l.startswith(("ok ", "error ")) for l in lines) and sum(l.startswith("error") for l in lines) > 100


def _torch_exported():
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "torch_onnx_*.npz")))
    assert len(files) >= 4
    return files


@pytest.mark.parametrize("path", _torch_exported(), ids=lambda p: p.split("torch_onnx_")[-1][:-4])
def test_files_written_by_torchs_exporter(path):
    """.onnx bytes from PyTorch's own exporter (tests/golden/make_torch_onnx_golden.py), i.e. an encoder this repo did not write: the
    engine's reader accepts them and reports what the oracle's reader sees, and the oracle's evaluator reproduces the outputs
    torch computed for the stored input - which pins the oracle's ONNX operator semantics to torch through a mainstream exporter."""
    z = np.load(path)
    blob = z["onnx"].tobytes()
    m = onnx_ref.parse(blob)
    p = onnx_probe(blob)
    assert p["nodes"] == len(m["nodes"]) and p["initializers"] == len(m["init"])
    assert all(d in (-1, s_) for d, s_ in zip(p["input_dims"], z["x"].shape))          # -1 = exported as a dynamic axis
    for xin, pre in (("x", "y"), ("x2", "z")):                                          # x2: another size through a dynamic-shape file
        if xin not in z.files:
            continue
        outs = onnx_ref.run(m, z[xin])
        want = [z[f"{pre}{i}"] for i in range(len(outs))]
        assert f"{pre}{len(outs)}" not in z.files
        for o, w in zip(outs, want):
            o = np.asarray(o)
            assert o.shape == w.shape and float(np.abs(o - w).max()) <= 2e-6 * max(1.0, float(np.abs(w).max()))

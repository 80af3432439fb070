"""CPU: the C-ABI library loads, exports every symbol include/facet_engine.h declares, and the ctypes table
matches the header. No compute calls (there is no GPU here) — and creating a context must fail loudly."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "facet_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from facet_amd._lib import load_library, SIGNATURES
    lib = load_library()
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in facet_engine.h but not exported"
        assert name in SIGNATURES, f"{name} has no ctypes signature in facet_amd/_lib.py"
    for name in SIGNATURES:
        assert name in declared, f"{name} bound in _lib.py but not declared in the header"


def test_version_string():
    from facet_amd._lib import load_library
    assert b"gfx950" in load_library().fe_version()


def test_no_silent_cpu_fallback():
    """Without a gfx950 device fe_create must fail with a message, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from facet_amd import Engine, EngineError
    with pytest.raises(EngineError, match="no HIP device|not .*gfx950|fe_create failed"):
        Engine(0)


def test_product_package_does_not_import_oracle():
    """facet_amd/ (the shipped path) must never import oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "facet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports oracle"


def test_plain_c_program_links_and_runs(tmp_path):
    """examples/score_topiq.c: a C99 program compiled against include/facet_engine.h and linked to the in-tree library. On a box
    without a GPU it must still start, parse a model with the host-only entry point and report the missing device cleanly."""
    import shutil
    import subprocess
    from standins import synthetic_onnx as S
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    exe = str(tmp_path / "score_topiq")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "score_topiq.c"), "-o", exe,
                    "-L" + os.path.join(root, "facet_amd"), "-lfacet_engine", "-Wl,-rpath," + os.path.join(root, "facet_amd")], check=True)
    model = tmp_path / "m.onnx"
    model.write_bytes(S.landmark_like(seed=1)[0])
    out = subprocess.run([exe, str(model)], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "facet_amd" in out and "73 nodes" in out and "input [1,3,192,192]" in out
    assert ("no engine context" in out) or ("fe_topiq_score -> " in out)      # CPU box / GPU box without weights


def test_every_exported_symbol_is_mapped_in_integration_md():
    """INTEGRATION.md §4 names, for each entry point of include/facet_engine.h, the reference call it stands behind."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "facet_engine.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    syms = sorted(set(re.findall(r"\b(fe_[a-z0-9_]+)\s*\(", header)))
    families = [p[:-1] for p in re.findall(r"`(fe_[a-z0-9_]+_\*)`", doc)]          # e.g. `fe_timer_*`
    missing = [s for s in syms if s not in doc and not any(s.startswith(f) for f in families)]
    assert len(syms) >= 50 and not missing, missing

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# no checkpoint files exist offline: the drop-in wrappers may fall back to the seeded synthetic checkpoints in tests only
# (without this opt-in they raise FileNotFoundError, see facet_amd/weights.py::checkpoint_or_synthetic)
os.environ.setdefault("FACET_AMD_SYNTHETIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One engine context on GPU 0 for the whole session. No skip: on the GPU box a missing
    libfacet_engine.so / device must fail loudly (there is no CPU fallback in the product path)."""
    from facet_amd import Engine
    eng = Engine(0, arena_bytes=6 << 30)
    yield eng
    eng.close()


def assert_int_boxes_match(got, want, tol=5e-3):
    """Face boxes after .astype(int) (what the reference stores, analyzers/face.py:113,145) must be IDENTICAL for every coordinate,
    the one exception being a coordinate whose oracle value lies within `tol` (5e-3 px) of an integer: GPU and CPU evaluate the fp32
    detector with different summation orders (measured worst case ~2e-3 px at 1024x1024), so a value like 162.9999 /
    163.0001 can truncate differently on the two sides - exact equality with a CPU fp32 detector is bounded by that, not by the
    engine. Coordinates themselves must agree within `tol`. Prints how many coordinates fell inside the band."""
    import numpy as np
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < tol, float(np.abs(got - want).max())
    in_band = np.abs(want - np.rint(want)) < tol
    same = got.astype(int) == want.astype(int)
    print(f"[face boxes] {want.size} coordinates: {int(in_band.sum())} within {tol} px of an integer, {int((~same).sum())} truncate differently, "
          f"max |gpu - cpu| = {np.abs(got - want).max():.2e} px")
    assert (same | in_band).all(), "a box coordinate outside the near-integer band truncates differently"

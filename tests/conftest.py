import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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


def assert_int_boxes_match(got, want, tol=2e-2):
    """Face boxes after .astype(int) (what the reference stores, analyzers/face.py:113,145) must be identical, except where a
    coordinate sits within `tol` of an integer: GPU and CPU evaluate the detector in fp32 with different summation orders, so a
    value like 162.9999 / 163.0001 can legitimately truncate differently. Coordinates themselves must agree within `tol`."""
    import numpy as np
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < tol, float(np.abs(got - want).max())
    near_edge = np.abs(want - np.rint(want)) < tol
    same = got.astype(int) == want.astype(int)
    assert (same | near_edge).all()
    assert same.mean() > 0.99

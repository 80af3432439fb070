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

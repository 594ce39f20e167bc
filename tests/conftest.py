import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "scalable-ccd_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/sccd_oracle.c) -- the checker, never the product."""
    import orc as _orc

    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def sccd():
    import sccd as _sccd

    return _sccd


@pytest.fixture(scope="session")
def ctx(sccd):
    """A GPU context; the HIP extension must be the thing that runs (no fallback)."""
    c = sccd.Context(0)
    yield c
    c.close()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def renderer():
    """One rt_handle on cuda:0 for the whole GPU session.  Fails (does not skip) when the HIP
    library or the device is missing: the GPU tests must exercise the native path."""
    from raytracer_amd import api
    r = api.Renderer(0)
    yield r
    r.close()

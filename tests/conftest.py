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
    # the tests name the candidate source they look at; the library's own per-scene choice has a test of its own
    r.set_option(api.OPT_ACCEL, api.ACCEL_SCAN)
    yield r
    r.close()


@pytest.fixture(autouse=True)
def _scan_unless_a_test_says_otherwise(monkeypatch):
    """Contexts the tests create themselves take the scan of every record as their candidate source (RT_ACCEL=0 is read
    by rt_create) unless a test sets the option: the tests name the source they look at.  The library's own per-scene
    choice (RT_ACCEL_AUTO) has a test of its own, which removes the variable."""
    monkeypatch.setenv("RT_ACCEL", "0")
    # the small frames of the tests go through two sub-pipelines on two streams, as the large ones of the product do
    # (left to the library that starts at 3 M pixels; its own choice has a test of its own)
    monkeypatch.setenv("RT_OVERLAP_MIN_PIXELS", "65536")

import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fdgs():
    """The product package (directory name starts with a digit, so it is imported by string)."""
    return importlib.import_module("4dgaussians_amd")


def have_reference():
    return os.path.isdir("/root/reference/scene")


def set_knob(name, value):
    """Set one of libfdgs's development knobs (fdgs_tuning_set; replaces the FDGS_* environment variables the library used to read per call).
    The autouse fixture below restores the load-time values after every test."""
    importlib.import_module("4dgaussians_amd._lib").tuning_set(name, int(value))


@pytest.fixture(autouse=True)
def _reset_knobs_after_test(request):
    # every test starts without a pair-count history: its first frame of an (image size, Gaussian count) takes the exact path, later frames of
    # the SAME test may take the capacity path -- and no test inherits a capacity learnt on another test's scene
    if request.node.get_closest_marker("gpu") is not None:
        importlib.import_module("4dgaussians_amd.rasterizer")._seen.clear()
    yield
    if request.node.get_closest_marker("gpu") is not None:
        L = importlib.import_module("4dgaussians_amd._lib")
        if L._lib is not None:
            L._lib.fdgs_tuning_reset()

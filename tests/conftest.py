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

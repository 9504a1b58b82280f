import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg():
    """The product package (directory name has a hyphen, so it is imported by name via importlib)."""
    return importlib.import_module("speech-backbones_amd")


@pytest.fixture(scope="session")
def sba():
    return pkg()


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))

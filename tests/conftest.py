import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "multimodal-av-model_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub: str = ""):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def avpkg():
    return pkg()

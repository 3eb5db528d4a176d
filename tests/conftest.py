import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG_NAME = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has hyphens, so import by string).  If the in-tree build is
    missing (a checkout without the git-ignored .so files) the test session builds it first -- the product
    itself never does that: load() fails loudly when libmlggd.so is absent."""
    mod = importlib.import_module(PKG_NAME)
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mod


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG_NAME + ".synth")


@pytest.fixture(scope="session")
def pyoracle():
    from oracle import pyoracle as po
    po.build()
    return po

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.import_package()


@pytest.fixture(scope="session")
def oracle(pkg):
    from oracle import loader
    return loader.load(pkg)


@pytest.fixture(scope="session")
def hip(pkg):
    """the product library; GPU tests fail loudly if it is missing"""
    try:
        # torch carries its own copy of the HIP runtime; it only finds the GPU when it initialises BEFORE the system
        # runtime liblvi_hip.so is linked against (measured on the GPU box), and some tests use torch.cuda later
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    return pkg.load_hip()

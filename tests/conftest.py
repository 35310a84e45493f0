import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_lib():
    """libfsgm_hip.so with a device present; GPU tests must never silently fall back."""
    from fsgm_amd import _lib
    lib = _lib.load()
    assert lib.fsgm_device_count() >= 1, "no HIP device visible: -m gpu tests need the GPU box"
    return lib

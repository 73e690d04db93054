import os
import sys

import pytest

# The library's default host policy is "auto" (bandwidth-bound hooks decline images in plain host memory: the CPU path is faster than two PCIe crossings).  The parity
# suites feed host arrays to EVERY hook on purpose, here and in the subprocesses they start (the reference's own test binary): they run under "always".
# tests/test_hal_dropin.py::test_default_host_policy checks the default in a process of its own.
os.environ.setdefault("MI355CV_HOST_POLICY", "always")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libocvref.so (the real reference built by oracle/ref/Makefile)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    return _orc


@pytest.fixture(scope="session")
def ref():
    import orc as _orc
    r = _orc.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libocvref.so not built (needs /root/reference: make -C oracle/ref)")
    return r

import os
import sys

import pytest

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

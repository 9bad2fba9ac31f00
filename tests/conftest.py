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
    from tests import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def hip():
    """The product library through its C ABI.  Fails loudly if it is not built."""
    import sipp_amd
    return sipp_amd.lib()

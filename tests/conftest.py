import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def ref():
    """The reference's own bindings compiled in place (oracle/_ref); skip when absent."""
    import oracle as o

    mod = o.load_ref()
    if mod is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return mod

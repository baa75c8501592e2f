import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import pgo_loader
    p = pgo_loader.load()
    p.build()
    return p


@pytest.fixture(scope="session")
def ds():
    import pgo_loader
    return pgo_loader.datasets()


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu(pkg):
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (no CPU fallback exists)")
    return pkg

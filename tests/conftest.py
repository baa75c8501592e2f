import os
import sys

import pytest

# Virtual ranks (tests/test_gpu_sharded.py) are streams of one device; the device-initiated exchange makes a launch of one rank wait
# for a kernel of another, which must not sit behind it in the same hardware queue.  ROCm maps streams onto GPU_MAX_HW_QUEUES (4 by
# default) queues per process: give every virtual rank's stream its own.  Read by the HIP runtime when it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

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


@pytest.fixture
def knobs(pkg):
    """knobs(name=value, ...): development / test knobs of the library (pgo_tuning_set; csrc/pgo_tuning.h) for the rest of the test;
    value None = the default.  Put back to what they were at teardown."""
    before = {}

    def set_(**kw):
        for k, v in kw.items():
            if k not in before:
                before[k] = pkg.tuning_get(k)
            pkg.tuning_set(k, v)
    yield set_
    for k, v in before.items():
        pkg.tuning_set(k, v)


@pytest.fixture(scope="session")
def gpu(pkg):
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (no CPU fallback exists)")
    return pkg

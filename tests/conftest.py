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
    """The CPU oracle (oracle/liboracle.so), built on demand.  Checker only."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def gpu_engine_cls():
    from crabml_amd.engine import Engine, device_available
    if not device_available():
        pytest.fail("gpu test selected but no HIP device is usable: the HIP path has no CPU fallback")
    return Engine

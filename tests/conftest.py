import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip():
    """The product's native backend; GPU tests fail loudly if it is missing or sees no device."""
    from adelie_amd import _abi

    b = _abi.hip_backend()
    assert b.fn("device_count")() > 0, "no HIP device visible"
    return b


import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from tests.helpers import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def product():
    """The HIP library through the C-ABI; GPU tests fail loudly if it is missing."""
    from ft_grandprix_amd import capi
    return capi.load()

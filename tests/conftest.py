import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the product library (hipcc cross-compiles without a GPU) and the oracle once per session."""
    from nimble_amd import _build
    from oracle import oracle
    _build.build()
    oracle.build()
    return True

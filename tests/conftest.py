import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The HIP library bound to cuda:0.  GPU tests fail loudly (no skip) when it is unavailable."""
    from metalign_amd._hip import Hip
    return Hip.get(0)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle

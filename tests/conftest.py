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


def apply_test_knobs():
    """MG_TEST_KNOBS="key=value,key=value" in a CHILD process's environment (a test that runs a script of its own: the rank scripts, this
    suite once more under another scanner) -> mg_debug_set.  The library itself reads no environment variable."""
    from metalign_amd import _hip
    for kv in filter(None, os.environ.get("MG_TEST_KNOBS", "").split(",")):
        k, v = kv.split("=")
        _hip.debug_set(k, int(v))


def pytest_sessionstart(session):
    if os.environ.get("MG_TEST_KNOBS"):
        apply_test_knobs()


@pytest.fixture()
def knobs():
    """set(key, value) -> mg_debug_set; every knob back to its default when the test ends."""
    from metalign_amd import _hip
    yield _hip.debug_set
    _hip.debug_set(None)

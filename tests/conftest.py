import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the single-shot entry point's on-disk program cache stays out of the tests (the test of the cache itself names a directory)
os.environ.setdefault("CWC_PROGRAM_CACHE", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import cwc_import
    p = cwc_import.load()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def oracle_c():
    from oracle import cbind
    cbind.lib()
    return cbind

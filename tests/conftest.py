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
def oracle():
    """The CPU restatement (oracle/liboracle.so) — the checker, never the product."""
    import oracle_bind
    return oracle_bind.load()


@pytest.fixture(scope="session")
def pgt():
    """The product package; building libpgtwin.so needs hipcc but no GPU."""
    import popgenomicstools_amd as pkg
    from popgenomicstools_amd import _lib
    _lib.load()
    return pkg


@pytest.fixture(scope="session")
def ctx(pgt):
    c = pgt.Context()
    yield c
    c.close()

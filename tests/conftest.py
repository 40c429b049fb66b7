import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure both native libraries exist (hipcc cross-compiles gfx950 without a GPU)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture()
def oracle_factory(built):
    from oracle_backend import OraclePhysxSystem

    return lambda tpl, n, cfg: OraclePhysxSystem(tpl, n, cfg)

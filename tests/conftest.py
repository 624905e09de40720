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
    from tests.oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree gfx950 extension; built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build_hip()
    import pomcpp_amd
    return pomcpp_amd.load_library()

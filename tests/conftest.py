import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fs():
    """The product package (ctypes host mirror over libfluidsim_hip.so)."""
    import __graft_entry__ as ge
    ge.build_product()      # the product build does not depend on the checker (orc fixture below)
    import gpu_fluid_simulation_amd as g
    return g


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as O
    O.build()
    return O

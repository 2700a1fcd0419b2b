import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Tests call through the C-ABI: make sure libsgcn.so and the oracle's C file are built
    (a no-op when the in-tree .so files travelled with the snapshot)."""
    import __graft_entry__ as g
    g.build(quiet=True)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")
    # Tests call through the C-ABI and some import the package at collection time: build
    # libsgcn.so and the oracle's C file BEFORE collection (a no-op when the in-tree .so files are
    # current, e.g. when they travelled with the gpurun snapshot).
    import __graft_entry__ as g
    g.build(quiet=True)

"""pytest configuration: markers, import path, and one-time builds of the oracle and the HIP library."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the CPU oracle (gcc) and the HIP library (hipcc cross-compiles without a GPU) once."""
    from oracle import oracle as _o
    _o.build()
    from feedback_gnn_amd import _lib
    _lib.build()  # make is incremental: a no-op when up to date, a rebuild after any edit of csrc/ (never a stale binary)
    yield

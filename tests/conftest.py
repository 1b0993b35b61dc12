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


# Order under `-x`: oracle-parity files first, harness files (bench contract, process launch, RCCL, backward) last, so that a harness
# failure can never again stop the run before the parity suite has been seen (round-4 GPUTEST: a bench-contract assertion sorted before
# every test_g*.py and `-x` skipped 294 parity tests).
_PARITY_FIRST = ("test_abi", "test_math", "test_gpu_math_bits", "test_golden_outputs", "test_gpu_parity", "test_gpu_pins",
                 "test_gpu_literal_forms", "test_gpu_gnn_order", "test_gpu_bp4_shared_lse", "test_gpu_big_codes", "test_gpu_api")
_HARNESS_LAST = ("test_gpu_backward", "test_gpu_dist", "test_gpu_rccl", "test_distributed_cpu", "test_launch", "test_bench_contract")


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _PARITY_FIRST:
        return _PARITY_FIRST.index(name)
    if name in _HARNESS_LAST:
        return 1000 + _HARNESS_LAST.index(name)
    return 500


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_file_rank)  # stable: the order inside a file is kept

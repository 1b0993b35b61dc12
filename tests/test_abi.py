"""The C-ABI library loads without a GPU and exports every symbol include/fgnn.h declares."""
import ctypes
import os
import re

import pytest

from feedback_gnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fgnn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_shim_binds():
    declared = _declared_functions()
    assert len(declared) >= 18
    assert set(declared) == set(_lib.ABI_SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libfgnn_hip.so must be built (conftest builds it)"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(L, name), name
    assert _lib.lib().fgnn_version() == 1


def test_argument_errors_cross_the_abi_as_codes():
    L = _lib.lib()
    out = ctypes.c_void_p()
    rc = L.fgnn_graph_create(0, 1, 1, 1, None, None, 1, None, None, 0, ctypes.byref(out))
    assert rc == -1 and b"positive" in L.fgnn_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "fgnn.h")).read()
    assert "torch" not in text.lower() and "at::" not in text and "#include <hip" not in text

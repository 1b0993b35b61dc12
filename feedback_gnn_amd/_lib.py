"""ctypes binding of libfgnn_hip.so (the C ABI declared in include/fgnn.h).

There is no CPU fallback: if the library is missing or a call fails, this module raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfgnn_hip.so")
CSRC = os.path.join(_HERE, "csrc")

CN_TYPES = {"boxplus": 0, "boxplus-phi": 1, "minsum": 2}
ROWS_X_LOGIT, ROWS_Z_LOGIT, ROWS_HX_PERP, ROWS_HZ_PERP, ROWS_LX, ROWS_LZ = 0, 1, 2, 3, 4, 5


class FgnnError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise FgnnError("building libfgnn_hip.so failed")
    return LIB_PATH


# Which sources make which kernel: profiles/traffic.json (rocprofv3 PMC counts taken offline) records the fingerprint of the
# sources its counts were measured on, and bench.py quotes a count only while the tree still matches it.
KERNEL_SOURCES = {
    "bp4": ("fgnn_bp4.hip", "fgnn_math.h", "fgnn_internal.h", "fgnn_rng.h", "Makefile"),
    "gnn": ("fgnn_gnn.hip", "fgnn_math.h", "fgnn_internal.h", "fgnn_pk.h", "Makefile"),
    "gnnbp4": ("fgnn_gnnbp4.hip", "fgnn_math.h", "fgnn_internal.h", "fgnn_pk.h", "Makefile"),
}


def source_fingerprint(kind):
    """sha256 over the csrc files (names and contents) that are compiled into the `kind` kernels ("bp4" / "gnn")."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kind]:
        h.update(name.encode() + b"\0")
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def library_sha256(path=None):
    import hashlib
    h = hashlib.sha256()
    with open(path or os.environ.get("FGNN_LIB_PATH", LIB_PATH), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


_SIGNATURES = {
    "fgnn_last_error": (C.c_char_p, []),
    "fgnn_version": (C.c_int, []),
    "fgnn_graph_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "fgnn_graph_destroy": (None, [C.c_void_p]),
    "fgnn_graph_set_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_graph_set_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "fgnn_graph_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "fgnn_graph_force_generic": (C.c_int, [C.c_void_p, C.c_int]),
    "fgnn_graph_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fgnn_graph_edges": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "fgnn_profile_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fgnn_bp4_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_int] + [C.c_void_p] * 10),
    "fgnn_bp4_decode_trace": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_int] + [C.c_void_p] * 10),
    "fgnn_weights_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "fgnn_weights_destroy": (None, [C.c_void_p]),
    "fgnn_feedback_gnn": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_pauli_noise": (C.c_int, [C.c_uint64, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_pauli_noise_xyz": (C.c_int, [C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_pauli_noise_wt": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_pauli_noise_dev": (C.c_int, [C.c_uint64, C.c_float, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_syndrome": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_flag_update": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_merge": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_residual": (C.c_int, [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4),
    "fgnn_count_flags": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_count_flags_batches": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_pack_decisions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_unpack_decisions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_sandwich_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "fgnn_sandwich_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fgnn_bp2_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "fgnn_bsc_noise": (C.c_int, [C.c_uint64, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_residual_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4),
    "fgnn_graph_set_basis": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fgnn_osd0": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "fgnn_compact": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgnn_gnnbp4_weights_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fgnn_gnnbp4_weights_destroy": (None, [C.c_void_p]),
    "fgnn_gnnbp4_weights_create_general": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "fgnn_gnnbp4_weights_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int]),
    "fgnn_gnnbp4_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "fgnn_gnnbp4_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
                           + [C.c_size_t, C.c_void_p]),
    "fgnn_weights_create_general": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fgnn_bp4_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
                          + [C.c_void_p] * 7),
    "fgnn_feedback_gnn_backward": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 9),
    "fgnn_feedback_gnn_backward_general": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3
                                           + [C.c_int, C.c_void_p]),
}

ABI_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """The loaded library (raises FgnnError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FgnnError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback for the decoder)")
        # PyTorch-ROCm must be loaded FIRST: the library then binds to the HIP runtime instance torch already
        # brought into the process (same soname), which is what makes torch's device pointers and streams
        # valid inside libfgnn_hip.so.  Loading the library before torch would start a second runtime.
        import torch  # noqa: F401
        override = os.environ.get("FGNN_LIB_PATH")  # A/B builds of the same ABI (tools/)
        L = C.CDLL(override or LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if override:  # an A/B build of an older tree may predate an entry point; the shipped library must export them all
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().fgnn_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise FgnnError(f"libfgnn_hip error {rc}: {msg}")

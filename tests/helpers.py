"""Shared helpers for the test-suite: code zoo, golden fixtures, oracle/GPU graph caches."""
import functools
import os

import numpy as np

from feedback_gnn_amd import codes_q as cq

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CODE_MAKERS = {
    "steane": lambda: cq.css_code(cq.hamming_code(3), cq.hamming_code(3), name="Steane_n7_k1_d3"),
    "rsurf3": lambda: cq.create_rotated_surface_codes(3),
    "rsurf5": lambda: cq.create_rotated_surface_codes(5),
    "surf3": lambda: cq.create_surface_codes(3),
    "toric4": lambda: cq.create_checkerboard_toric_codes(4),
    "gb48": lambda: cq.create_generalized_bicycle_codes(24, [0, 2, 8, 15], [0, 2, 12, 17], name="GB_n48_k6_d8"),
    "gb126": lambda: cq.create_generalized_bicycle_codes(63, [0, 1, 14, 16, 22], [0, 3, 13, 20, 42]),
    "gb254": lambda: cq.create_generalized_bicycle_codes(127, [0, 15, 20, 28, 66], [0, 58, 59, 100, 121]),
    "hp_c7": lambda: cq.hypergraph_product(cq.create_circulant_matrix(7, [0, 1, 3]), cq.create_circulant_matrix(7, [0, 1, 3])),
    "ibm72": lambda: cq.create_bivariate_QC_codes(6, 6, [3], [1, 2], [1, 2], [3]),
    "ghp882": lambda: cq.create_QC_GHP_codes(63, cq.create_cyclic_permuting_matrix(7, [27, 54, 0]), [0, 1, 6]),
    "ghp1270": lambda: cq.create_QC_GHP_codes(
        127, np.array([[0, -1, 51, 52, -1], [-1, 0, -1, 111, 20], [0, -1, 98, -1, 122], [0, 80, -1, 119, -1],
                       [-1, 0, 5, -1, 106]]), [0, 1, 7], name="GHP_n1270_k28"),
}



def _overcomplete(key):
    """The reference's over-complete GB check matrices (A-list data files, QLDPC.ipynb cell 5) from tests/golden/overcomplete.npz."""
    g = np.load(os.path.join(GOLDEN, "overcomplete.npz"))
    return cq.css_code(hx=unpack(g, key, "hx").astype(int), hz=unpack(g, key, "hz").astype(int), name=None, name_prefix="GB")


def _hp_big():
    """A hypergraph product too large for a CU's LDS: [[6480, 1296]] from a seeded (3,6)-like 36 x 72 matrix — 45 576 edges, so the BP4
    state of one codeword (E + 3n = 65 016 floats = 254 KB) takes the library's global-memory variant of the runtime-degree kernel."""
    rng = np.random.RandomState(7)
    m, n, dv = 36, 72, 3
    per = dv * n // m // dv
    h = np.zeros((m, n), dtype=int)
    for _ in range(dv):
        perm = rng.permutation(n)
        for r in range(m):
            h[r, perm[r * per:(r + 1) * per]] = 1
    return cq.hypergraph_product(h, h, "HP_n6480_lds_overflow")


# not one of the reference's constructions with a fixture under tests/golden (tests/test_codes.py walks CODE_MAKERS): its own table
EXTRA_CODE_MAKERS = {"hp_big": _hp_big}
CODE_MAKERS["gb46_oc"] = lambda: _overcomplete("gb46_oc")
CODE_MAKERS["gb48_oc"] = lambda: _overcomplete("gb48_oc")

WEIGHTS_882 = "feedback_GNN_n882_k24_wt_4_60_iter_64_16_mixed.npz"
WEIGHTS_1270 = "feedback_GNN_n1270_k28_wt_10_80_iter_64_16_mixed.npz"


@functools.lru_cache(maxsize=None)
def code(name):
    return (CODE_MAKERS.get(name) or EXTRA_CODE_MAKERS[name])()


@functools.lru_cache(maxsize=None)
def _oracle_graph(name, stage_one, forms):
    from oracle.oracle import OracleGraph
    return OracleGraph(code(name), stage_one=stage_one, forms=forms)


def oracle_library_forms(name, stage_one=True):
    """The checker of the default parity tests: the oracle's restatement of the operation sequence libfgnn_hip runs BY DEFAULT — what
    `gpu_graph(name)` computes out of the box.  Since round 6 that is the reference's formulas term by term (FGNN_OPT_GNN_FACTORED =
    FGNN_OPT_BP4_SHARED_LSE = 0: one Dense per edge, one log-sum-exp per edge), i.e. the same restatement as `oracle_literal_forms`;
    the two opt-in re-associations are `oracle_reassociated_forms`, and tests/test_gpu_bp4_shared_lse.py, test_gpu_gnn_order.py and
    test_gpu_literal_forms.py hold the kernels to the oracle with the options on and off.  One cached graph per (code, stage_one);
    tests that flip a form with its setters restore it (LIBRARY_FORMS below)."""
    return _oracle_graph(name, stage_one, "library-default")


def oracle_literal_forms(name, stage_one=True):
    """The oracle's restatement of the reference's formulas term by term: one log-sum-exp per edge (decoding_q.py:254-273), one Dense per
    edge (feedback_gnn.py:175-184, gnn.py:573-610)."""
    return _oracle_graph(name, stage_one, "literal")


def oracle_reassociated_forms(name, stage_one=True):
    """The oracle's restatement of the library's two opt-in re-associations (Dense layers factored, the log-sum-exp term shared per qubit
    side): the checker of a GPU graph with `set_gnn_factored(True)` and `set_bp4_shared_lse(True)`."""
    return _oracle_graph(name, stage_one, "reassociated")


# what a fresh TannerGraph / OracleGraph(forms="library-default") runs: tests that switch a form restore these values
LIBRARY_GNN_FACTORED = False
LIBRARY_BP4_SHARED_LSE = False


@functools.lru_cache(maxsize=None)
def gpu_graph(name, stage_one=True):
    from feedback_gnn_amd.graph import TannerGraph
    return TannerGraph(code(name), stage_one=stage_one)


def llr_const(p0):
    """log(3(1-p0)/p0) in float32 arithmetic, feedback_gnn.py:312."""
    p0 = np.float32(p0)
    return float(np.log(np.float32(3.0) * (np.float32(1.0) - p0) / p0, dtype=np.float32))


def golden_codes():
    out = dict(np.load(os.path.join(GOLDEN, "codes.npz")))
    out.update(np.load(os.path.join(GOLDEN, "overcomplete.npz")))
    return out


def unpack(g, key, attr):
    sh = g[f"{key}/{attr}_shape"]
    if sh[0] == 0:
        return np.zeros(sh, dtype=np.uint8)
    return np.unpackbits(g[f"{key}/{attr}"], axis=1)[:, :sh[1]]


def to_gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()

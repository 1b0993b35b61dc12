"""Codes whose per-codeword BP state exceeds a CU's LDS (E + 3n > about 40 000 floats): the library runs the same runtime-degree
kernel with that state in a global-memory workspace (bp4_kernel<..., GMEM>, fgnn_bp4.hip) instead of refusing the code.  Same float
operations in the same order, so everything is held to the oracle with exact equality like the LDS-resident kernels
(decoding_q.py:661-797 on a [[6480,1296]] hypergraph product with 45 576 edges; the reference handles any size — its tensors are in
device memory to begin with)."""
import numpy as np
import pytest
import torch

from helpers import LIBRARY_BP4_SHARED_LSE, WEIGHTS_882, code, gpu_graph, llr_const, oracle_library_forms, oracle_literal_forms, to_gpu

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def _inputs(B, p, first=3):
    og = oracle_library_forms("hp_big")
    ex, ez = og.pauli_noise(SEED, p, first, B)
    sx, sz = og.syndrome(ex, ez)
    return og, ex, ez, sx, sz


@pytest.mark.parametrize("cn_type,factor,per_qubit,B", [("boxplus-phi", 1.0, False, 5), ("boxplus-phi", 0.8, True, 5), ("minsum", 0.8, True, 5),
                                                         ("boxplus", 0.625, False, 5), ("boxplus-phi", 1.0, True, 261)])
def test_bp4_on_a_code_beyond_the_lds_budget(cn_type, factor, per_qubit, B):
    """B = 5 launches a thread per node (1 024 threads per codeword: the small-launch geometry), B = 261 the 256-thread geometry of
    chip-filling launches."""
    c = code("hp_big")
    assert c.N == 6480 and int(c.hx.sum() + c.hz.sum()) > 40896  # messages alone overflow 160 KB
    IT = 6
    og, ex, ez, sx, sz = _inputs(B, 0.03)
    gg = gpu_graph("hp_big")
    kw = dict(llr_const=llr_const(0.05))
    gkw = dict(kw)
    if per_qubit:
        llr = np.random.RandomState(1).uniform(1.0, 5.0, size=(B, 3, c.N)).astype(np.float32)
        kw, gkw = dict(llr_ch=llr), dict(llr_ch=to_gpu(llr))
    o = og.bp4_decode(sx, sz, IT, cn_type, factor, return_msgs=True, **kw)
    g = gg.bp4_decode(to_gpu(sx), to_gpu(sz), IT, cn_type, factor, return_msgs=True, **gkw)
    for k in ("llr", "x_hat", "z_hat", "x_logit", "z_logit", "msg_x", "msg_z"):
        assert np.array_equal(o[k], g[k].cpu().numpy()), k
    assert (o["x_hat"] ^ ex).any() or (o["x_hat"] == ex).all()  # (decisions are real outputs, not zeros)
    # a restart from the returned messages continues the same decode: 6 + 4 iterations = 10 iterations
    o10 = og.bp4_decode(sx, sz, IT + 4, cn_type, factor, **kw)
    g4 = gg.bp4_decode(to_gpu(sx), to_gpu(sz), 4, cn_type, factor, msg_init=(g["msg_x"], g["msg_z"]), **gkw)
    for k in ("llr", "x_hat", "z_hat"):
        assert np.array_equal(o10[k], g4[k].cpu().numpy()), k


def test_literal_form_and_sandwich_on_a_code_beyond_the_lds_budget():
    """The literal log-sum-exp form, and the whole sandwich (the decoders' fused flag test keeps its decision bytes in the workspace
    row; the feedback GNN runs its runtime-degree kernel) against the oracle."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    B = 4
    og, ex, ez, sx, sz = _inputs(B, 0.04, first=40)
    gg = gpu_graph("hp_big")
    ol = oracle_literal_forms("hp_big")
    gg.set_bp4_shared_lse(False)
    try:
        o = ol.bp4_decode(sx, sz, 5, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
        g = gg.bp4_decode(to_gpu(sx), to_gpu(sz), 5, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    finally:
        gg.set_bp4_shared_lse(LIBRARY_BP4_SHARED_LSE)
    for k in ("llr", "x_hat", "z_hat", "x_logit", "z_logit"):
        assert np.array_equal(o[k], g[k].cpu().numpy()), k
    w = read_weight_list(WEIGHTS_882)
    so = og.sandwich_decode(sx, sz, [5, 3, 2], [w, w], llr_const(0.05), return_llr=True)
    sg = gg.sandwich_decode(to_gpu(sx), to_gpu(sz), [5, 3, 2], [GnnWeights(w, gg.device)] * 2, llr_const(0.05), return_llr=True,
                            return_rounds=True)
    for k in ("x_hat", "z_hat", "llr", "rounds"):
        assert np.array_equal(so[k], sg[k].cpu().numpy()), k
    s0, l0, f0 = og.residual(ex, ez, so["x_hat"], so["z_hat"])
    s1, l1, f1 = gg.residual(to_gpu(ex), to_gpu(ez), sg["x_hat"], sg["z_hat"], want_arrays=True)
    assert np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(l0, l1.cpu().numpy()) and np.array_equal(f0, f1.cpu().numpy())

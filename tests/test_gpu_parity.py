"""GPU parity tests: every HIP entry point of libfgnn_hip.so against the CPU oracle.

The kernels and the oracle share one float32 operation sequence (fgnn_math.h, canonical summation
order), so the bar here is EXACT equality of every float and every decision — for converged and
non-converged samples alike — not a tolerance.  (The oracle itself is pinned to the reference by
tests/test_oracle_kat.py within the north-star tolerance of 1e-4.)
"""
import os

import numpy as np
import pytest
import torch

from helpers import (LIBRARY_BP4_SHARED_LSE, LIBRARY_GNN_FACTORED, WEIGHTS_882, WEIGHTS_1270, code, gpu_graph, llr_const,
                     oracle_library_forms, to_gpu)

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def _noise_and_syndromes(name, p, B, first=0):
    og, gg = oracle_library_forms(name), gpu_graph(name)
    ex, ez = og.pauli_noise(SEED, p, first, B)
    gx, gz = gg.pauli_noise(SEED, p, first, B)
    assert np.array_equal(ex, gx.cpu().numpy()) and np.array_equal(ez, gz.cpu().numpy())
    sx, sz = og.syndrome(ex, ez)
    tx, tz = gg.syndrome(gx, gz)
    assert np.array_equal(sx, tx.cpu().numpy()) and np.array_equal(sz, tz.cpu().numpy())
    return (ex, ez, sx, sz), (gx, gz, tx, tz)


def _assert_bp_equal(o, g, what=""):
    for k in ("llr", "x_logit", "z_logit", "msg_x", "msg_z"):
        if k in o and o[k] is not None and g.get(k) is not None:
            a, b = o[k], g[k].cpu().numpy()
            assert np.array_equal(a, b), f"{what} {k}: max|d|={np.abs(a - b).max()} at {np.argwhere(a != b)[:3]}"
    for k in ("x_hat", "z_hat"):
        assert np.array_equal(o[k], g[k].cpu().numpy()), f"{what} {k}"


@pytest.mark.parametrize("name,p", [("ghp882", 0.05), ("ghp882", 0.10), ("ghp1270", 0.08)])
@pytest.mark.parametrize("iters", [1, 2, 16, 64])
def test_bp4_phi_bit_exact(name, p, iters):
    B = 48
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B)
    L0 = llr_const(0.05)
    o = oracle_library_forms(name).bp4_decode(sx, sz, iters, "boxplus-phi", 1.0, llr_const=L0, return_msgs=True)
    g = gpu_graph(name).bp4_decode(tx, tz, iters, "boxplus-phi", 1.0, llr_const=L0, return_msgs=True)
    _assert_bp_equal(o, g, f"{name} p={p} it={iters}")


@pytest.mark.parametrize("cn_type,factor", [("minsum", 0.625), ("boxplus", 0.625), ("boxplus-phi", 0.8)])
def test_bp4_cn_variants_bit_exact(cn_type, factor):
    B = 40
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes("ghp882", 0.07, B, first=1000)
    L0 = llr_const(0.3)
    o = oracle_library_forms("ghp882").bp4_decode(sx, sz, 24, cn_type, factor, llr_const=L0, return_msgs=True)
    g = gpu_graph("ghp882").bp4_decode(tx, tz, 24, cn_type, factor, llr_const=L0, return_msgs=True)
    _assert_bp_equal(o, g, cn_type)


@pytest.mark.parametrize("name", ["steane", "rsurf3", "rsurf5", "surf3", "toric4", "gb48", "gb126", "hp_c7", "ibm72"])
@pytest.mark.parametrize("cn_type", ["boxplus-phi", "minsum", "boxplus"])
def test_bp4_small_and_irregular_codes(name, cn_type):
    B = 77  # not a multiple of codewords-per-block: exercises the padded last workgroup
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.08, B)
    L0 = llr_const(0.05)
    o = oracle_library_forms(name).bp4_decode(sx, sz, 12, cn_type, 0.8, llr_const=L0, return_msgs=True)
    g = gpu_graph(name).bp4_decode(tx, tz, 12, cn_type, 0.8, llr_const=L0, return_msgs=True)
    _assert_bp_equal(o, g, f"{name} {cn_type}")


def test_bp4_message_step_with_edge_cases():
    """One iteration from crafted c->v messages: exact zeros, +-20, duplicates, tiny values, huge values."""
    name, B = "ghp882", 16
    og, gg = oracle_library_forms(name), gpu_graph(name)
    rng = np.random.RandomState(1)
    mx = rng.uniform(-10, 10, size=(B, og.E_x)).astype(np.float32)
    mz = rng.uniform(-10, 10, size=(B, og.E_z)).astype(np.float32)
    specials = np.array([0.0, -0.0, 20.0, -20.0, 8.5e-8, -8.5e-8, 1e-9, 16.635532, -16.635532, 37.5, -60.0, 1e-3],
                        dtype=np.float32)
    mx[1::2, ::7] = specials[rng.randint(0, len(specials), size=mx[1::2, ::7].shape)]
    mz[1::2, ::5] = specials[rng.randint(0, len(specials), size=mz[1::2, ::5].shape)]
    mx[2] = 0.0
    mz[2] = 0.0
    mx[3] = 3.25  # all equal -> duplicate minima everywhere
    mz[3] = -3.25
    sx = rng.randint(0, 2, size=(B, og.m_x)).astype(np.uint8)
    sz = rng.randint(0, 2, size=(B, og.m_z)).astype(np.uint8)
    llr = rng.uniform(0.2, 6.0, size=(B, 3, og.n)).astype(np.float32)
    for cn_type in ("boxplus-phi", "minsum", "boxplus"):
        for iters in (1, 3):
            o = og.bp4_decode(sx, sz, iters, cn_type, 0.9, llr_ch=llr, msg_init=(mx, mz), return_msgs=True)
            g = gg.bp4_decode(to_gpu(sx), to_gpu(sz), iters, cn_type, 0.9, llr_ch=to_gpu(llr),
                              msg_init=(to_gpu(mx), to_gpu(mz)), return_msgs=True)
            _assert_bp_equal(o, g, f"step {cn_type} {iters}")


def test_bp4_non_stage_one_logits():
    """Soft syndrome over the dense hx_perp/hz_perp rows (decoding_q.py:33-34 when not stage_one)."""
    name, B = "gb48", 33
    og, gg = oracle_library_forms(name, False), gpu_graph(name, False)
    ex, ez = og.pauli_noise(SEED, 0.06, 0, B)
    sx, sz = og.syndrome(ex, ez)
    o = og.bp4_decode(sx, sz, 10, "boxplus-phi", 0.625, llr_const=llr_const(0.1))
    g = gg.bp4_decode(to_gpu(sx), to_gpu(sz), 10, "boxplus-phi", 0.625, llr_const=llr_const(0.1))
    assert o["x_logit"].shape[1] == code(name).hx_perp.shape[0]
    _assert_bp_equal(o, g, "non-stage-one")


@pytest.mark.parametrize("name,wfile,p", [("ghp882", WEIGHTS_882, 0.10), ("ghp1270", WEIGHTS_1270, 0.09)])
def test_feedback_gnn_bit_exact(name, wfile, p):
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    B = 24
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B)
    w = read_weight_list(wfile)
    o = og.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    on = og.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    gw = GnnWeights(w, gg.device)
    gn = gg.feedback_gnn(gw, to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), tx, tz).cpu().numpy()
    assert np.array_equal(on, gn), f"max|d|={np.abs(on - gn).max()}"
    assert np.isfinite(gn).all()


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name,wfile,iters,p", [("ghp882", WEIGHTS_882, [64, 16], 0.10),
                                                ("ghp882", WEIGHTS_882, [64, 16, 16, 16], 0.12),
                                                ("ghp1270", WEIGHTS_1270, [64, 64], 0.10)])
def test_sandwich_and_residual_bit_exact(name, wfile, iters, p, compact):
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    B = 96
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B, first=12345)
    w = read_weight_list(wfile)
    gw = GnnWeights(w, gg.device)
    nl = len(iters)
    o = og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), llr_const(0.05), return_llr=True)
    g = gg.sandwich_decode(tx, tz, iters, [gw] * (nl - 1), llr_const(0.05), compact=compact, return_llr=True,
                           return_rounds=True)
    assert np.array_equal(o["x_hat"], g["x_hat"].cpu().numpy())
    assert np.array_equal(o["z_hat"], g["z_hat"].cpu().numpy())
    assert np.array_equal(o["rounds"], g["rounds"].cpu().numpy())
    assert o["rounds"].sum() > 0, "test must exercise the feedback rounds"
    if not compact:
        assert np.array_equal(o["llr"], g["llr"].cpu().numpy())
    s0, l0, f0 = og.residual(ex, ez, o["x_hat"], o["z_hat"])
    s1, l1, f1 = gg.residual(gx, gz, g["x_hat"], g["z_hat"])
    assert np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(l0, l1.cpu().numpy())
    assert np.array_equal(f0, f1.cpu().numpy())


@pytest.mark.parametrize("cn_type", ["boxplus-phi", "minsum", "boxplus"])
def test_regular_and_runtime_degree_kernels_agree(cn_type):
    """[[882,24]] is (3,3,6)-regular and takes the specialised kernel; forcing the CSR kernel must give the same bits."""
    B = 40
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes("ghp882", 0.09, B, first=31)
    gg = gpu_graph("ghp882")
    assert gg.info()["regular"] == 1
    o = oracle_library_forms("ghp882").bp4_decode(sx, sz, 20, cn_type, 0.8, llr_const=llr_const(0.05), return_msgs=True)
    a = gg.bp4_decode(tx, tz, 20, cn_type, 0.8, llr_const=llr_const(0.05), return_msgs=True)
    gg.force_generic(True)
    try:
        b = gg.bp4_decode(tx, tz, 20, cn_type, 0.8, llr_const=llr_const(0.05), return_msgs=True)
    finally:
        gg.force_generic(False)
    _assert_bp_equal(o, a, "regular")
    _assert_bp_equal(o, b, "generic")


@pytest.mark.parametrize("cn_type,factor", [("minsum", 0.8), ("boxplus", 0.625), ("boxplus-phi", 0.875)])
def test_regular_cn_rules_equal_runtime_degree_kernel_at_scale(cn_type, factor):
    """The register-resident check updates of the (3,3,6)-regular kernel (cn_phi_regular, cn_minsum_regular, cn_tanh_regular) against
    the runtime-degree kernel — itself pinned to the oracle above — on 8192 codewords x 64 iterations either side of the waterfall:
    same marginals, decisions and soft syndromes, bit for bit."""
    gg = gpu_graph("ghp882")
    B = 8192
    gg.set_saturation_shortcut(False)
    try:
        for p in (0.04, 0.11):
            ex, ez = gg.pauli_noise(SEED + 11, p, 0, B)
            sx, sz = gg.syndrome(ex, ez)
            a = gg.bp4_decode(sx, sz, 64, cn_type, factor, llr_const=llr_const(0.05))
            gg.force_generic(True)
            try:
                b = gg.bp4_decode(sx, sz, 64, cn_type, factor, llr_const=llr_const(0.05))
            finally:
                gg.force_generic(False)
            for k in ("llr", "x_logit", "z_logit"):
                assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (cn_type, p, k)
            assert torch.equal(a["x_hat"], b["x_hat"]) and torch.equal(a["z_hat"], b["z_hat"])
    finally:
        gg.set_saturation_shortcut(True)


def test_launch_geometry_does_not_change_results():
    B = 24
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes("ghp882", 0.09, B, first=99)
    gg = gpu_graph("ghp882")
    o = oracle_library_forms("ghp882").bp4_decode(sx, sz, 16, "boxplus-phi", 1.0, llr_const=llr_const(0.05), return_msgs=True)
    try:
        for tpc, cpb in ((256, 1), (128, 2), (512, 1), (1024, 1), (64, 4)):
            gg.set_launch(tpc, cpb)
            _assert_bp_equal(o, gg.bp4_decode(tx, tz, 16, "boxplus-phi", 1.0, llr_const=llr_const(0.05), return_msgs=True),
                             f"tpc={tpc} cpb={cpb}")
    finally:
        gg.set_launch(0, 0)


def test_gnn_mfma_and_valu_kernels_agree():
    """The MFMA kernel (regular graphs) and the scalar-weight VALU kernel (any graph) both equal the oracle."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    name, B = "ghp882", 19
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.11, B, first=4242)
    w = read_weight_list(WEIGHTS_882)
    # random weights too: the trained ones could hide a permutation error in a near-zero column
    rng = np.random.RandomState(5)
    wr = [rng.uniform(-0.7, 0.7, size=a.shape).astype(np.float32) for a in w]
    o = og.bp4_decode(sx, sz, 32, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    for ww in (w, wr):
        ref = og.feedback_gnn(ww, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
        gw = GnnWeights(ww, gg.device)
        args = (gw, to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), tx, tz)
        a = gg.feedback_gnn(*args).cpu().numpy()
        gg.force_generic(True)
        try:
            b = gg.feedback_gnn(*args).cpu().numpy()
        finally:
            gg.force_generic(False)
        assert np.array_equal(ref, b), f"VALU kernel: max|d|={np.abs(ref - b).max()}"
        assert np.array_equal(ref, a), f"MFMA kernel: max|d|={np.abs(ref - a).max()} first {np.argwhere(ref != a)[:4]}"


@pytest.mark.parametrize("name,wfile", [("ghp882", WEIGHTS_882), ("ghp1270", WEIGHTS_1270)])
def test_small_launch_geometry_is_exact(name, wfile):
    """Launches of at most 256 codewords get a thread per node (fgnn_geom) and, below 1 024 codewords, the feedback GNN deals a
    codeword's tiles to several wave-quads (nsplit): a compacted feedback round on a handful of samples.  Every kernel on the path
    must give the same bits for the first samples of a small launch and of a large one (and both equal the oracle)."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    gg, og = gpu_graph(name), oracle_library_forms(name)
    L0 = llr_const(0.05)
    w = read_weight_list(wfile)
    gw = GnnWeights(w, gg.device)
    Bbig = 1500
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.10, Bbig, first=31337)
    big = gg.bp4_decode(tx, tz, 20, "boxplus-phi", 1.0, llr_const=L0, return_msgs=True)
    big_gnn = gg.feedback_gnn(gw, big["llr"], big["z_logit"], big["x_logit"], tx, tz)
    big2 = gg.bp4_decode(tx, tz, 7, "boxplus-phi", 0.9, llr_ch=big_gnn)
    for B in (1, 3, 64, 200, 256, 257):
        small = gg.bp4_decode(tx[:B].contiguous(), tz[:B].contiguous(), 20, "boxplus-phi", 1.0, llr_const=L0, return_msgs=True)
        for k in ("llr", "x_hat", "z_hat", "x_logit", "z_logit", "msg_x", "msg_z"):
            assert torch.equal(small[k], big[k][:B]), (B, k)
        sg = gg.feedback_gnn(gw, small["llr"], small["z_logit"], small["x_logit"], tx[:B].contiguous(), tz[:B].contiguous())
        assert torch.equal(sg, big_gnn[:B]), (B, "gnn")
        s2 = gg.bp4_decode(tx[:B].contiguous(), tz[:B].contiguous(), 7, "boxplus-phi", 0.9, llr_ch=sg)
        for k in ("llr", "x_hat", "z_hat"):
            assert torch.equal(s2[k], big2[k][:B]), (B, k, "stage 2")
        sx_s, sz_s = gg.syndrome(gx[:B].contiguous(), gz[:B].contiguous())
        assert torch.equal(sx_s, tx[:B]) and torch.equal(sz_s, tz[:B])
        r_s = gg.residual(gx[:B].contiguous(), gz[:B].contiguous(), small["x_hat"], small["z_hat"])
        r_b = gg.residual(gx, gz, big["x_hat"], big["z_hat"])
        for a_, b_ in zip(r_s, r_b):
            assert torch.equal(a_, b_[:B]), (B, "residual")
    o = og.bp4_decode(sx[:64], sz[:64], 20, "boxplus-phi", 1.0, llr_const=L0, return_msgs=True)
    _assert_bp_equal(o, {k: v[:64] for k, v in big.items()}, "large launch vs oracle")
    ref = og.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx[:64], sz[:64])
    assert np.array_equal(ref, big_gnn[:64].cpu().numpy())


@pytest.mark.parametrize("name,p", [("ghp882", 0.01), ("ghp882", 0.06), ("ghp1270", 0.02)])
def test_saturation_shortcut_is_exact(name, p):
    """Low p: codewords converge early and the wave-uniform shortcut fires for most of the 64 iterations.
    With the option on and off the kernel must reproduce the oracle bit for bit."""
    B = 64
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B, first=777)
    gg = gpu_graph(name)
    o = oracle_library_forms(name).bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05), return_msgs=True)
    assert np.abs(o["msg_x"]).max() == np.float32(16.635532)  # saturated state reached
    try:
        for on in (True, False):
            gg.set_saturation_shortcut(on)
            g = gg.bp4_decode(tx, tz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05), return_msgs=True)
            _assert_bp_equal(o, g, f"shortcut={on}")
        # per-qubit channel LLRs (second-stage style input) and a normalisation factor != 1
        llr = np.random.RandomState(3).uniform(0.3, 3.0, size=(B, 3, gg.n)).astype(np.float32)
        o2 = oracle_library_forms(name).bp4_decode(sx, sz, 40, "boxplus-phi", 0.8, llr_ch=llr, return_msgs=True)
        for on in (True, False):
            gg.set_saturation_shortcut(on)
            _assert_bp_equal(o2, gg.bp4_decode(tx, tz, 40, "boxplus-phi", 0.8, llr_ch=to_gpu(llr), return_msgs=True),
                             f"llr_ch shortcut={on}")
    finally:
        gg.set_saturation_shortcut(True)


@pytest.mark.parametrize("name,p,iters,factor", [("ghp882", 0.01, 64, 1.0), ("ghp882", 0.05, 64, 1.0), ("ghp882", 0.09, 64, 1.0),
                                                  ("ghp882", 0.03, 200, 1.0), ("ghp882", 0.03, 3, 1.0), ("ghp882", 0.04, 7, 0.8),
                                                  ("ghp882", 0.05, 1, 1.0), ("ghp882", 0.05, 2, 0.8), ("ghp1270", 0.05, 1, 0.625),
                                                  ("ghp1270", 0.02, 64, 1.0), ("ghp1270", 0.09, 48, 0.9)])
def test_fixed_point_exit_is_exact(name, p, iters, factor):
    """FGNN_OPT_FIXED_POINT_EXIT: a workgroup leaves the iteration loop once its messages are provably at a bit-exact fixed
    point.  With the exit on, off, and with the shortcut off altogether the kernel must reproduce the oracle (which runs
    every iteration) bit for bit — messages, marginals, decisions, soft syndromes — for converged, oscillating and
    non-converged samples alike, also when launched with few or many iterations."""
    B = 96
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B, first=4242)
    gg = gpu_graph(name)
    L0 = llr_const(0.05)
    o = oracle_library_forms(name).bp4_decode(sx, sz, iters, "boxplus-phi", factor, llr_const=L0, return_msgs=True)
    try:
        for shortcut, fpe in ((True, True), (True, False), (False, True)):
            gg.set_saturation_shortcut(shortcut)
            gg.set_fixed_point_exit(fpe)
            g = gg.bp4_decode(tx, tz, iters, "boxplus-phi", factor, llr_const=L0, return_msgs=True)
            _assert_bp_equal(o, g, f"shortcut={shortcut} exit={fpe}")
        gg.set_saturation_shortcut(True)
        gg.set_fixed_point_exit(True)
        # chained launches (stage-two style): restart from the returned messages, with per-qubit channel LLRs
        llr = np.random.RandomState(5).uniform(0.5, 3.0, size=(B, 3, gg.n)).astype(np.float32)
        o1 = oracle_library_forms(name).bp4_decode(sx, sz, iters, "boxplus-phi", factor, llr_ch=llr, return_msgs=True)
        o2 = oracle_library_forms(name).bp4_decode(sx, sz, 5, "boxplus-phi", factor, llr_ch=llr, msg_init=(o1["msg_x"], o1["msg_z"]),
                                           return_msgs=True)
        g1 = gg.bp4_decode(tx, tz, iters, "boxplus-phi", factor, llr_ch=to_gpu(llr), return_msgs=True)
        g2 = gg.bp4_decode(tx, tz, 5, "boxplus-phi", factor, llr_ch=to_gpu(llr), msg_init=(g1["msg_x"], g1["msg_z"]),
                           return_msgs=True)
        _assert_bp_equal(o1, g1, "llr_ch")
        _assert_bp_equal(o2, g2, "restart")
    finally:
        gg.set_saturation_shortcut(True)
        gg.set_fixed_point_exit(True)


@pytest.mark.parametrize("name,p,iters,factor,launch", [("ghp882", 0.02, 64, 1.0, "generic"), ("ghp882", 0.09, 40, 0.9, "generic"),
                                                         ("gb254", 0.02, 64, 1.0, (256, 1)), ("gb254", 0.02, 64, 0.625, None),
                                                         ("gb126", 0.02, 50, 0.8, (128, 1)), ("hp_c7", 0.02, 30, 1.0, (64, 1)),
                                                         ("hp_c7", 0.02, 30, 1.0, None), ("gb48_oc", 0.02, 12, 1.0, (256, 1)),
                                                         ("rsurf5", 0.03, 30, 1.0, None)])
def test_exact_shortcuts_on_runtime_degree_graphs(name, p, iters, factor, launch):
    """The saturation shortcut and the fixed-point exit in the CSR kernel (any degrees; the detector needs <= 32 edges per
    qubit and one codeword per workgroup, otherwise only the shortcut acts): on, off, and against the oracle, bit for bit."""
    B = 80
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B, first=99)
    gg = gpu_graph(name)
    L0 = llr_const(0.1)
    o = oracle_library_forms(name).bp4_decode(sx, sz, iters, "boxplus-phi", factor, llr_const=L0, return_msgs=True)
    try:
        if launch == "generic":
            gg.force_generic(True)
        elif launch is not None:
            gg.set_launch(*launch)
        for shortcut, fpe in ((True, True), (True, False), (False, False)):
            gg.set_saturation_shortcut(shortcut)
            gg.set_fixed_point_exit(fpe)
            g = gg.bp4_decode(tx, tz, iters, "boxplus-phi", factor, llr_const=L0, return_msgs=True)
            _assert_bp_equal(o, g, f"{name} shortcut={shortcut} exit={fpe}")
        gg.set_saturation_shortcut(True)
        gg.set_fixed_point_exit(True)
        for cn in ("minsum", "boxplus"):  # the qubit-side shortcut also runs under the other rules
            o2 = oracle_library_forms(name).bp4_decode(sx, sz, 12, cn, 0.8, llr_const=L0, return_msgs=True)
            _assert_bp_equal(o2, gg.bp4_decode(tx, tz, 12, cn, 0.8, llr_const=L0, return_msgs=True), f"{name} {cn}")
    finally:
        gg.force_generic(False)
        gg.set_launch(0, 0)
        gg.set_saturation_shortcut(True)
        gg.set_fixed_point_exit(True)


def _gnnbp4_weights(seed=11):
    from feedback_gnn_amd.graph import GNNBP4_SHAPES
    rng = np.random.RandomState(seed)
    w = []
    for shp in GNNBP4_SHAPES:
        lim = 0.6 if len(shp) == 1 else np.sqrt(6.0 / (shp[0] + shp[1]))
        w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    return w


@pytest.mark.parametrize("name,B,iters", [("gb48", 21, 5), ("rsurf5", 9, 3), ("ghp882", 6, 4), ("ghp1270", 5, 10)])
def test_gnn_bp4_bit_exact(name, B, iters):
    """GNN_BP4 (BASELINE configs[4]) kernel vs the oracle: embeddings-derived LLRs, soft syndromes of every iteration
    and hard decisions, exactly."""
    from feedback_gnn_amd.graph import GnnBp4Weights
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.05, B)
    w = _gnnbp4_weights()
    o = og.gnn_bp4(w, sx, sz, iters)
    g = gg.gnn_bp4_decode(GnnBp4Weights(w, gg.device), tx, tz, iters)
    for k in ("llr", "x_logit_all", "z_logit_all", "x_hat", "z_hat"):
        a, b = o[k], g[k].cpu().numpy()
        assert np.array_equal(a, b), f"{name} {k}: max|d|={np.abs(a.astype(np.float64) - b).max()}"


def test_gnn_bp4_class_contract():
    import feedback_gnn_amd as F
    c = code("gb48")
    dec = F.GNN_BP4(c, num_embed_dims=20, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, num_iter=3, reduce_op="mean",
                    activation="tanh", use_bias=True)
    w = _gnnbp4_weights(3)
    dec.set_weights(w)
    og = oracle_library_forms("gb48")
    ex, ez = og.pauli_noise(SEED, 0.05, 0, 7)
    sx, sz = og.syndrome(ex, ez)
    llr_hat, x_hat, z_hat = dec((to_gpu(sx.astype(np.int64)), to_gpu(sz.astype(np.int64))))
    o = og.gnn_bp4(w, sx, sz, 3)
    assert len(llr_hat) == 3 and llr_hat[0][0].shape == (og.m_z + og.rows_lz, 7) and x_hat.shape == (c.N, 7)
    assert np.array_equal(o["x_logit_all"][2].T, llr_hat[2][0].cpu().numpy())
    assert np.array_equal(o["x_hat"].T, x_hat.cpu().numpy()) and x_hat.dtype == torch.int64
    with pytest.raises(ValueError, match="unknown reduce operation"):
        F.GNN_BP4(c, 16, 20, 40, 2, 3, reduce_op="median")
    with pytest.raises(NotImplementedError):
        F.GNN_BP4(c, 16, 20, 40, 2, 3, activation="gelu")
    with pytest.raises(NotImplementedError):
        F.GNN_BP4(c, 16, 20, 40, 2, 3, loss_type="sine")


GNNBP4_GEN_CONFIGS = [(20, 40, 2, "mean", "tanh", True, False, 0, 0), (8, 16, 1, "max", "relu", False, False, 0, 0),
                      (12, 24, 3, "sum", "sigmoid", True, True, 3, 2), (5, 7, 2, "min", "linear", True, True, 2, 0),
                      (32, 96, 4, "mean", "tanh", False, True, 0, 4)]


def _gnnbp4_gen_weights(graph, cfg, seed=3):
    from feedback_gnn_amd.graph import gnnbp4_weight_shapes
    rng = np.random.RandomState(seed)
    out = []
    for shp in gnnbp4_weight_shapes(graph, cfg):
        lim = 0.6 if len(shp) == 1 else np.sqrt(6.0 / (shp[0] + shp[1]))
        out.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    return out


def _gnnbp4_cfg_codes(cfg):
    from feedback_gnn_amd.graph import ACTIVATIONS, REDUCE_OPS
    return (cfg[0], cfg[1], cfg[2], REDUCE_OPS[cfg[3]], ACTIVATIONS[cfg[4]], int(cfg[5]), int(cfg[6]), cfg[7], cfg[8])


@pytest.mark.parametrize("cfg", GNNBP4_GEN_CONFIGS)
@pytest.mark.parametrize("name", ["rsurf5", "gb48", "ghp882"])
def test_general_gnn_bp4_bit_exact(name, cfg):
    """GNN_BP4 with any constructor setting (fgnn_gnnbp4_weights_create_general, runtime-shaped kernel: widths, depth, all four
    reduce ops, all four activations, bias on / off, trainable node and edge attributes) against the oracle's og_gnn_bp4_general,
    exactly; on the benchmark setting it also equals the specialised kernels in the literal association."""
    from feedback_gnn_amd.graph import GnnBp4Weights
    og, gg = oracle_library_forms(name), gpu_graph(name)
    B, iters = (4, 3) if name == "ghp882" else (11, 4)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.06, B, first=17)
    w = _gnnbp4_gen_weights(gg, cfg)
    o = og.gnn_bp4_general(_gnnbp4_cfg_codes(cfg), w, sx, sz, iters)
    g = gg.gnn_bp4_decode(GnnBp4Weights(w, gg.device, config=cfg, graph=gg, force_general=True), tx, tz, iters)
    for k in ("llr", "x_logit_all", "z_logit_all", "x_hat", "z_hat"):
        a, b = o[k], g[k].cpu().numpy()
        assert np.array_equal(a, b), f"{name} {cfg} {k}: max|d|={np.abs(a.astype(np.float64) - b).max()}"
    assert np.isfinite(o["llr"]).all()
    if cfg == GNNBP4_GEN_CONFIGS[0]:
        gg.set_gnn_factored(False)
        try:
            sp = gg.gnn_bp4_decode(GnnBp4Weights(w, gg.device), tx, tz, iters)
        finally:
            gg.set_gnn_factored(LIBRARY_GNN_FACTORED)
        assert torch.equal(sp["llr"], g["llr"]) and torch.equal(sp["x_logit_all"], g["x_logit_all"])


def test_general_gnn_bp4_class_surface():
    """The class with a non-benchmark setting: Keras-style initial weights (zero `_llr_inv_embed` kernel, ones biases, zero
    attributes), set_weights / get_weights, the reference's return structure, and the oracle's numbers."""
    import feedback_gnn_amd as F
    c = code("gb48")
    dec = F.GNN_BP4(c, num_embed_dims=12, num_msg_dims=99, num_hidden_units=24, num_mlp_layers=3, num_iter=3, reduce_op="sum",
                    activation="sigmoid", use_bias=True, use_attributes=True, node_attribute_dims=3, msg_attribute_dims=2)
    cfg = (12, 24, 3, "sum", "sigmoid", True, True, 3, 2)
    assert dec.config == cfg
    w0 = dec.get_weights()
    assert len(w0) == (7 * 3 + 1) * 2 + 7 and np.all(w0[42] == 0) and np.all(w0[43] == 1) and all(np.all(a == 0) for a in w0[44:])
    w = _gnnbp4_gen_weights(dec.graph, cfg, seed=5)
    dec.set_weights(w)
    og = oracle_library_forms("gb48")
    ex, ez = og.pauli_noise(SEED, 0.05, 0, 6)
    sx, sz = og.syndrome(ex, ez)
    llr_hat, x_hat, z_hat = dec((to_gpu(sx.astype(np.int64)), to_gpu(sz.astype(np.int64))))
    o = og.gnn_bp4_general(_gnnbp4_cfg_codes(cfg), w, sx, sz, 3)
    assert len(llr_hat) == 3 and llr_hat[0][0].shape == (og.m_z + og.rows_lz, 6) and x_hat.shape == (c.N, 6)
    assert np.array_equal(o["x_logit_all"][2].T, llr_hat[2][0].cpu().numpy()) and np.array_equal(o["z_logit_all"][0].T, llr_hat[0][1].cpu().numpy())
    assert np.array_equal(o["x_hat"].T, x_hat.cpu().numpy())


def test_gnn_bp4_mfma_and_valu_kernels_agree():
    from feedback_gnn_amd.graph import GnnBp4Weights
    name, B, iters = "ghp882", 5, 3
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.05, B, first=9)
    w = _gnnbp4_weights(21)
    o = og.gnn_bp4(w, sx, sz, iters)
    gw = GnnBp4Weights(w, gg.device)
    a = gg.gnn_bp4_decode(gw, tx, tz, iters)
    gg.force_generic(True)
    try:
        b = gg.gnn_bp4_decode(gw, tx, tz, iters)
    finally:
        gg.force_generic(False)
    # codes whose two phases' operand tables do not fit LDS together stage them per phase: same bits (FGNN_GNNBP4_NO_RESIDENT forces it)
    os.environ["FGNN_GNNBP4_NO_RESIDENT"] = "1"
    try:
        c = gg.gnn_bp4_decode(gw, tx, tz, iters)
    finally:
        del os.environ["FGNN_GNNBP4_NO_RESIDENT"]
    for k in ("llr", "x_logit_all", "z_logit_all", "x_hat", "z_hat"):
        assert np.array_equal(o[k], b[k].cpu().numpy()), f"VALU {k}"
        assert np.array_equal(o[k], c[k].cpu().numpy()), f"MFMA, tables staged per phase: {k}"
        assert np.array_equal(o[k], a[k].cpu().numpy()), f"MFMA {k}: {np.abs(o[k].astype(np.float64) - a[k].cpu().numpy()).max()}"


@pytest.mark.parametrize("name", ["ghp882", "gb48", "rsurf5"])
@pytest.mark.parametrize("cn_type,factor", [("boxplus-phi", 1.0), ("minsum", 0.8), ("boxplus", 0.625)])
def test_binary_syndrome_bp_bit_exact(name, cn_type, factor):
    """LDPCBPDecoder(is_syndrome=True) on hx (decoding.py:874-1048) — kernel vs oracle, soft logits and hard decisions."""
    B = 37
    og, gg = oracle_library_forms(name), gpu_graph(name)
    e = og.bsc_noise(SEED, 0.04, 0, B)
    ge = gg.bsc_noise(SEED, 0.04, 0, B)
    assert np.array_equal(e, ge.cpu().numpy())
    synd = (e.astype(np.int64) @ code(name).hx.T % 2).astype(np.uint8)
    L = float(-np.log((np.float32(1) - np.float32(0.2)) / np.float32(0.2), dtype=np.float32))
    s0, h0 = og.bp2_decode(synd, 24, cn_type, factor, llr_const=L)
    s1, h1 = gg.bp2_decode(to_gpu(synd), 24, cn_type, factor, llr_const=L)
    assert np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(h0, h1.cpu().numpy())
    llr = np.random.RandomState(2).uniform(-25, 3, size=(B, og.n)).astype(np.float32)  # exercises the +-20 clip
    s0, h0 = og.bp2_decode(synd, 7, cn_type, factor, llr_ch=llr)
    s1, h1 = gg.bp2_decode(to_gpu(synd), 7, cn_type, factor, llr_ch=to_gpu(llr))
    assert np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(h0, h1.cpu().numpy())


@pytest.mark.parametrize("cn_type,factor", [("boxplus-phi", 1.0), ("boxplus-phi", 0.875), ("minsum", 0.8)])
def test_binary_syndrome_bp_regular_kernel_equals_generic(cn_type, factor):
    """The (3,6)-regular register-resident check update of fgnn_bp2.hip vs the runtime-degree kernel (itself pinned to the oracle
    above) on 4096 syndromes x 64 iterations of [[882,24]] hx, noise on both sides of the waterfall: same bits."""
    gg = gpu_graph("ghp882")
    B = 4096
    for p in (0.02, 0.08):
        e = gg.bsc_noise(SEED + 5, p, 0, B)
        sx, _ = gg.syndrome(torch.zeros_like(e), e)
        L = float(np.log((1 - 0.2) / 0.2))
        s1, h1 = gg.bp2_decode(sx, 64, cn_type, factor, llr_const=L)
        gg.force_generic(True)
        try:
            s0, h0 = gg.bp2_decode(sx, 64, cn_type, factor, llr_const=L)  # runtime degrees, predicated register-resident update
            os.environ["FGNN_BP2_NO_PRED"] = "1"
            try:
                s2, h2 = gg.bp2_decode(sx, 64, cn_type, factor, llr_const=L)  # runtime degrees, the loop over the slot list
            finally:
                del os.environ["FGNN_BP2_NO_PRED"]
        finally:
            gg.force_generic(False)
        assert torch.equal(s0.view(torch.int32), s1.view(torch.int32)) and torch.equal(h0, h1)
        assert torch.equal(s2.view(torch.int32), s1.view(torch.int32)) and torch.equal(h2, h1)


def test_bp_bsc_model_contract_and_published_band():
    """examples/QLDPC.ipynb cell 7: binary BP-64 on hx of [[882,24]] over a BSC, p0=0.2, logical_pcm=hz_perp;
    row p=0.047 -> 798/10000 flagged = BLER (binomial band, independent draw)."""
    import feedback_gnn_amd as F
    c = code("ghp882")
    dec = F.LDPCBPDecoder(c.hx, is_syndrome=True, num_iter=64)
    model = F.BP_BSC_Model(pcm=c.hx, decoder=dec, logical_pcm=c.hz_perp, p0=0.2)
    p = 0.07 * 2 / 3
    s_hat, ls_hat = model(10000, p)
    assert s_hat.shape == (10000, 441) and ls_hat.shape == (10000, 453)
    fl, bl = float(s_hat.any(1).float().mean()), float(ls_hat.any(1).float().mean())
    assert abs(bl - 0.0798) < 4 * np.sqrt(0.0798 * 0.92 / 10000) * np.sqrt(2), bl
    assert fl <= bl + 1e-9
    noise, noise_hat = F.BP_BSC_Model(pcm=c.hx, decoder=dec, p0=0.2)(64, 0.02)
    assert noise.shape == (64, 882) and noise_hat.shape == (64, 882)
    # decoder layer contract: (llr[bs,n] logits, syndrome[m,bs]) -> hard decisions
    e = noise.to(torch.uint8)
    synd = torch.from_numpy((e.cpu().numpy().astype(np.int64) @ c.hx.T % 2)).cuda()
    llr = torch.full((64, 882), float(-np.log(0.8 / 0.2)), dtype=torch.float32, device="cuda")
    out = dec((llr, synd.t()))
    assert torch.equal(out, noise_hat)


@pytest.mark.parametrize("name,p", [("ghp882", 0.10), ("gb48", 0.09), ("ghp1270", 0.10)])
def test_osd0_bit_exact_and_solves_the_syndrome(name, p):
    """OSD-0 (bp_osd.py:14-77) on the BP failures: kernel vs oracle, and H e_hat = syndrome for every processed sample."""
    c = code(name)
    B = 256
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, p, B, first=5)
    L0 = llr_const(p)
    o = og.bp4_decode(sx, sz, 30, "minsum", 0.8, llr_const=L0)
    g = gg.bp4_decode(tx, tz, 30, "minsum", 0.8, llr_const=L0, want_logits=False)
    fl = og.residual(ex, ez, o["x_hat"], o["z_hat"])[2]
    idx = np.nonzero(fl & 1)[0].astype(np.int32)
    assert len(idx) >= 3, "test point must produce BP failures"
    gfl = gg.residual(gx, gz, g["x_hat"], g["z_hat"], want_arrays=False)[2]
    gi, nact = gg.compact(gfl, 1)
    assert nact == len(idx) and sorted(gi[:nact].cpu().numpy().tolist()) == idx.tolist()
    gg.set_basis(0, c.pivot_hx)
    gg.set_basis(1, c.pivot_hz)
    zh, xh = o["z_hat"].copy(), o["x_hat"].copy()
    og.osd0(0, c.pivot_hx, sx, marg=o["llr"], index=idx, e_hat=zh)
    og.osd0(1, c.pivot_hz, sz, marg=o["llr"], index=idx, e_hat=xh)
    gg.osd0(0, tx, g["z_hat"], marg=g["llr"], index=gi, nact=nact)
    gg.osd0(1, tz, g["x_hat"], marg=g["llr"], index=gi, nact=nact)
    assert np.array_equal(zh, g["z_hat"].cpu().numpy()) and np.array_equal(xh, g["x_hat"].cpu().numpy())
    assert np.array_equal(zh[idx].astype(np.int64) @ c.hx.T % 2, sx[idx]) and np.array_equal(xh[idx].astype(np.int64) @ c.hz.T % 2, sz[idx])
    # ties: saturated marginals give many equal reliabilities — run OSD on a converged sample too (stable order must agree)
    ok = np.nonzero((fl & 1) == 0)[0][:4].astype(np.int32)
    z2, x2 = o["z_hat"].copy(), o["x_hat"].copy()
    og.osd0(0, c.pivot_hx, sx, marg=o["llr"], index=ok, e_hat=z2)
    gz2 = g["z_hat"].clone()
    gg.osd0(0, tx, gz2, marg=g["llr"], index=to_gpu(ok), nact=len(ok))
    assert np.array_equal(z2[ok], gz2.cpu().numpy()[ok])
    assert np.array_equal(z2[ok].astype(np.int64) @ c.hx.T % 2, sx[ok])


def test_bp4_osd_model_published_band():
    """examples/OSD.ipynb cell 2: [[882,24]], min-sum BP4 100 it factor 0.8 + OSD-0: p=0.10 -> BLER 3.70e-4 (111/300 000)."""
    import feedback_gnn_amd as F
    c = code("ghp882")
    dec = F.QLDPCBPDecoder(code=c, num_iter=100, normalization_factor=0.8, cn_type="minsum", stage_one=True)
    model = F.BP4_OSD_Model(c, dec, F.OSD0_Decoder(c.N))
    zeros, ls_hat = model(300000, 0.10)
    assert ls_hat.shape == (300000, 48) and not bool(zeros.any())
    errs = int(ls_hat.any(1).sum())
    # Poisson band around 111 events (two independent draws): 4 sigma
    assert abs(errs - 111) < 4 * np.sqrt(2 * 111), errs
    assert 0.002 < model.last_num_osd / 300000 < 0.05  # BP failure rate at p=0.10 is ~1 %


def test_bp2_osd_model_published_band():
    """examples/OSD.ipynb cell 3: binary min-sum BP 100 it (factor 0.8, soft output) + OSD-0 on hx of [[882,24]] over a BSC,
    logical_pcm = lx: p=0.05 -> BLER 5.85e-4 (117/200 000)."""
    import feedback_gnn_amd as F
    c = code("ghp882")
    bp2 = F.LDPCBPDecoder(c.hx, is_syndrome=True, hard_out=False, cn_type="minsum", num_iter=100, normalization_factor=0.8)
    model = F.BP2_OSD_Model(c.hx, c.hx_basis, c.pivot_hx, c.lx, bp2, F.OSD0_Decoder(c.N))
    zeros, ls_hat = model(200000, 0.05)
    assert ls_hat.shape == (200000, 24) and not bool(zeros.any())
    errs = int(ls_hat.any(1).sum())
    assert abs(errs - 117) < 4 * np.sqrt(2 * 117), errs
    assert model.last_num_osd > 0


def test_overcomplete_irregular_graph():
    """Over-complete check matrices (the reference's GB_*_H_*.alist use case, QLDPC.ipynb cell 5): redundant rows give
    irregular degrees and qubits with tens of checks; the CSR kernels must still match the oracle bit for bit."""
    from feedback_gnn_amd import codes_q as cq
    from feedback_gnn_amd.graph import TannerGraph
    from oracle.oracle import OracleGraph
    base = code("gb48")
    rng = np.random.RandomState(4)

    def overcomplete(h, rows):
        out = [h]
        while sum(x.shape[0] for x in out) < rows:
            pick = rng.choice(h.shape[0], size=rng.randint(2, 4), replace=False)
            out.append((h[pick].sum(0) % 2)[None, :])
        return np.vstack(out)[:rows]

    c = cq.css_code(overcomplete(base.hx, 150), overcomplete(base.hz, 130), name="gb48_oc")
    assert c.K == base.K and len(set(c.hx.sum(1))) > 2 and c.hx.sum(0).max() > 20
    og, gg = OracleGraph(c, forms="library-default"), TannerGraph(c)
    assert gg.info()["regular"] == 0
    ex, ez = og.pauli_noise(SEED, 0.06, 0, 50)
    sx, sz = og.syndrome(ex, ez)
    for cn_type in ("boxplus-phi", "minsum"):
        o = og.bp4_decode(sx, sz, 6, cn_type, 1.0, llr_const=llr_const(0.3), return_msgs=True)
        g = gg.bp4_decode(to_gpu(sx), to_gpu(sz), 6, cn_type, 1.0, llr_const=llr_const(0.3), return_msgs=True)
        _assert_bp_equal(o, g, f"overcomplete {cn_type}")
    s0, l0, f0 = og.residual(ex, ez, o["x_hat"], o["z_hat"])
    s1, l1, f1 = gg.residual(to_gpu(ex), to_gpu(ez), g["x_hat"], g["z_hat"])
    assert np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(f0, f1.cpu().numpy())


def test_north_star_full_size_properties_and_strided_oracle_check():
    """BASELINE.json's north-star size: [[882,24]], batch 65 536, p = 0.01, sandwich (64, G, 16).  The oracle cannot
    decode 65 536 samples in seconds, so the full batch is checked through properties that do not depend on the size —
    (i) a strided subset (every 128th sample) equals the oracle bit for bit, (ii) the decision depends on the syndrome
    only: adding a stabilizer to the error leaves it unchanged, (iii) samples the sandwich does not flag satisfy both
    syndrome equations, (iv) a permuted batch gives permuted results, (v) the device counters equal the host count."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    name, B, p = "ghp882", 65536, 0.01
    og, gg = oracle_library_forms(name), gpu_graph(name)
    c = code(name)
    w = read_weight_list(WEIGHTS_882)
    gw = GnnWeights(w, gg.device)
    L0 = llr_const(0.05)
    ex, ez = gg.pauli_noise(SEED, p, 0, B)
    sx, sz = gg.syndrome(ex, ez)
    full = gg.sandwich_decode(sx, sz, [64, 16], [gw], L0, return_llr=True, return_rounds=True)
    # (i) strided subset against the oracle (same Philox samples: index = global sample index)
    idx = torch.arange(0, B, 128, device=gg.device)
    o = og.sandwich_decode(sx[idx].cpu().numpy(), sz[idx].cpu().numpy(), [64, 16], [w], L0, return_llr=True)
    assert np.array_equal(o["x_hat"], full["x_hat"][idx].cpu().numpy())
    assert np.array_equal(o["z_hat"], full["z_hat"][idx].cpu().numpy())
    assert np.array_equal(o["llr"], full["llr"][idx].cpu().numpy())
    assert np.array_equal(o["rounds"], full["rounds"][idx].cpu().numpy())
    # (ii) e and e + stabilizer share the syndrome, hence the decision: X-type stabilizers are rows of hx, Z-type rows of hz
    rng = np.random.RandomState(0)
    rows_x = torch.from_numpy(c.hx[rng.randint(0, c.hx.shape[0], size=B)].astype(np.uint8)).to(gg.device)
    rows_z = torch.from_numpy(c.hz[rng.randint(0, c.hz.shape[0], size=B)].astype(np.uint8)).to(gg.device)
    sx2, sz2 = gg.syndrome(ex ^ rows_x, ez ^ rows_z)
    assert torch.equal(sx, sx2) and torch.equal(sz, sz2)
    # (iii) unflagged samples reproduce their syndromes; flags/counters consistent
    s_hat, ls_hat, flags = gg.residual(ex, ez, full["x_hat"], full["z_hat"])
    ok = (flags & 1) == 0
    hx_t = torch.from_numpy(c.hx.astype(np.float32)).to(gg.device)
    hz_t = torch.from_numpy(c.hz.astype(np.float32)).to(gg.device)
    syn_of_zhat = (full["z_hat"].float() @ hx_t.t()).remainder(2).to(torch.uint8)
    syn_of_xhat = (full["x_hat"].float() @ hz_t.t()).remainder(2).to(torch.uint8)
    assert torch.equal(syn_of_zhat[ok], sx[ok]) and torch.equal(syn_of_xhat[ok], sz[ok])
    assert bool((s_hat[ok] == 0).all()) and bool((((flags >> 1) & 1) >= (flags & 1)).all())  # flagged implies block error
    counts = torch.zeros(3, dtype=torch.int64, device=gg.device)
    gg.count_flags(flags, counts)
    assert counts.tolist() == [int((flags & 1).sum()), int(((flags >> 1) & 1).sum()), B]
    assert int(ok.sum()) > 0.999 * B  # p = 0.01 is deep in the waterfall
    # (iv) permutation equivariance over the whole batch (workgroup placement must not matter)
    perm = torch.randperm(B, device=gg.device, generator=torch.Generator(device=gg.device).manual_seed(1))
    pr = gg.sandwich_decode(sx[perm].contiguous(), sz[perm].contiguous(), [64, 16], [gw], L0, return_llr=True)
    assert torch.equal(pr["x_hat"], full["x_hat"][perm]) and torch.equal(pr["z_hat"], full["z_hat"][perm])
    assert torch.equal(pr["llr"], full["llr"][perm])
    # (v) the compacted driver (feedback rounds on flagged samples only) gives the same decisions at full size
    cp = gg.sandwich_decode(sx, sz, [64, 16], [gw], L0, compact=True)
    assert torch.equal(cp["x_hat"], full["x_hat"]) and torch.equal(cp["z_hat"], full["z_hat"])


def test_config3_full_shard_properties_and_strided_oracle_check():
    """BASELINE.json configs[3]: [[1270,28]], sandwich (64, G, 64) with the trained weights (/root/reference n1270.py:37,57), at
    the per-GPU shard of the 8-GPU run: 262 144 / 8 = 32 768 codewords.  Same size-independent properties as the north-star test:
    (i) every 256th sample equals the oracle bit for bit (decisions, marginals, rounds), (ii) unflagged samples reproduce both
    syndromes and flags imply block errors, (iii) a permuted batch gives permuted results, (iv) compact == full, (v) the shard
    decoded as rank 3 of 8 of the global Philox stream equals the same samples decoded in one piece."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    name, B, p = "ghp1270", 32768, 0.05
    og, gg = oracle_library_forms(name), gpu_graph(name)
    c = code(name)
    w = read_weight_list(WEIGHTS_1270)
    gw = GnnWeights(w, gg.device)
    L0 = llr_const(0.05)
    first = 3 * B  # rank 3 of 8: global samples [3B, 4B)
    ex, ez = gg.pauli_noise(SEED, p, first, B)
    sx, sz = gg.syndrome(ex, ez)
    full = gg.sandwich_decode(sx, sz, [64, 64], [gw], L0, return_llr=True, return_rounds=True)
    idx = torch.arange(0, B, 256, device=gg.device)
    oex, oez = og.pauli_noise(SEED, p, first, B)  # the oracle's own Philox stream: same global sample indices
    assert np.array_equal(oex[::256], ex[idx].cpu().numpy()) and np.array_equal(oez[::256], ez[idx].cpu().numpy())
    osx, osz = og.syndrome(oex[::256], oez[::256])
    o = og.sandwich_decode(osx, osz, [64, 64], [w], L0, return_llr=True)
    assert np.array_equal(o["x_hat"], full["x_hat"][idx].cpu().numpy())
    assert np.array_equal(o["z_hat"], full["z_hat"][idx].cpu().numpy())
    assert np.array_equal(o["llr"], full["llr"][idx].cpu().numpy())
    assert np.array_equal(o["rounds"], full["rounds"][idx].cpu().numpy())
    assert int(full["rounds"].sum()) > 0, "p = 0.05 must send some samples through the feedback round"
    s_hat, ls_hat, flags = gg.residual(ex, ez, full["x_hat"], full["z_hat"])
    ok = (flags & 1) == 0
    hx_t = torch.from_numpy(c.hx.astype(np.float32)).to(gg.device)
    hz_t = torch.from_numpy(c.hz.astype(np.float32)).to(gg.device)
    assert torch.equal((full["z_hat"].float() @ hx_t.t()).remainder(2).to(torch.uint8)[ok], sx[ok])
    assert torch.equal((full["x_hat"].float() @ hz_t.t()).remainder(2).to(torch.uint8)[ok], sz[ok])
    assert bool((s_hat[ok] == 0).all()) and bool((((flags >> 1) & 1) >= (flags & 1)).all())
    assert int(ok.sum()) > 0.99 * B
    perm = torch.randperm(B, device=gg.device, generator=torch.Generator(device=gg.device).manual_seed(2))
    pr = gg.sandwich_decode(sx[perm].contiguous(), sz[perm].contiguous(), [64, 64], [gw], L0, return_llr=True)
    assert torch.equal(pr["x_hat"], full["x_hat"][perm]) and torch.equal(pr["z_hat"], full["z_hat"][perm])
    assert torch.equal(pr["llr"], full["llr"][perm])
    cp = gg.sandwich_decode(sx, sz, [64, 64], [gw], L0, compact=True)
    assert torch.equal(cp["x_hat"], full["x_hat"]) and torch.equal(cp["z_hat"], full["z_hat"])
    # (v) sharding: the model class with rank = 3, world_size = 8 draws exactly these samples for its first batch
    import feedback_gnn_amd as F
    d0 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=gg)
    G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh",
                       use_bias=True, graph=gg)
    F.load_weights(G, WEIGHTS_1270)
    m = F.Sandwich_BP_GNN_Evaluation_Model(c, [d0, d0], [G], num_layers=2, rank=3, world_size=8, seed=SEED)
    d = m.decode(B, p)
    assert torch.equal(d["noise_x"], ex) and torch.equal(d["x_hat"], full["x_hat"]) and torch.equal(d["z_hat"], full["z_hat"])


def test_config4_full_shard_gnn_bp4_properties_and_strided_oracle_check():
    """BASELINE.json configs[4]: [[1270,28]] GNN_BP4 (/root/reference sionna/fec/ldpc/gnn.py:383-423, repaired arity), 10
    iterations, at the per-GPU shard of the 8-GPU run: 131 072 / 8 = 16 384 codewords.  (i) every 512th sample (32 samples)
    equals the oracle bit for bit — marginals, hard decisions and the soft syndromes of every iteration; (ii) a permuted batch
    gives permuted results; (iii) the decoder is a function of the syndrome alone: duplicated syndromes give duplicated rows."""
    from feedback_gnn_amd.graph import GnnBp4Weights
    name, B, iters = "ghp1270", 16384, 10
    og, gg = oracle_library_forms(name), gpu_graph(name)
    w = _gnnbp4_weights(5)
    gw = GnnBp4Weights(w, gg.device)
    ex, ez = gg.pauli_noise(SEED, 0.05, 7 * B, B)
    sx, sz = gg.syndrome(ex, ez)
    sx[1], sz[1] = sx[0], sz[0]  # (iii)
    full = gg.gnn_bp4_decode(gw, sx, sz, iters)
    idx = torch.arange(0, B, 512, device=gg.device)
    o = og.gnn_bp4(w, sx[idx].cpu().numpy(), sz[idx].cpu().numpy(), iters)
    for k in ("llr", "x_hat", "z_hat"):
        assert np.array_equal(o[k], full[k][idx].cpu().numpy()), k
    for k in ("x_logit_all", "z_logit_all"):
        assert np.array_equal(o[k], full[k][:, idx].cpu().numpy()), k
    assert torch.equal(full["llr"][0], full["llr"][1]) and torch.equal(full["x_hat"][0], full["x_hat"][1])
    assert bool(torch.isfinite(full["llr"]).all())
    perm = torch.randperm(B, device=gg.device, generator=torch.Generator(device=gg.device).manual_seed(3))
    pr = gg.gnn_bp4_decode(gw, sx[perm].contiguous(), sz[perm].contiguous(), iters, return_logits=False)
    assert torch.equal(pr["llr"], full["llr"][perm]) and torch.equal(pr["x_hat"], full["x_hat"][perm])
    assert torch.equal(pr["z_hat"], full["z_hat"][perm])


@pytest.mark.parametrize("name", ["gb46_oc", "gb48_oc"])
@pytest.mark.parametrize("cn_type", ["boxplus-phi", "minsum"])
def test_reference_overcomplete_codes_bit_exact(name, cn_type):
    """The reference's over-complete GB matrices (QLDPC.ipynb cell 5; 400+400 / 1000+1000 checks on 46 / 48 qubits, row
    weights 8-12, column weights up to 258): runtime-degree kernel, 6 iterations as in cell 11, against the oracle."""
    B = 37
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.08, B)
    L0 = llr_const(0.3)
    o = oracle_library_forms(name).bp4_decode(sx, sz, 6, cn_type, 1.0, llr_const=L0, return_msgs=True)
    g = gpu_graph(name).bp4_decode(tx, tz, 6, cn_type, 1.0, llr_const=L0, return_msgs=True)
    _assert_bp_equal(o, g, f"{name} {cn_type}")
    s0, l0, f0 = oracle_library_forms(name).residual(ex, ez, o["x_hat"], o["z_hat"])
    s1, l1, f1 = gpu_graph(name).residual(gx, gz, g["x_hat"], g["z_hat"])
    assert np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(l0, l1.cpu().numpy()) and np.array_equal(f0, f1.cpu().numpy())


PUBLISHED_BP4_ROWS = [
    # (code, iterations, factor, p0, p, flag errors, block errors, num blocks) — examples/QLDPC.ipynb cell 12
    ("gb48_oc", 6, 1.0, 0.3, 0.10, 746, 1673, 10000), ("gb48_oc", 6, 1.0, 0.3, 0.06, 63, 217, 10000),
    ("gb46_oc", 6, 1.0, 0.3, 0.10, 505, 892, 10000), ("gb46_oc", 6, 1.0, 0.3, 0.07, 82, 191, 10000),
    ("gb254", 64, 0.625, 0.1, 0.09, 2786, 2786, 10000), ("gb254", 64, 0.625, 0.1, 0.07, 425, 425, 10000),
    ("gb126", 64, 0.8, 0.1, 0.10, 4786, 4829, 10000), ("gb126", 64, 0.8, 0.1, 0.06, 533, 553, 10000),
    ("ghp882", 64, 0.8, 0.3, 0.10, 4385, 4385, 10000), ("ghp882", 64, 0.8, 0.3, 0.09, 1318, 1318, 10000),
    ("ghp882", 64, 0.8, 0.3, 0.08, 147, 147, 10000),
    ("ghp1270", 64, 0.8, 0.3, 0.10, 4813, 4813, 10000), ("ghp1270", 64, 0.8, 0.3, 0.09, 1027, 1027, 10000),
]


@pytest.mark.parametrize("name,iters,factor,p0,p,flagged,block,total", PUBLISHED_BP4_ROWS)
def test_published_rows_plain_bp4(name, iters, factor, p0, p, flagged, block, total):
    """QLDPC.ipynb cell 12 (plain flooding BP4, boxplus-phi; the over-complete GB codes, GB_n254, GB_n126 with unflagged
    logical errors, and both GHP codes at factor 0.8 / p0 = 0.3): flagged and logical error counts of the published rows
    inside binomial 4-sigma bands (two independent draws), on 4x the published sample count."""
    from feedback_gnn_amd import QLDPCBPDecoder, Sandwich_BP_GNN_Evaluation_Model
    c = code(name)
    dec = QLDPCBPDecoder(code=c, num_iter=iters, normalization_factor=factor, cn_type="boxplus-phi", stage_one=True,
                         graph=gpu_graph(name))
    m = Sandwich_BP_GNN_Evaluation_Model(c, [dec], [], num_layers=1, p0=p0)
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    n = 4 * total
    m.mc_step(n, p, counts)
    fl, bl, tot = [int(v) for v in counts.cpu()]
    assert tot == n
    for got, pub in ((fl, flagged), (bl, block)):
        r = pub / total
        sigma = np.sqrt(r * (1 - r) * (1 / total + 1 / n))
        assert abs(got / n - r) < 4 * sigma + 2 / total, (name, p, got / n, r)


GEN_CONFIGS = [(20, 40, 2, "mean", "tanh", True), (8, 16, 1, "max", "relu", False), (12, 24, 3, "sum", "sigmoid", True),
               (5, 7, 2, "min", None, True), (32, 96, 4, "mean", "relu", False)]


def _gen_weights(cfg, seed=3):
    from feedback_gnn_amd.graph import gnn_weight_shapes
    rng = np.random.RandomState(seed)
    return [rng.uniform(-0.5, 0.5, size=s).astype(np.float32) for s in gnn_weight_shapes(cfg[0], cfg[1], cfg[2], cfg[5])]


def _cfg_codes(cfg):
    from feedback_gnn_amd.graph import ACTIVATIONS, REDUCE_OPS
    return (cfg[0], cfg[1], cfg[2], REDUCE_OPS[cfg[3]], ACTIVATIONS[cfg[4]], int(cfg[5]))


@pytest.mark.parametrize("cfg", GEN_CONFIGS)
@pytest.mark.parametrize("name", ["rsurf5", "ghp882"])
def test_general_feedback_gnn_bit_exact(name, cfg):
    """Feedback_GNN with any constructor setting (fgnn_weights_create_general, runtime-shaped kernel) against the oracle."""
    from feedback_gnn_amd.graph import GnnWeights
    B = 21
    og, gg = oracle_library_forms(name), gpu_graph(name)
    (ex, ez, sx, sz), (gx, gz, tx, tz) = _noise_and_syndromes(name, 0.08, B)
    o = og.bp4_decode(sx, sz, 6, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    w = _gen_weights(cfg)
    ref = og.feedback_gnn_general(_cfg_codes(cfg), w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    gw = GnnWeights(w, gg.device, cfg, force_general=True)
    got = gg.feedback_gnn(gw, to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), tx, tz).cpu().numpy()
    assert np.array_equal(ref, got), np.abs(ref - got).max()
    if cfg == GEN_CONFIGS[0]:  # the shipped setting: the runtime-shaped kernel (always the literal association) and the specialised
        # kernels in the literal order (the library default) agree bit for bit; the opt-in factored order is the same function with other roundings
        args = (GnnWeights(w, gg.device), to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), tx, tz)
        gg.set_gnn_factored(True)
        try:
            fact = gg.feedback_gnn(*args).cpu().numpy()
            gg.set_gnn_factored(False)
            sp = gg.feedback_gnn(*args).cpu().numpy()
        finally:
            gg.set_gnn_factored(LIBRARY_GNN_FACTORED)
        assert np.array_equal(got, sp)
        assert np.abs(fact - sp).max() <= 1e-5 * max(1.0, np.abs(sp).max())


def test_general_feedback_gnn_in_the_sandwich_and_class_surface():
    """A non-shipped Feedback_GNN inside Sandwich_BP_GNN_Evaluation_Model equals the composition of the oracle's stages with the
    per-sample masking of feedback_gnn.py:324-340; weights round-trip through get_weights / set_weights / save / load."""
    import os
    import tempfile
    from feedback_gnn_amd import QLDPCBPDecoder, Feedback_GNN, Sandwich_BP_GNN_Evaluation_Model, load_weights, save_weights
    name, B, p = "ghp882", 64, 0.11
    cfg = (8, 16, 3, "max", "relu", True)
    c = code(name)
    og, gg = oracle_library_forms(name), gpu_graph(name)
    G = Feedback_GNN(code=c, num_msg_dims=8, num_hidden_units=16, num_mlp_layers=3, reduce_op="max", activation="relu",
                     use_bias=True, graph=gg)
    assert not G.is_shipped_architecture and G.count_params() == sum(int(np.prod(a.shape)) for a in G.get_weights())
    w = _gen_weights(cfg, seed=8)
    w[1] = w[1] + 3.0  # keep the new channel LLRs in a sane range
    G.set_weights(w)
    with tempfile.TemporaryDirectory() as d:
        save_weights(G, os.path.join(d, "g.npz"))
        G2 = Feedback_GNN(code=c, num_msg_dims=8, num_hidden_units=16, num_mlp_layers=3, reduce_op="max", activation="relu",
                          use_bias=True, graph=gg)
        load_weights(G2, os.path.join(d, "g.npz"))
        assert all(np.array_equal(a, b) for a, b in zip(G2.get_weights(), w))
    with pytest.raises(ValueError):
        Feedback_GNN(code=c, num_msg_dims=8, num_hidden_units=16, num_mlp_layers=3, reduce_op="median", graph=gg)
    with pytest.raises(NotImplementedError):
        Feedback_GNN(code=c, num_msg_dims=8, num_hidden_units=16, num_mlp_layers=3, activation="gelu", graph=gg)
    d1 = QLDPCBPDecoder(code=c, num_iter=20, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=gg)
    d2 = QLDPCBPDecoder(code=c, num_iter=8, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=gg)
    m = Sandwich_BP_GNN_Evaluation_Model(c, [d1, d2], [G], num_layers=2, seed=SEED)
    out = m.decode(B, p, first_sample=0)
    ex, ez = og.pauli_noise(SEED, p, 0, B)
    sx, sz = og.syndrome(ex, ez)
    L0 = llr_const(0.05)
    o1 = og.bp4_decode(sx, sz, 20, "boxplus-phi", 1.0, llr_const=L0)
    rx, rz = og.syndrome(o1["x_hat"], o1["z_hat"])  # (hx z_hat, hz x_hat)
    bad = (rx != sx).any(1) | (rz != sz).any(1)
    new = og.feedback_gnn_general(_cfg_codes(cfg), w, o1["llr"], o1["z_logit"], o1["x_logit"], sx, sz)
    o2 = og.bp4_decode(sx, sz, 8, "boxplus-phi", 1.0, llr_ch=new)
    xh = np.where(bad[:, None], o2["x_hat"], o1["x_hat"])
    zh = np.where(bad[:, None], o2["z_hat"], o1["z_hat"])
    assert bad.sum() > 3
    assert np.array_equal(out["x_hat"].cpu().numpy(), xh) and np.array_equal(out["z_hat"].cpu().numpy(), zh)


def test_maximum_batch_of_configs3_in_one_launch_has_no_index_overflow():
    """BASELINE configs[3] decodes 262 144 codewords of [[1270,28]] (there: 32 768 per GPU).  One launch over ALL of them on one GPU
    — 1.0e9 floats of marginals, 2.0e9 bytes of message tape addressing in the trace kernel's index arithmetic — must give, for the
    first, the last and a strided selection of codewords, exactly what a small launch over those same Philox samples gives (any
    32-bit overflow in an index would land somewhere else)."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    name, B = "ghp1270", 262144
    gg = gpu_graph(name)
    L0 = llr_const(0.05)
    gw = GnnWeights(read_weight_list(WEIGHTS_1270), gg.device)
    ex, ez = gg.pauli_noise(SEED, 0.08, 0, B)
    sx, sz = gg.syndrome(ex, ez)
    big = gg.bp4_decode(sx, sz, 6, "boxplus-phi", 1.0, llr_const=L0)
    big_gnn = gg.feedback_gnn(gw, big["llr"], big["z_logit"], big["x_logit"], sx, sz)
    big2 = gg.bp4_decode(sx, sz, 3, "boxplus-phi", 1.0, llr_ch=big_gnn, want_logits=False)
    fl_big = gg.residual(ex, ez, big2["x_hat"], big2["z_hat"], want_arrays=False)[2]
    sel = torch.tensor([0, 1, 65535, 65536, 131071, 200000, B - 2, B - 1], device=gg.device)
    for b in sel.tolist():
        e1, e2 = gg.pauli_noise(SEED, 0.08, b, 1)
        assert torch.equal(e1[0], ex[b]) and torch.equal(e2[0], ez[b])
        s1, s2 = gg.syndrome(e1, e2)
        o = gg.bp4_decode(s1, s2, 6, "boxplus-phi", 1.0, llr_const=L0)
        for k in ("llr", "x_hat", "z_hat", "x_logit", "z_logit"):
            assert torch.equal(o[k][0], big[k][b]), (b, k)
        gn = gg.feedback_gnn(gw, o["llr"], o["z_logit"], o["x_logit"], s1, s2)
        assert torch.equal(gn[0], big_gnn[b]), (b, "gnn")
        o2 = gg.bp4_decode(s1, s2, 3, "boxplus-phi", 1.0, llr_ch=gn, want_logits=False)
        assert torch.equal(o2["x_hat"][0], big2["x_hat"][b]) and torch.equal(o2["llr"][0], big2["llr"][b])
        assert int(gg.residual(e1, e2, o2["x_hat"], o2["z_hat"], want_arrays=False)[2][0]) == int(fl_big[b])
    del big, big_gnn, big2
    # the trace kernel's [T+1, B, E] tape: 2 iterations over the first 131 072 codewords = 3 x 131 072 x 3 810 floats per side
    Bt = 131072
    tr = gg.bp4_decode_trace(sx[:Bt], sz[:Bt], 2, "boxplus-phi", 1.0, llr_const=L0, want_tape=True)
    for b in (0, 70000, Bt - 1):
        t1 = gg.bp4_decode_trace(sx[b:b + 1].contiguous(), sz[b:b + 1].contiguous(), 2, "boxplus-phi", 1.0, llr_const=L0, want_tape=True)
        for k in ("x_logit", "z_logit", "tape_x", "tape_z"):
            assert torch.equal(t1[k][:, 0], tr[k][:, b]), (b, k)
    torch.cuda.empty_cache()


def test_random_codes_fuzz_every_runtime_degree_kernel():
    """Fuzz of the runtime-degree paths: twelve seeded random generalized-bicycle codes (circulant size 5..40, 1..5 terms per
    circulant: qubit degrees 1..5 per side, check degrees 2..10, packed several codewords per workgroup), every check-node rule in both
    forms of the qubit update with random restarts, the one-launch trace, the feedback GNN and GNN_BP4 on their VALU kernels in both
    associations — each against the oracle, exactly."""
    from feedback_gnn_amd import codes_q as cq
    from feedback_gnn_amd.graph import GNNBP4_SHAPES, GnnBp4Weights, GnnWeights, TannerGraph
    from feedback_gnn_amd.weights_io import read_weight_list
    from oracle.oracle import OracleGraph
    rng = np.random.RandomState(20260)
    w = read_weight_list(WEIGHTS_882)
    wb = [rng.uniform(-0.4, 0.4, size=s).astype(np.float32) for s in GNNBP4_SHAPES]
    done = 0
    while done < 12:
        l = int(rng.randint(5, 41))
        a = sorted(set(rng.randint(0, l, size=rng.randint(1, 6)).tolist()))
        b = sorted(set(rng.randint(0, l, size=rng.randint(1, 6)).tolist()))
        c = cq.create_generalized_bicycle_codes(l, a, b)
        if c.K == 0:
            continue  # css_code derives no logical operators: GNN_BP4's logical rows would be empty — covered by the other 12
        done += 1
        og, gg = OracleGraph(c, forms="library-default"), TannerGraph(c)
        B = int(rng.randint(1, 40))
        ex, ez = og.pauli_noise(SEED, 0.07, 100 * done, B)
        sx, sz = og.syndrome(ex, ez)
        tx, tz = to_gpu(sx), to_gpu(sz)
        L0 = llr_const(0.05)
        llr = rng.uniform(0.3, 3.0, size=(B, 3, og.n)).astype(np.float32)
        init = (rng.uniform(-15, 15, size=(B, og.E_x)).astype(np.float32), rng.uniform(-15, 15, size=(B, og.E_z)).astype(np.float32))
        tag = f"GB l={l} a={a} b={b} B={B}"
        for cn, fac in (("boxplus-phi", 1.0), ("minsum", 0.625), ("boxplus", 0.8)):
            for lse in (0, 1):
                og.set_vn_shared_lse(lse)
                gg.set_bp4_shared_lse(lse)
                it = int(rng.randint(1, 9))
                o = og.bp4_decode(sx, sz, it, cn, fac, llr_const=L0, return_msgs=True)
                _assert_bp_equal(o, gg.bp4_decode(tx, tz, it, cn, fac, llr_const=L0, return_msgs=True), f"{tag} {cn} lse={lse}")
                o2 = og.bp4_decode(sx, sz, it, cn, fac, llr_ch=llr, msg_init=init, return_msgs=True)
                _assert_bp_equal(o2, gg.bp4_decode(tx, tz, it, cn, fac, llr_ch=to_gpu(llr), msg_init=tuple(to_gpu(x) for x in init),
                                                   return_msgs=True), f"{tag} {cn} lse={lse} restart")
                tr = gg.bp4_decode_trace(tx, tz, it, cn, fac, llr_ch=to_gpu(llr), msg_init=tuple(to_gpu(x) for x in init), want_tape=True)
                assert np.array_equal(o2["x_logit"], tr["x_logit"][it].cpu().numpy()) and np.array_equal(o2["msg_z"], tr["tape_z"][it].cpu().numpy()), tag
        og.set_vn_shared_lse(LIBRARY_BP4_SHARED_LSE)
        gg.set_bp4_shared_lse(LIBRARY_BP4_SHARED_LSE)
        o = og.bp4_decode(sx, sz, 6, "boxplus-phi", 1.0, llr_const=L0)
        for order in (0, 1):
            og.set_gnn_order(order)
            gg.set_gnn_factored(order)
            ref = og.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
            got = gg.feedback_gnn(GnnWeights(w, gg.device), to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), tx, tz)
            assert np.array_equal(ref, got.cpu().numpy()), f"{tag} feedback GNN order={order}"
            rb = og.gnn_bp4(wb, sx, sz, 2)
            gb = gg.gnn_bp4_decode(GnnBp4Weights(wb, gg.device), tx, tz, 2)
            for k in ("llr", "x_logit_all", "z_logit_all", "x_hat", "z_hat"):
                assert np.array_equal(rb[k], gb[k].cpu().numpy()), f"{tag} GNN_BP4 order={order} {k}"
        og.set_gnn_order(LIBRARY_GNN_FACTORED)
        gg.set_gnn_factored(LIBRARY_GNN_FACTORED)

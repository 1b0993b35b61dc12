"""FGNN_OPT_BP4_SHARED_LSE: the variable-node update with the (a - b)-dependent part of its log-sum-exp formed once per qubit and side.

decoding_q.py:254-273 evaluates reduce_logsumexp([-(Z - mu_e), -(Y - mu_e)]) on every hx edge e of a qubit; its second term,
log(1 + exp(-|(Z - mu_e) - (Y - mu_e)|)), has the same real argument Z - Y on all of them.  The option computes it once per side from
the unshifted totals.  The oracle restates both forms (og_graph_set_vn_shared_lse) and every kernel variant must equal it bit for bit
in either: compile-time and runtime degrees, constant and per-qubit channel LLRs (LDS and register variants), the exact saturation
shortcut and fixed-point exit on and off, all three check-node rules, restarts, the one-launch trace, the sandwich.  The two forms
are the same function: from identical messages one qubit update moves a message by a few ulp of the totals; converged samples
end on the same saturated fixed point.
"""
import numpy as np
import pytest
import torch

from helpers import WEIGHTS_882, WEIGHTS_1270, code, gpu_graph, llr_const, oracle_library_forms, to_gpu

pytestmark = pytest.mark.gpu
SEED = 0x5EED
KEYS = ("llr", "x_logit", "z_logit", "msg_x", "msg_z", "x_hat", "z_hat")


class _shared:
    def __init__(self, name, on):
        self.og, self.gg, self.on = oracle_library_forms(name), gpu_graph(name), on

    def __enter__(self):
        self.prev = self.gg.bp4_shared_lse
        self.og.set_vn_shared_lse(self.on)
        self.gg.set_bp4_shared_lse(self.on)
        return self.og, self.gg

    def __exit__(self, *exc):
        self.og.set_vn_shared_lse(self.prev)
        self.gg.set_bp4_shared_lse(self.prev)


def _eq(o, g, what):
    for k in KEYS:
        if k in o and o[k] is not None and g.get(k) is not None:
            a, b = o[k], g[k].cpu().numpy()
            assert np.array_equal(a, b), f"{what} {k}: max|d|={np.abs(a.astype(np.float64) - b).max()}"


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("name,p,iters,cn,factor", [("ghp882", 0.05, 1, "boxplus-phi", 1.0), ("ghp882", 0.10, 2, "boxplus-phi", 1.0),
                                                    ("ghp882", 0.08, 16, "boxplus-phi", 0.8), ("ghp882", 0.03, 64, "boxplus-phi", 1.0),
                                                    ("ghp1270", 0.08, 64, "boxplus-phi", 1.0), ("ghp882", 0.06, 12, "minsum", 0.625),
                                                    ("ghp882", 0.06, 12, "boxplus", 0.625), ("gb254", 0.03, 30, "boxplus-phi", 1.0),
                                                    ("rsurf5", 0.04, 20, "boxplus-phi", 0.9), ("gb48_oc", 0.02, 8, "boxplus-phi", 1.0)])
def test_both_forms_bit_exact_in_every_kernel_variant(name, p, iters, cn, factor, shared):
    B = 48
    with _shared(name, shared) as (og, gg):
        ex, ez = og.pauli_noise(SEED, p, 555, B)
        sx, sz = og.syndrome(ex, ez)
        tx, tz = to_gpu(sx), to_gpu(sz)
        L0 = llr_const(0.05)
        o = og.bp4_decode(sx, sz, iters, cn, factor, llr_const=L0, return_msgs=True)
        rng = np.random.RandomState(4)
        llr = rng.uniform(0.4, 3.5, size=(B, 3, gg.n)).astype(np.float32)
        init = (rng.uniform(-12, 12, size=(B, gg.E_x)).astype(np.float32), rng.uniform(-12, 12, size=(B, gg.E_z)).astype(np.float32))
        o2 = og.bp4_decode(sx, sz, iters, cn, factor, llr_ch=llr, msg_init=init, return_msgs=True)
        try:
            for shortcut, fpe, generic in ((True, True, False), (False, False, False), (True, False, True), (False, False, True)):
                gg.set_saturation_shortcut(shortcut)
                gg.set_fixed_point_exit(fpe)
                gg.force_generic(generic)
                tag = f"{name} shared={shared} shortcut={shortcut} exit={fpe} generic={generic}"
                _eq(o, gg.bp4_decode(tx, tz, iters, cn, factor, llr_const=L0, return_msgs=True), tag)
                _eq(o2, gg.bp4_decode(tx, tz, iters, cn, factor, llr_ch=to_gpu(llr), msg_init=tuple(to_gpu(a) for a in init), return_msgs=True),
                    tag + " llr_ch + restart")
            gg.force_generic(False)
            tr = gg.bp4_decode_trace(tx, tz, iters, cn, factor, llr_ch=to_gpu(llr), msg_init=tuple(to_gpu(a) for a in init), want_tape=True)
            assert np.array_equal(o2["x_logit"], tr["x_logit"][iters].cpu().numpy()) and np.array_equal(o2["msg_x"], tr["tape_x"][iters].cpu().numpy())
            assert np.array_equal(o2["llr"], tr["llr"].cpu().numpy())
        finally:
            gg.set_saturation_shortcut(True)
            gg.set_fixed_point_exit(True)
            gg.force_generic(False)


def test_the_two_forms_differ_by_rounding_in_one_update_and_meet_on_the_fixed_point():
    name, B = "ghp882", 512
    og, gg = oracle_library_forms(name), gpu_graph(name)
    ex, ez = og.pauli_noise(SEED, 0.05, 9000, B)
    sx, sz = og.syndrome(ex, ez)
    tx, tz = to_gpu(sx), to_gpu(sz)
    L0 = llr_const(0.05)
    rng = np.random.RandomState(8)
    init = tuple(to_gpu(rng.uniform(-16, 16, size=(B, e)).astype(np.float32)) for e in (gg.E_x, gg.E_z))
    res = {}
    for shared in (False, True):
        with _shared(name, shared):
            gg.set_saturation_shortcut(False)
            try:
                # minsum keeps |c->v| = min|v->c|: the check phase passes the qubit update's perturbation on unamplified
                one = gg.bp4_decode(tx, tz, 1, "minsum", 1.0, llr_const=L0, msg_init=init, return_msgs=True)
                full = gg.bp4_decode(tx, tz, 64, "boxplus-phi", 1.0, llr_const=L0)
            finally:
                gg.set_saturation_shortcut(True)
        res[shared] = (one, full)
    d = (res[False][0]["msg_x"] - res[True][0]["msg_x"]).abs().max().item()
    scale = res[False][0]["llr"].abs().max().item()
    assert 0 < d <= 4 * np.spacing(np.float32(scale)) , (d, scale)  # a few ulp of the totals the subtractions are rounded at
    a, b = res[False][1], res[True][1]
    hx, hz = torch.from_numpy(np.asarray(code(name).hx)).cuda().float(), torch.from_numpy(np.asarray(code(name).hz)).cuda().float()

    def conv(o):
        return ~((((o["x_hat"].float() @ hz.t()) % 2) != tz.float()).any(1) | (((o["z_hat"].float() @ hx.t()) % 2) != tx.float()).any(1))

    both = conv(a) & conv(b)
    assert both.float().mean() > 0.95 and (conv(a) ^ conv(b)).float().mean() < 0.01
    same = (a["x_hat"] == b["x_hat"]).all(1) & (a["z_hat"] == b["z_hat"]).all(1)
    assert same[both].float().mean() >= 0.99
    dl = (a["llr"] - b["llr"]).abs().flatten(1).max(1).values
    assert (dl[both & same] <= 1e-4).float().mean() >= 0.99  # the saturated fixed point is the same floats


@pytest.mark.parametrize("name,wfile,iters,p", [("ghp882", WEIGHTS_882, [64, 16, 16, 16], 0.10), ("ghp1270", WEIGHTS_1270, [64, 64], 0.10)])
@pytest.mark.parametrize("compact", [False, True])
def test_sandwich_bit_exact_with_the_shared_form(name, wfile, iters, p, compact):
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    B = 160
    w = read_weight_list(wfile)
    with _shared(name, True) as (og, gg):
        ex, ez = og.pauli_noise(SEED, p, 4321, B)
        sx, sz = og.syndrome(ex, ez)
        gw = GnnWeights(w, gg.device)
        nl = len(iters)
        o = og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), llr_const(0.05), return_llr=True)
        g = gg.sandwich_decode(to_gpu(sx), to_gpu(sz), iters, [gw] * (nl - 1), llr_const(0.05), compact=compact, return_llr=True,
                               return_rounds=True)
    assert np.array_equal(o["x_hat"], g["x_hat"].cpu().numpy()) and np.array_equal(o["z_hat"], g["z_hat"].cpu().numpy())
    assert np.array_equal(o["rounds"], g["rounds"].cpu().numpy()) and o["rounds"].sum() > 0
    if not compact:
        assert np.array_equal(o["llr"], g["llr"].cpu().numpy())

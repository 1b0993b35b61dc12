"""Pins on GPU OUTPUT that do not pass through feedback_gnn_amd/csrc/fgnn_math.h on the checking side.

The parity tests (test_gpu_parity.py) prove HIP == C oracle bit for bit, but the C oracle is compiled from the product's own
fgnn_math.h / fgnn_rng.h: a wrong constant or formula in those headers would be invisible to them.  Here the HIP results are
checked against (i) the exact float32 known answers of the reference's committed notebook output and (ii) oracle/numpy_ref.py,
an independently written restatement on NumPy's own exp / log / log1p / tanh and the reference's batch-minor tensor layout —
with the bars tests/test_oracle_kat.py holds the C oracle to.  Syndromes are recomputed with a plain NumPy matmul."""
import numpy as np
import pytest
import torch

from feedback_gnn_amd.weights_io import read_weight_list
from helpers import WEIGHTS_1270, WEIGHTS_882, code, gpu_graph, llr_const
from oracle import numpy_ref as R

pytestmark = pytest.mark.gpu
SEED = 0x5EED
LLR_TOL = 1e-4  # north-star tolerance on LLRs


def _gpu_case(name, p, B, first=0):
    """Noise from the HIP Philox kernel; syndromes BOTH from the HIP kernel and from an int64 NumPy matmul (they must agree)."""
    g, c = gpu_graph(name), code(name)
    ex, ez = g.pauli_noise(SEED, p, first, B)
    sx, sz = g.syndrome(ex, ez)
    nx, nz = ex.cpu().numpy().astype(np.int64), ez.cpu().numpy().astype(np.int64)
    ref_sx, ref_sz = (nz @ np.asarray(c.hx).T % 2).astype(np.uint8), (nx @ np.asarray(c.hz).T % 2).astype(np.uint8)
    assert np.array_equal(sx.cpu().numpy(), ref_sx) and np.array_equal(sz.cpu().numpy(), ref_sz)
    return g, c, sx, sz, ref_sx, ref_sz


def _np(d):
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


@pytest.mark.parametrize("shortcut", [True, False])
@pytest.mark.parametrize("name", ["ghp882", "ghp1270"])
def test_saturation_known_answer_on_gpu_output(name, shortcut):
    """/root/reference examples/n1270.ipynb cell 12: after 64 BP4 iterations with p0 = 0.05 the marginals saturate at
    max [53.9496498 103.856247 53.9496498], min [-45.8635445 -95.7701416 -45.8635445] (float32 prints) = log(57) +- deg * 16.635532:
    pins the phi clip constants (decoding_q.py:372), the column weight and the LLR convention on what the kernel writes."""
    g, c, sx, sz, _, _ = _gpu_case(name, 0.05, 64)
    g.set_saturation_shortcut(shortcut)
    try:
        o = _np(g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05)))
    finally:
        g.set_saturation_shortcut(True)
    mx, mn = o["llr"].max(axis=(0, 2)), o["llr"].min(axis=(0, 2))
    assert [f"{v:.9g}" for v in mx] == ["53.9496498", "103.856247", "53.9496498"]
    assert [f"{v:.9g}" for v in mn] == ["-45.8635445", "-95.7701416", "-45.8635445"]
    assert f"{llr_const(0.05):.8g}" == "4.0430512"


def test_single_iteration_gpu_vs_numpy_restatement():
    """After ONE iteration nothing has been amplified yet: marginals within 1e-5 of the NumPy restatement, decisions equal."""
    g, c, sx, sz, nsx, nsz = _gpu_case("ghp882", 0.08, 32)
    o = _np(g.bp4_decode(sx, sz, 1, "boxplus-phi", 1.0, llr_const=llr_const(0.05)))
    r = R.bp4_decode(R.Graph(c), nsx, nsz, 1, llr_const=llr_const(0.05))
    assert np.abs(o["llr"] - r["llr"]).max() <= 1e-5
    assert np.array_equal(o["x_hat"], r["x_hat"]) and np.array_equal(o["z_hat"], r["z_hat"])


@pytest.mark.parametrize("name,p,iters", [("ghp882", 0.05, 64), ("ghp882", 0.10, 32), ("ghp1270", 0.05, 64), ("gb48", 0.04, 16),
                                          ("rsurf3", 0.05, 20)])
def test_gpu_vs_numpy_restatement(name, p, iters):
    """HIP kernel vs oracle/numpy_ref.py with the bar of test_oracle_kat.py::test_c_oracle_vs_numpy_restatement: on samples
    both decode to the syndrome, identical decisions and LLRs within 1e-4 (the float32 phi of decoding_q.py:372-373 is
    rounding noise above ~12, so two faithful implementations drift apart in the transient and meet on the fixed point)."""
    B = 96
    g, c, sx, sz, nsx, nsz = _gpu_case(name, p, B, first=77)
    L0 = llr_const(0.05)
    o = _np(g.bp4_decode(sx, sz, iters, "boxplus-phi", 1.0, llr_const=L0))
    r = R.bp4_decode(R.Graph(c), nsx, nsz, iters, llr_const=L0)
    hx, hz = np.asarray(c.hx), np.asarray(c.hz)
    CODE = c

    def converged(d):
        return ~(((d["x_hat"].astype(int) @ hz.T % 2) != nsz).any(1) | ((d["z_hat"].astype(int) @ hx.T % 2) != nsx).any(1))

    both = converged(o) & converged(r)
    flipped = converged(o) ^ converged(r)
    same = (o["x_hat"] == r["x_hat"]).all(1) & (o["z_hat"] == r["z_hat"]).all(1)
    assert both.sum() >= B // 4, "too few converged samples for the comparison to mean anything"
    assert flipped.mean() <= 0.15 and abs(converged(o).mean() - converged(r).mean()) <= 0.06
    # [[n,k]] stabilizer codes are degenerate: two estimates that differ by a stabilizer (a sum of check rows) are the SAME
    # correction.  Chaotic transients occasionally end on different representatives (weight-6 check rows, observed), so the
    # bar is: every commonly converged sample gets the same correction class — d_x in rowspace(hx) <=> hx_perp . d_x = 0,
    # likewise z — and all but a few get the identical representative.
    HXP, HZP = np.asarray(CODE.hx_perp).astype(int), np.asarray(CODE.hz_perp).astype(int)
    equiv = ~(((o["x_hat"] ^ r["x_hat"]).astype(int) @ HXP.T % 2).any(1) | ((o["z_hat"] ^ r["z_hat"]).astype(int) @ HZP.T % 2).any(1))
    assert equiv[both].mean() >= 0.98, "different correction classes on samples both implementations converge on"
    assert same[both].mean() >= 0.95, "too many different representatives on samples both implementations converge on"
    d = np.abs(o["llr"] - r["llr"]).reshape(B, -1).max(1)
    dl = np.abs(o["x_logit"] - r["x_logit"]).reshape(B, -1).max(1)
    if c.N >= 800:
        assert (d[both & same] <= LLR_TOL).mean() >= (0.95 if iters >= 64 else 0.8) and np.median(d[both & same]) <= LLR_TOL
        assert np.median(dl[both & same]) <= LLR_TOL
    else:
        rel = d[both & same] / np.abs(o["llr"]).reshape(B, -1).max(1)[both & same]
        assert np.median(rel) <= 1e-4


@pytest.mark.parametrize("name,wfile", [("ghp882", WEIGHTS_882), ("ghp1270", WEIGHTS_1270)])
def test_gpu_feedback_gnn_vs_numpy_restatement(name, wfile):
    """HIP feedback GNN (MFMA kernel) on the HIP decoder's own output vs numpy_ref.feedback_gnn (np.matmul, np.tanh): <= 1e-4,
    and the output band of examples/n1270.ipynb cell 12 (roughly 0.2 .. 2.7, all positive)."""
    from feedback_gnn_amd.graph import GnnWeights
    w = read_weight_list(wfile)
    assert sum(a.size for a in w) == 3923  # examples/Feedback_GNN.ipynb cell 6: "Total params: 3,923"
    g, c, sx, sz, nsx, nsz = _gpu_case(name, 0.10, 48)
    o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    out = g.feedback_gnn(GnnWeights(w, g.device), o["llr"], o["z_logit"], o["x_logit"], sx, sz).cpu().numpy()
    on = _np(o)
    ref = R.feedback_gnn(R.Graph(c), w, on["llr"], on["z_logit"], on["x_logit"], nsx, nsz)
    assert np.abs(out - ref).max() <= 1e-4
    assert out.min() > 0.0 and out.max() < 4.0 and 1.0 < out.mean() < 2.6
    # the runtime-degree VALU kernel of the same layer
    g.force_generic(True)
    try:
        out2 = g.feedback_gnn(GnnWeights(w, g.device), o["llr"], o["z_logit"], o["x_logit"], sx, sz).cpu().numpy()
    finally:
        g.force_generic(False)
    assert np.abs(out2 - ref).max() <= 1e-4


@pytest.mark.parametrize("name", ["gb48", "rsurf5"])
def test_gpu_gnn_bp4_vs_numpy_restatement(name):
    """GNN_BP4 (gnn.py:383-423, repaired arity): HIP kernel vs the NumPy matmul restatement, random weights."""
    from feedback_gnn_amd.graph import GNNBP4_SHAPES, GnnBp4Weights
    rng = np.random.RandomState(11)
    w = []
    for shp in GNNBP4_SHAPES:
        lim = 0.6 if len(shp) == 1 else np.sqrt(6.0 / (shp[0] + shp[1]))
        w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    g, c, sx, sz, nsx, nsz = _gpu_case(name, 0.05, 12)
    o = _np(g.gnn_bp4_decode(GnnBp4Weights(w, g.device), sx, sz, 4))
    r = R.gnn_bp4(c, w, nsx, nsz, 4)
    assert np.abs(o["llr"] - r["llr"]).max() <= 2e-4
    assert np.abs(o["x_logit_all"] - r["x_logit_all"]).max() <= 1e-3 and np.abs(o["z_logit_all"] - r["z_logit_all"]).max() <= 1e-3
    assert (o["x_hat"] == r["x_hat"]).mean() > 0.98  # random weights leave marginals near ties; exact ties may break either way

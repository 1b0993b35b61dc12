"""Pin the CPU oracle to the reference: exact float32 known answers from the committed notebooks, an
independently written NumPy restatement, and the published error-rate tables."""
import numpy as np
import pytest

from feedback_gnn_amd.weights_io import read_weight_list
from helpers import WEIGHTS_1270, WEIGHTS_882, code, llr_const, oracle_library_forms
from oracle import numpy_ref as R

SEED = 0x5EED
LLR_TOL = 1e-4  # north-star tolerance on LLRs


def _setup(name, p, B, first=0):
    g = oracle_library_forms(name)
    ex, ez = g.pauli_noise(SEED, p, first, B)
    sx, sz = g.syndrome(ex, ez)
    return g, ex, ez, sx, sz


@pytest.mark.parametrize("name", ["ghp882", "ghp1270"])
def test_saturation_known_answer(name):
    """examples/n1270.ipynb cell 12: after 64 BP4 iterations with p0=0.05 the marginals saturate at
    max [53.9496498 103.856247 53.9496498], min [-45.8635445 -95.7701416 -45.8635445] (float32 prints)."""
    g, ex, ez, sx, sz = _setup(name, 0.05, 64)
    o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    mx, mn = o["llr"].max(axis=(0, 2)), o["llr"].min(axis=(0, 2))
    assert [f"{v:.9g}" for v in mx] == ["53.9496498", "103.856247", "53.9496498"]
    assert [f"{v:.9g}" for v in mn] == ["-45.8635445", "-95.7701416", "-45.8635445"]
    assert f"{llr_const(0.05):.8g}" == "4.0430512"


def test_syndromes_and_noise_statistics():
    g, ex, ez, sx, sz = _setup("ghp882", 0.09, 2000)
    c = code("ghp882")
    assert np.array_equal(sx, ez.astype(np.int64) @ c.hx.T % 2) and np.array_equal(sz, ex.astype(np.int64) @ c.hz.T % 2)
    # depolarizing: X, Y, Z each with probability p/3 (pauli.py:98-108 with feedback_gnn.py:298)
    n_tot = ex.size
    for cnt in ((ex & ~ez & 1).sum(), (ex & ez).sum(), (~ex & ez & 1).sum()):
        assert abs(cnt / n_tot - 0.03) < 5 * np.sqrt(0.03 / n_tot)
    # the stream is keyed by the global sample index: any split of the batch gives the same samples
    e2x, e2z = g.pauli_noise(SEED, 0.09, 700, 300)
    assert np.array_equal(e2x, ex[700:1000]) and np.array_equal(e2z, ez[700:1000])


def numpy_pauli_xyz(seed, px, py, pz, first, B, n):
    """Pauli.call (pauli.py:98-108) in NumPy float32 on the build's Philox stream: u = Uint32ToFloat(word q % 4 of block q // 4 of
    sample first + b); noise_x = u < px (:103); noise_z = (u >= px - py) & (u < (px + pz) - py) (:104-106)."""
    from oracle import oracle as O
    px, py, pz = np.float32(px), np.float32(py), np.float32(pz)
    u = np.empty((B, 4 * ((n + 3) // 4)), np.float32)
    for b in range(B):
        s = first + b
        for blk in range((n + 3) // 4):
            r = O.philox([s & 0xffffffff, s >> 32, blk, 0], [seed & 0xffffffff, seed >> 32])
            u[b, 4 * blk:4 * blk + 4] = ((r >> np.uint32(9)) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1)
    u = u[:, :n]
    return (u < px).astype(np.uint8), ((u >= px - py) & (u < (px + pz) - py)).astype(np.uint8)


@pytest.mark.parametrize("triple", [(0.06, 0.03, 0.06), (0.11, 0.02, 0.05), (0.05, 0.0, 0.0), (0.0, 0.0, 0.08), (0.3, 0.3, 0.3)])
def test_general_pauli_triple_vs_numpy_threshold_restatement(triple):
    """Pauli.call takes ANY (px, py, pz) (pauli.py:98-108); the oracle's og_pauli_noise_xyz against NumPy's own float32 comparisons on
    the same Philox words — asymmetric channel, pure bit-flip, pure phase-flip, pure Y — and the depolarizing entry point is the triple
    (2p/3, p/3, 2p/3) formed in float32."""
    g = oracle_library_forms("steane")
    first, B = (1 << 32) - 3, 6  # crosses the 32-bit boundary of the sample counter
    ex, ez = g.pauli_noise_xyz(SEED, *triple, first, B)
    rx, rz = numpy_pauli_xyz(SEED, *triple, first, B, g.n)
    assert np.array_equal(ex, rx) and np.array_equal(ez, rz)
    # event rates on a larger draw: X = px - py, Y = py, Z = pz - py
    g2 = oracle_library_forms("gb254")
    ex, ez = g2.pauli_noise_xyz(SEED, *triple, 0, 2000)
    px, py, pz = triple
    for cnt, want in (((ex & ~ez & 1).sum(), px - py), ((ex & ez).sum(), py), ((~ex & ez & 1).sum(), pz - py)):
        assert abs(cnt / ex.size - want) <= 5 * np.sqrt(max(want, 1e-6) / ex.size), (triple, cnt / ex.size, want)
    p = np.float32(0.09)
    a = g2.pauli_noise(SEED, p, 17, 300)
    b = g2.pauli_noise_xyz(SEED, (np.float32(2) * p) / np.float32(3), p / np.float32(3), (np.float32(2) * p) / np.float32(3), 17, 300)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("name,p,iters", [("ghp882", 0.05, 64), ("ghp882", 0.10, 32), ("gb48", 0.04, 16), ("rsurf3", 0.05, 20)])
def test_c_oracle_vs_numpy_restatement(name, p, iters):
    """The C oracle (polynomial exp/log, scalar loops) against oracle/numpy_ref.py (NumPy exp/log,
    batch-minor tensors).

    The reference's float32 phi (decoding_q.py:372-373) is rounding noise for arguments above ~12
    (softplus(x) - log(exp(x)-1) cancels to 0 or 1 ulp(x)), and phi(phi(.)) turns that noise into O(1)
    differences of strong messages, so two faithful implementations (TensorFlow CPU vs GPU included)
    drift apart during the transient and meet again on the saturated fixed point.  Hence the bar: on
    samples that BOTH implementations decode to the syndrome, identical decisions and LLRs within 1e-4;
    the few trapping-set samples whose convergence flips are counted, not compared."""
    B = 96
    g, ex, ez, sx, sz = _setup(name, p, B, first=77)
    L0 = llr_const(0.05)
    o = g.bp4_decode(sx, sz, iters, "boxplus-phi", 1.0, llr_const=L0)
    r = R.bp4_decode(R.Graph(code(name)), sx, sz, iters, llr_const=L0)
    c = CODE = code(name)

    def converged(d):
        return ~(((d["x_hat"].astype(int) @ c.hz.T % 2) != sz).any(1) | ((d["z_hat"].astype(int) @ c.hx.T % 2) != sx).any(1))

    both = converged(o) & converged(r)
    flipped = converged(o) ^ converged(r)
    same = (o["x_hat"] == r["x_hat"]).all(1) & (o["z_hat"] == r["z_hat"]).all(1)
    assert both.sum() >= B // 4, "too few converged samples for the comparison to mean anything"
    # near threshold (p=0.10, 32 iterations) many samples are still mid-flight: flips are expected, a
    # systematic difference in the convergence RATE would not be
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
    if code(name).N >= 800:
        # large graphs: converged samples sit on the exact saturated fixed point after enough iterations
        # (samples that converged only just before the last iteration are not saturated yet)
        assert (d[both & same] <= LLR_TOL).mean() >= (0.95 if iters >= 64 else 0.8) and np.median(d[both & same]) <= LLR_TOL
        assert np.median(dl[both & same]) <= LLR_TOL
    else:
        # small codes never saturate: the marginals stay analog, the comparison is relative
        rel = d[both & same] / np.abs(o["llr"]).reshape(B, -1).max(1)[both & same]
        assert np.median(rel) <= 1e-4


def test_single_iteration_tight_agreement():
    """After ONE iteration nothing has been amplified yet: every message-derived quantity agrees to 1e-5."""
    g, ex, ez, sx, sz = _setup("ghp882", 0.08, 32)
    o = g.bp4_decode(sx, sz, 1, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    r = R.bp4_decode(R.Graph(code("ghp882")), sx, sz, 1, llr_const=llr_const(0.05))
    assert np.abs(o["llr"] - r["llr"]).max() <= 1e-5
    assert np.array_equal(o["x_hat"], r["x_hat"]) and np.array_equal(o["z_hat"], r["z_hat"])


@pytest.mark.parametrize("name,wfile", [("ghp882", WEIGHTS_882), ("ghp1270", WEIGHTS_1270)])
def test_gnn_vs_numpy_restatement_and_output_band(name, wfile):
    w = read_weight_list(wfile)
    assert sum(a.size for a in w) == 3923  # examples/Feedback_GNN.ipynb cell 6: "Total params: 3,923"
    g, ex, ez, sx, sz = _setup(name, 0.10, 48)
    o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    out = g.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    ref = R.feedback_gnn(R.Graph(code(name)), w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    assert np.abs(out - ref).max() <= 1e-4
    # sanity band of examples/n1270.ipynb cell 12 (GNN output on BP failures: roughly 0.2 .. 2.7, all positive)
    assert out.min() > 0.0 and out.max() < 4.0 and 1.0 < out.mean() < 2.6


def test_feedback_rescues_bp_failures():
    """SURVEY §8c evidence: one GNN pass + 16 BP iterations corrects (nearly) all BP-64 failures at p=0.10."""
    g, ex, ez, sx, sz = _setup("ghp882", 0.10, 600, first=5000)
    w = read_weight_list(WEIGHTS_882)
    o1 = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    f1 = g.residual(ex, ez, o1["x_hat"], o1["z_hat"])[2]
    o2 = g.sandwich_decode(sx, sz, [64, 16], [w], llr_const(0.05))
    f2 = g.residual(ex, ez, o2["x_hat"], o2["z_hat"])[2]
    assert (f1 & 1).sum() >= 5 and (f2 & 1).sum() <= (f1 & 1).sum() // 2
    assert (o2["rounds"] == (f1 & 1)).all()  # exactly the flagged samples enter the feedback round
    # a sample that was not flagged after stage one keeps its stage-one estimate (feedback_gnn.py:339-340)
    keep = (f1 & 1) == 0
    assert np.array_equal(o2["x_hat"][keep], o1["x_hat"][keep]) and np.array_equal(o2["z_hat"][keep], o1["z_hat"][keep])


def test_published_error_rate_band():
    """examples/n882.ipynb cell 2 (3 feedback rounds, factor 1.0): p=0.14 -> 2375/5000 flagged = BLER.
    800 oracle samples: binomial 3-sigma band around 0.475."""
    B = 800
    g, ex, ez, sx, sz = _setup("ghp882", 0.14, B, first=10_000)
    w = read_weight_list(WEIGHTS_882)
    o = g.sandwich_decode(sx, sz, [64, 16, 16, 16], [w, w, w], llr_const(0.05))
    flags = g.residual(ex, ez, o["x_hat"], o["z_hat"])[2]
    rate = ((flags >> 1) & 1).mean()
    sigma = np.sqrt(0.475 * 0.525 / B)
    assert abs(rate - 0.475) < 3.5 * sigma, rate
    assert ((flags & 1) <= ((flags >> 1) & 1)).all()  # flagged implies block error


def test_rotated_surface_flagged_vs_bler():
    """examples/QLDPC.ipynb cell 9: rotated surface d=3, BP4-60, factor 0.8, p0=0.05, p=0.10:
    flagged 0.2747, BLER 0.3078 (10 000 samples)."""
    B = 10000
    g, ex, ez, sx, sz = _setup("rsurf3", 0.10, B)
    o = g.bp4_decode(sx, sz, 60, "boxplus-phi", 0.8, llr_const=llr_const(0.05))
    flags = g.residual(ex, ez, o["x_hat"], o["z_hat"])[2]
    fl, bl = (flags & 1).mean(), ((flags >> 1) & 1).mean()
    assert abs(fl - 0.2747) < 4 * np.sqrt(0.2747 * 0.7253 / B) * np.sqrt(2), fl
    assert abs(bl - 0.3078) < 4 * np.sqrt(0.3078 * 0.6922 / B) * np.sqrt(2), bl


def _gnnbp4_weights(seed=11):
    from feedback_gnn_amd.graph import GNNBP4_SHAPES
    rng = np.random.RandomState(seed)
    w = []
    for shp in GNNBP4_SHAPES:
        lim = 0.6 if len(shp) == 1 else np.sqrt(6.0 / (shp[0] + shp[1]))
        w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    return w


@pytest.mark.parametrize("name", ["gb48", "rsurf5"])
def test_gnn_bp4_oracle_vs_numpy_restatement(name):
    """GNN_BP4 (gnn.py:383-423, repaired): C oracle (fmaf chains, polynomial tanh) vs NumPy matmul restatement."""
    g = oracle_library_forms(name)
    ex, ez = g.pauli_noise(SEED, 0.05, 0, 12)
    sx, sz = g.syndrome(ex, ez)
    w = _gnnbp4_weights()
    o = g.gnn_bp4(w, sx, sz, 4)
    r = R.gnn_bp4(code(name), w, sx, sz, 4)
    assert np.abs(o["llr"] - r["llr"]).max() <= 2e-4
    assert np.abs(o["x_logit_all"] - r["x_logit_all"]).max() <= 1e-3 and np.abs(o["z_logit_all"] - r["z_logit_all"]).max() <= 1e-3
    agree = (o["x_hat"] == r["x_hat"]).mean()
    assert agree > 0.98  # random weights leave many marginals near ties; exact ties may break either way
    assert o["x_logit_all"].shape == (4, 12, g.m_z + g.rows_lz)


def test_torch_float64_restatement_matches_oracle_forward():
    """oracle/torch_ref.py (float64, autograd-ready) vs the C oracle on a well-conditioned case: per-iteration soft
    syndromes and the GNN output agree to float32 accuracy."""
    import torch
    from oracle import torch_ref as T
    name = "gb48"
    c = code(name)
    g = oracle_library_forms(name)
    ex, ez = g.pauli_noise(SEED, 0.05, 0, 6)
    sx, sz = g.syndrome(ex, ez)
    rng = np.random.RandomState(1)
    llr = rng.uniform(0.5, 2.5, size=(6, 3, c.N)).astype(np.float32)
    tg = T.Graph(c)
    xs, zs, _ = T.bp4_logit_trace(tg, torch.from_numpy(llr).to(T.DT), torch.from_numpy(sx), torch.from_numpy(sz), 4, factor=0.9)
    for it in range(5):
        o = g.bp4_decode(sx, sz, it, "boxplus-phi", 0.9, llr_ch=llr)
        assert np.abs(o["x_logit"] - xs[it].numpy()).max() < 2e-4, it
        assert np.abs(o["z_logit"] - zs[it].numpy()).max() < 2e-4, it
    w = read_weight_list(WEIGHTS_882)  # architecture only; gb48 has the same feature widths
    o = g.bp4_decode(sx, sz, 3, "boxplus-phi", 1.0, llr_ch=llr)
    ref = g.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    tw = [torch.from_numpy(a).to(T.DT) for a in w]
    out = T.feedback_gnn(tg, tw, torch.from_numpy(o["llr"]).to(T.DT), torch.from_numpy(o["z_logit"]).to(T.DT),
                         torch.from_numpy(o["x_logit"]).to(T.DT), torch.from_numpy(sx), torch.from_numpy(sz))
    assert np.abs(ref - out.numpy()).max() < 1e-4


GEN_CONFIGS = [(20, 40, 2, 1, 1, 1), (8, 16, 1, 2, 2, 0), (12, 24, 3, 0, 3, 1), (5, 7, 2, 3, 0, 1), (32, 96, 4, 1, 2, 0)]


def _gen_weights(cfg, seed=3):
    from feedback_gnn_amd.graph import gnn_weight_shapes
    rng = np.random.RandomState(seed)
    return [rng.uniform(-0.5, 0.5, size=s).astype(np.float32) for s in gnn_weight_shapes(cfg[0], cfg[1], cfg[2], bool(cfg[5]))]


def test_general_feedback_gnn_oracle_equals_specialised_on_the_shipped_setting():
    """og_feedback_gnn_general (always the literal association: it serves every reduce_op) with (20, 40, 2, mean, tanh, bias) walks
    the same float ops as og_feedback_gnn in the literal order (the default); the opt-in factored order is the same function with other
    roundings."""
    g = oracle_library_forms("gb48")
    ex, ez = g.pauli_noise(SEED, 0.06, 0, 9)
    sx, sz = g.syndrome(ex, ez)
    o = g.bp4_decode(sx, sz, 5, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    w = read_weight_list(WEIGHTS_882)
    b = g.feedback_gnn_general((20, 40, 2, 1, 1, 1), w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    assert not g.gnn_factored
    a = g.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    g.set_gnn_order(1)
    try:
        f = g.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    finally:
        g.set_gnn_order(0)  # the library default
    assert np.array_equal(a, b)
    assert 0 < np.abs(f - a).max() <= 2e-6


@pytest.mark.parametrize("cfg", GEN_CONFIGS)
def test_general_feedback_gnn_oracle_vs_float64_restatement(cfg):
    """Every constructor setting of Feedback_GNN (layers, widths, reduce_op, activation, bias): C oracle vs the float64 torch
    restatement with torch's own matmul / activations / reductions."""
    import torch
    from oracle import torch_ref as T
    name = "rsurf5"  # irregular degrees: qubits with 1 or 2 checks per side, so sum / mean / max / min all differ
    c = code(name)
    g = oracle_library_forms(name)
    ex, ez = g.pauli_noise(SEED, 0.08, 0, 5)
    sx, sz = g.syndrome(ex, ez)
    o = g.bp4_decode(sx, sz, 4, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    w = _gen_weights(cfg)
    got = g.feedback_gnn_general(cfg, w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    tw = [torch.from_numpy(a).to(T.DT) for a in w]
    ref = T.feedback_gnn_general(T.Graph(c), cfg, tw, torch.from_numpy(o["llr"]).to(T.DT), torch.from_numpy(o["z_logit"]).to(T.DT),
                                 torch.from_numpy(o["x_logit"]).to(T.DT), torch.from_numpy(sx), torch.from_numpy(sz)).numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


GNNBP4_GEN_CFGS = [(8, 16, 1, 2, 2, 0, 0, 0, 0), (12, 24, 3, 0, 3, 1, 1, 3, 2), (5, 7, 2, 3, 0, 1, 1, 2, 0), (32, 96, 4, 1, 1, 0, 1, 0, 4),
                   (20, 40, 2, 1, 1, 1, 1, 4, 4)]


def _rand_weights(shapes, seed):
    rng = np.random.RandomState(seed)
    return [rng.uniform(-(0.6 if len(s) == 1 else np.sqrt(6.0 / (s[0] + s[1]))), 0.6 if len(s) == 1 else np.sqrt(6.0 / (s[0] + s[1])),
                        size=s).astype(np.float32) for s in shapes]


def test_general_gnn_bp4_oracle_equals_specialised_on_the_benchmark_setting():
    """og_gnn_bp4_general with (20, 40, 2, mean, tanh, bias, no attributes) walks the float ops of og_gnn_bp4 in the literal order."""
    from oracle import numpy_ref as R
    g, c = oracle_library_forms("gb48"), code("gb48")
    ex, ez = g.pauli_noise(SEED, 0.05, 0, 9)
    sx, sz = g.syndrome(ex, ez)
    cfg = (20, 40, 2, 1, 1, 1, 0, 0, 0)
    w = _rand_weights(R.gnn_bp4_general_shapes(c, cfg), 1)
    assert len(w) == 30
    b = g.gnn_bp4_general(cfg, w, sx, sz, 4)
    g.set_gnn_order(0)
    try:
        a = g.gnn_bp4(w, sx, sz, 4)
    finally:
        g.set_gnn_order(0)  # the library default
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("cfg", GNNBP4_GEN_CFGS)
@pytest.mark.parametrize("name", ["rsurf5", "gb48"])
def test_general_gnn_bp4_oracle_vs_numpy_restatement(name, cfg):
    """Every constructor setting of GNN_BP4 (widths, depth, reduce op, activation, bias, node / edge attributes in the reference's
    np.where edge order): the C oracle against the batch-first NumPy restatement (NumPy's matmul, reduceat and activations)."""
    from oracle import numpy_ref as R
    g, c = oracle_library_forms(name), code(name)
    ex, ez = g.pauli_noise(SEED, 0.06, 0, 5)
    sx, sz = g.syndrome(ex, ez)
    w = _rand_weights(R.gnn_bp4_general_shapes(c, cfg), 3)
    o = g.gnn_bp4_general(cfg, w, sx, sz, 3)
    r = R.gnn_bp4_general(c, cfg, w, sx, sz, 3)
    scale = max(1.0, np.abs(r["llr"]).max())
    assert np.abs(o["llr"] - r["llr"]).max() <= 2e-5 * scale
    assert np.abs(o["x_logit_all"] - r["x_logit_all"]).max() <= 5e-4 and np.abs(o["z_logit_all"] - r["z_logit_all"]).max() <= 5e-4
    assert (o["x_hat"] == r["x_hat"]).mean() > 0.98 and (o["z_hat"] == r["z_hat"]).mean() > 0.98


def test_the_oracle_starts_literal_and_a_checker_must_name_its_forms():
    """The C oracle is created as the LITERAL restatement of the reference (one log-sum-exp per edge, decoding_q.py:254-273; one Dense
    per edge, feedback_gnn.py:175-184) — since round 6 also what libfgnn_hip runs by default ("library-default"); the library's two
    opt-in re-associations are restated next to it ("reassociated") and have to be asked for.  OracleGraph has no default for `forms`:
    every checker says which restatement it compares against.  The two are different float32 operation sequences of the same
    function: equal decisions on an easy batch, marginals that differ in the last bits after ONE iteration from non-trivial messages."""
    from oracle.oracle import OracleGraph
    from helpers import WEIGHTS_882
    from feedback_gnn_amd.weights_io import read_weight_list
    c = code("gb48")
    with pytest.raises(TypeError):
        OracleGraph(c)
    with pytest.raises(ValueError):
        OracleGraph(c, forms="fast")
    lit, dflt, lib = OracleGraph(c, forms="literal"), OracleGraph(c, forms="library-default"), OracleGraph(c, forms="reassociated")
    assert (lit.gnn_factored, lit.vn_shared_lse, dflt.gnn_factored, dflt.vn_shared_lse, lib.gnn_factored, lib.vn_shared_lse) == \
        (False, False, False, False, True, True)
    ex, ez = lit.pauli_noise(0x5EED, 0.03, 0, 64)
    sx, sz = lit.syndrome(ex, ez)
    rng = np.random.RandomState(1)
    init = (rng.uniform(-6, 6, size=(64, lit.E_x)).astype(np.float32), rng.uniform(-6, 6, size=(64, lit.E_z)).astype(np.float32))
    a = lit.bp4_decode(sx, sz, 1, "minsum", 1.0, llr_const=llr_const(0.05), msg_init=init, return_msgs=True)
    b = lib.bp4_decode(sx, sz, 1, "minsum", 1.0, llr_const=llr_const(0.05), msg_init=init, return_msgs=True)
    d = np.abs(a["msg_x"] - b["msg_x"]).max()
    assert 0 < d <= 4 * np.spacing(np.float32(np.abs(a["llr"]).max())), d
    w = read_weight_list(WEIGHTS_882)
    oa = lit.sandwich_decode(sx, sz, [32, 8], [w], llr_const(0.05))
    ob = lib.sandwich_decode(sx, sz, [32, 8], [w], llr_const(0.05))
    assert np.array_equal(oa["x_hat"], ob["x_hat"]) and np.array_equal(oa["z_hat"], ob["z_hat"])

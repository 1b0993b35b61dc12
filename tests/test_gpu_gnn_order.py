"""The feedback GNN's two associations (FGNN_OPT_GNN_FACTORED): literal (feedback_gnn.py:175-184 term by term) and factored
(X/Y/Z part of the first Dense once per qubit and side; ONE last Dense on the edge-summed activations, then / deg, + b2).

Both are restated by the oracle (og_graph_set_gnn_order) and the kernels must equal the oracle bit for bit in either order, on the
streaming VALU kernel (regular graphs, factored order: the default), the MFMA-tile kernel (regular graphs, either order), the
runtime-degree VALU kernel (any graph), the small-launch geometry and inside the sandwich.  The two
orders are the same real-number function: their float32 outputs may differ by rounding only (asserted <= 2e-6 on outputs of
magnitude 0.2..2.7; measured 5e-7), both sit within 1e-4 of the NumPy restatement (numpy's own matmul order), and a sandwich built
on either reaches the same decisions on the samples it decodes.
"""
import numpy as np
import pytest
import torch

from helpers import LIBRARY_GNN_FACTORED, WEIGHTS_882, WEIGHTS_1270, code, gpu_graph, llr_const, oracle_library_forms, to_gpu

pytestmark = pytest.mark.gpu

SEED = 0x5EED


class _order:
    """Both the GPU graph and the oracle graph in one association, restored on exit (the graphs are shared by the whole session)."""

    def __init__(self, name, factored):
        self.og, self.gg, self.f = oracle_library_forms(name), gpu_graph(name), factored

    def __enter__(self):
        self.prev = self.gg.gnn_factored
        self.og.set_gnn_order(self.f)
        self.gg.set_gnn_factored(self.f)
        self.gg.set_gnn_stream("always")  # the library's own choice would be the MFMA tiles at test-sized batches
        return self.og, self.gg

    def __exit__(self, *exc):
        self.og.set_gnn_order(self.prev)
        self.gg.set_gnn_factored(self.prev)
        self.gg.set_gnn_stream(True)


def _bp_inputs(name, p, B, first=0, iters=64):
    og = oracle_library_forms(name)
    ex, ez = og.pauli_noise(SEED, p, first, B)
    sx, sz = og.syndrome(ex, ez)
    o = og.bp4_decode(sx, sz, iters, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    return (sx, sz), o


@pytest.mark.parametrize("name,wfile,p", [("ghp882", WEIGHTS_882, 0.10), ("ghp1270", WEIGHTS_1270, 0.09), ("ghp882", WEIGHTS_882, 0.01)])
def test_both_orders_bit_exact_on_mfma_and_valu_kernels(name, wfile, p):
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    B = 40
    (sx, sz), o = _bp_inputs(name, p, B, first=77)
    w = read_weight_list(wfile)
    rng = np.random.RandomState(9)
    wr = [rng.uniform(-0.7, 0.7, size=a.shape).astype(np.float32) for a in w]  # random weights: no near-zero column hides a permutation error
    outs = {}
    for ww, tag in ((w, "trained"), (wr, "random")):
        for fact in (False, True):
            with _order(name, fact) as (og, gg):
                ref = og.feedback_gnn(ww, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
                gw = GnnWeights(ww, gg.device)
                args = (gw, to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), to_gpu(sx), to_gpu(sz))
                a = gg.feedback_gnn(*args).cpu().numpy()
                gg.force_generic(True)
                try:
                    b = gg.feedback_gnn(*args).cpu().numpy()
                finally:
                    gg.force_generic(False)
                # a compacted round's geometry: few codewords, a codeword's tiles dealt to several wave-quads
                c = gg.feedback_gnn(gw, *[t[:3].contiguous() for t in args[1:]]).cpu().numpy()
                # a, c ran on the streaming VALU kernel (both orders have one on a regular graph; _order forces it at every launch
                # size): the MFMA-tile kernel in the same order must give the same bits
                gg.set_gnn_stream(False)
                try:
                    a2 = gg.feedback_gnn(*args).cpu().numpy()
                    c2 = gg.feedback_gnn(gw, *[t[:3].contiguous() for t in args[1:]]).cpu().numpy()
                finally:
                    gg.set_gnn_stream("always")
                assert np.array_equal(ref, a2), f"{tag} factored={fact} MFMA kernel: max|d|={np.abs(ref - a2).max()}"
                assert np.array_equal(ref[:3], c2), f"{tag} factored={fact} MFMA kernel, small launch"
            assert np.array_equal(ref, a), f"{tag} factored={fact} default kernel: max|d|={np.abs(ref - a).max()}"
            assert np.array_equal(ref, b), f"{tag} factored={fact} VALU kernel: max|d|={np.abs(ref - b).max()}"
            assert np.array_equal(ref[:3], c), f"{tag} factored={fact} small launch"
            outs[tag, fact] = ref
    d = np.abs(outs["trained", False] - outs["trained", True]).max()
    assert 0 < d <= 2e-6, d  # different roundings (not the same code path twice), same function
    assert np.abs(outs["random", False] - outs["random", True]).max() <= 2e-5  # outputs up to ~20 with +-0.7 weights


@pytest.mark.parametrize("name,B", [("gb48", 70), ("gb254", 9), ("gb126", 5), ("ibm72", 33)])
def test_streaming_kernel_on_the_other_regular_degrees(name, B):
    """The streaming VALU kernel is instantiated for 3, 4 and 5 checks per qubit and side (GB_n48: 4, GB_n126 / GB_n254: 5, the
    bivariate-bicycle [[72,12]]: 3): exact against the oracle with random weights, equal to the runtime-degree kernel, and the check
    logits of both signs and of saturated size."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    og, gg = oracle_library_forms(name), gpu_graph(name)
    assert gg.info()["regular"] and gg.gnn_factored == LIBRARY_GNN_FACTORED and gg.gnn_stream
    ex, ez = og.pauli_noise(SEED, 0.06, 3, B)
    sx, sz = og.syndrome(ex, ez)
    o = og.bp4_decode(sx, sz, 12, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    rng = np.random.RandomState(17)
    wr = [rng.uniform(-0.7, 0.7, size=a.shape).astype(np.float32) for a in read_weight_list(WEIGHTS_882)]
    llr = o["llr"].copy()
    llr[0] = 0.0                       # all-zero marginals
    llr[-1] = np.float32(53.9496498)   # saturated
    for fact in (True, False):  # both associations have a streaming kernel per degree
        with _order(name, fact) as (og_, gg_):
            ref = og_.feedback_gnn(wr, llr, o["z_logit"], o["x_logit"], sx, sz)
            gw = GnnWeights(wr, gg_.device)
            args = (gw, to_gpu(llr), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), to_gpu(sx), to_gpu(sz))
            a = gg_.feedback_gnn(*args).cpu().numpy()
            gg_.force_generic(True)
            try:
                b = gg_.feedback_gnn(*args).cpu().numpy()
            finally:
                gg_.force_generic(False)
            one = gg_.feedback_gnn(gw, *[t[B - 1:].contiguous() for t in args[1:]]).cpu().numpy()
        assert np.array_equal(ref, a), (fact, np.abs(ref - a).max())
        assert np.array_equal(ref, b) and np.array_equal(ref[B - 1:], one), fact
        assert np.isfinite(ref).all() and np.abs(ref).max() > 1.0


def test_both_orders_within_tolerance_of_the_numpy_restatement():
    """oracle/numpy_ref.py: batch-minor tensors, numpy matmul + bias + mean in the literal TensorFlow op structure."""
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    from oracle import numpy_ref as NR
    name, B = "ghp882", 32
    (sx, sz), o = _bp_inputs(name, 0.10, B, first=5)
    w = read_weight_list(WEIGHTS_882)
    ref = NR.feedback_gnn(NR.Graph(code(name)), w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    assert ref.shape == o["llr"].shape
    for fact in (False, True):
        with _order(name, fact) as (_, gg):
            a = gg.feedback_gnn(GnnWeights(w, gg.device), to_gpu(o["llr"]), to_gpu(o["z_logit"]), to_gpu(o["x_logit"]), to_gpu(sx),
                                to_gpu(sz)).cpu().numpy()
        assert np.abs(a - ref).max() <= 1e-5, (fact, np.abs(a - ref).max())
        assert 0.15 < a.min() and a.max() < 3.0  # the output band of n1270.ipynb cell 12


@pytest.mark.parametrize("name,wfile,iters,p", [("ghp882", WEIGHTS_882, [64, 16, 16, 16], 0.10), ("ghp1270", WEIGHTS_1270, [64, 64], 0.10)])
@pytest.mark.parametrize("compact", [False, True])
def test_sandwich_bit_exact_in_both_orders_and_same_corrections(name, wfile, iters, p, compact):
    _sandwich_both_orders(name, wfile, iters, p, compact)


def test_kernel_choice_by_launch_size_gives_the_same_bits():
    """FGNN_OPT_GNN_STREAM = 1 (the default): MFMA tiles below 4 096 codewords per launch (either association), the
    streaming kernel from there on — at both sides of the switch the output equals that of either kernel forced."""
    import torch
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    gg = gpu_graph("ghp882")
    ex, ez = gg.pauli_noise(SEED, 0.10, 0, 4096)
    sx, sz = gg.syndrome(ex, ez)
    o = gg.bp4_decode(sx, sz, 8, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    gw = GnnWeights(read_weight_list(WEIGHTS_882), gg.device)
    try:
        for fact, B in ((True, 4096), (False, 4096)):
            gg.set_gnn_factored(fact)
            outs = {}
            for mode in (False, True, "always"):
                gg.set_gnn_stream(mode)
                assert gg.gnn_stream == mode
                for n in (B, B - 1):
                    outs[mode, n] = gg.feedback_gnn(gw, *[t[:n].contiguous() for t in (o["llr"], o["z_logit"], o["x_logit"], sx, sz)])
            for n in (B, B - 1):
                assert torch.equal(outs[False, n], outs[True, n]) and torch.equal(outs[True, n], outs["always", n]), (fact, n)
    finally:
        gg.set_gnn_stream(True)
        gg.set_gnn_factored(LIBRARY_GNN_FACTORED)
    with pytest.raises(Exception, match="0, 1 or 2"):
        from feedback_gnn_amd import _lib
        _lib.check(_lib.lib().fgnn_graph_set_option(gg.handle, 6, 3))


def test_sandwich_with_the_mfma_tile_kernel():
    """FGNN_OPT_GNN_STREAM off: the factored order on the MFMA-tile kernel inside the sandwich (the streaming kernel is the default)."""
    _sandwich_both_orders("ghp882", WEIGHTS_882, [64, 16, 16, 16], 0.10, True, stream=False)


def _sandwich_both_orders(name, wfile, iters, p, compact, stream="always"):
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    B = 192
    og, gg = oracle_library_forms(name), gpu_graph(name)
    ex, ez = og.pauli_noise(SEED, p, 999, B)
    sx, sz = og.syndrome(ex, ez)
    w = read_weight_list(wfile)
    gw = GnnWeights(w, gg.device)
    nl = len(iters)
    res = {}
    for fact in (False, True):
        with _order(name, fact):
            gg.set_gnn_stream(stream)
            o = og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), llr_const(0.05), return_llr=True)
            g = gg.sandwich_decode(to_gpu(sx), to_gpu(sz), iters, [gw] * (nl - 1), llr_const(0.05), compact=compact, return_llr=True,
                                   return_rounds=True)
        assert np.array_equal(o["x_hat"], g["x_hat"].cpu().numpy()) and np.array_equal(o["z_hat"], g["z_hat"].cpu().numpy()), fact
        assert np.array_equal(o["rounds"], g["rounds"].cpu().numpy())
        if not compact:
            assert np.array_equal(o["llr"], g["llr"].cpu().numpy())
        assert o["rounds"].sum() > 0
        res[fact] = o
    # same function => the sandwich decodes (estimate reproduces the syndrome) essentially the same samples with the same estimate
    c = code(name)
    hx, hz = np.asarray(c.hx, dtype=np.int64), np.asarray(c.hz, dtype=np.int64)

    def solved(o):
        return ~(((o["x_hat"].astype(np.int64) @ hz.T) % 2 != sz).any(1) | ((o["z_hat"].astype(np.int64) @ hx.T) % 2 != sx).any(1))

    s0, s1 = solved(res[False]), solved(res[True])
    both = s0 & s1
    assert (s0 == s1).mean() >= 0.97 and both.sum() >= 0.8 * B
    same = (res[False]["x_hat"] == res[True]["x_hat"]).all(1) & (res[False]["z_hat"] == res[True]["z_hat"]).all(1)
    assert same[both].mean() >= 0.97, same[both].mean()


@pytest.mark.parametrize("name,B,iters", [("ghp882", 6, 4), ("ghp1270", 4, 10), ("gb48", 21, 5), ("rsurf5", 9, 3)])
def test_gnn_bp4_both_orders_bit_exact(name, B, iters):
    """GNN_BP4 (gnn.py:383-423): the message MLPs of both update directions in the literal and in the factored association (own half of
    the first Dense once per node and side; ONE last Dense on the signed sum of the hidden activations) — MFMA kernel and streaming
    packed-FMA kernel (regular graphs), runtime-degree VALU kernel (any graph), against the oracle in the same order, exactly; the two
    orders agree to rounding."""
    from feedback_gnn_amd.graph import GNNBP4_SHAPES, GnnBp4Weights
    rng = np.random.RandomState(11)
    w = []
    for shp in GNNBP4_SHAPES:
        lim = 0.6 if len(shp) == 1 else np.sqrt(6.0 / (shp[0] + shp[1]))
        w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    og0 = oracle_library_forms(name)
    ex, ez = og0.pauli_noise(SEED, 0.05, 40, B)
    sx, sz = og0.syndrome(ex, ez)
    res = {}
    for fact in (False, True):
        with _order(name, fact) as (og, gg):
            o = og.gnn_bp4(w, sx, sz, iters)
            gw = GnnBp4Weights(w, gg.device)
            prev_stream = gg.gnn_stream
            try:
                gg.set_gnn_stream(False)       # MFMA tiles on the (3,3,6)-regular graphs (the runtime-degree VALU kernel elsewhere)
                outs = [gg.gnn_bp4_decode(gw, to_gpu(sx), to_gpu(sz), iters)]
                gg.set_gnn_stream("always")    # round 4: the streaming packed-FMA kernel (regular graphs), whatever the launch size
                outs.append(gg.gnn_bp4_decode(gw, to_gpu(sx), to_gpu(sz), iters))
                gg.force_generic(True)
                outs.append(gg.gnn_bp4_decode(gw, to_gpu(sx), to_gpu(sz), iters))
            finally:
                gg.force_generic(False)
                gg.set_gnn_stream(prev_stream)
        for which, g in zip(("MFMA-tile kernel", "streaming packed-FMA kernel", "runtime-degree VALU kernel"), outs):
            for k in ("llr", "x_logit_all", "z_logit_all", "x_hat", "z_hat"):
                a, b = o[k], g[k].cpu().numpy()
                assert np.array_equal(a, b), f"{name} factored={fact} {which} {k}: max|d|={np.abs(a.astype(np.float64) - b).max()}"
        res[fact] = o
    d = np.abs(res[False]["llr"] - res[True]["llr"]).max()
    assert 0 < d <= 2e-5, d
    assert np.abs(res[False]["x_logit_all"] - res[True]["x_logit_all"]).max() <= 2e-5


def test_gnn_bp4_streaming_kernel_equals_the_mfma_kernel_on_a_chip_filling_launch():
    """FGNN_OPT_GNN_STREAM = 2 runs GNN_BP4 on the streaming packed-FMA kernel (one lane per node, v_pk_fma_f32 with scalar weight
    pairs); the default keeps the MFMA tiles, which are faster (profiles/r4_gnnbp4_stream_ab.txt).  The same float operations in the
    same order: a launch of 1 024 codewords must give the same bits on either kernel, and both equal the oracle on a sample."""
    from feedback_gnn_amd.graph import GNNBP4_SHAPES, GnnBp4Weights
    name, B, iters = "ghp882", 1024, 3
    rng = np.random.RandomState(5)
    w = []
    for shp in GNNBP4_SHAPES:
        lim = 0.6 if len(shp) == 1 else np.sqrt(6.0 / (shp[0] + shp[1]))
        w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    og, gg = oracle_library_forms(name), gpu_graph(name)
    assert gg.gnn_stream is True  # the library default
    ex, ez = gg.pauli_noise(SEED, 0.05, 0, B)
    sx, sz = gg.syndrome(ex, ez)
    gw = GnnBp4Weights(w, gg.device)
    mfma = gg.gnn_bp4_decode(gw, sx, sz, iters)
    try:
        gg.set_gnn_stream("always")
        stream = gg.gnn_bp4_decode(gw, sx, sz, iters)
    finally:
        gg.set_gnn_stream(True)
    for k in ("llr", "x_hat", "z_hat", "x_logit_all", "z_logit_all"):
        assert torch.equal(mfma[k], stream[k]), k
    idx = np.arange(0, B, 64)
    o = og.gnn_bp4(w, sx.cpu().numpy()[idx], sz.cpu().numpy()[idx], iters)
    assert np.array_equal(o["llr"], stream["llr"].cpu().numpy()[idx]) and np.array_equal(o["x_hat"], stream["x_hat"].cpu().numpy()[idx])

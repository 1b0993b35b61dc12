"""The reference's formulas term by term (the library DEFAULT since round 6) and the two opt-in re-associations, sample by sample.

libfgnn_hip evaluates decoding_q.py:254-273 (one reduce_logsumexp per edge) and feedback_gnn.py:175-184 (one Dense per edge) term by
term unless a caller switches on FGNN_OPT_BP4_SHARED_LSE / FGNN_OPT_GNN_FACTORED (include/fgnn.h).  The default parity tests check
"default kernel == literal oracle" (helpers.oracle_library_forms); tests/test_gpu_bp4_shared_lse.py and test_gpu_gnn_order.py check each
option on and off against the oracle's restatement of that form.  This file holds:

  * the kernels with both options OFF (set explicitly) equal the oracle's LITERAL restatement bit for bit through the whole sandwich;
  * what a caller who switches both options ON gets, per sample, against the default: MEASURED RATES on several windows of the sample
    stream — including the window the round-5 driver run timed, where one solved sample sits at 1.03e-4 — with honest bounds.  They are
    not guarantees: the re-associated forms are the reference's decoder statistically (include/fgnn.h), which is why they are opt-in;
  * in the waterfall (p = 0.05) the per-sample rates are recorded and bounded so that the documentation (include/fgnn.h, README) cannot
    drift from what the kernels do.
"""
import numpy as np
import pytest
import torch

from helpers import (LIBRARY_BP4_SHARED_LSE, LIBRARY_GNN_FACTORED, WEIGHTS_882, WEIGHTS_1270, code, gpu_graph, llr_const, oracle_library_forms,
                     oracle_literal_forms, oracle_reassociated_forms, to_gpu)

pytestmark = pytest.mark.gpu
SEED = 0x5EED


class _literal_kernels:
    """The GPU graph with FGNN_OPT_BP4_SHARED_LSE = FGNN_OPT_GNN_FACTORED = 0 for the duration of the block."""

    def __init__(self, name):
        self.gg = gpu_graph(name)

    def __enter__(self):
        self.prev = (self.gg.gnn_factored, self.gg.bp4_shared_lse)
        self.gg.set_gnn_factored(False)
        self.gg.set_bp4_shared_lse(False)
        return self.gg

    def __exit__(self, *exc):
        self.gg.set_gnn_factored(self.prev[0])
        self.gg.set_bp4_shared_lse(self.prev[1])


def _weights(wfile, gg):
    from feedback_gnn_amd.graph import GnnWeights
    from feedback_gnn_amd.weights_io import read_weight_list
    w = read_weight_list(wfile)
    return w, GnnWeights(w, gg.device)


@pytest.mark.parametrize("name,wfile,iters,p", [("ghp882", WEIGHTS_882, [64, 16], 0.01), ("ghp882", WEIGHTS_882, [64, 16], 0.06),
                                                ("ghp882", WEIGHTS_882, [64, 16, 16, 16], 0.10), ("ghp1270", WEIGHTS_1270, [64, 64], 0.08)])
def test_literal_kernels_equal_the_literal_oracle_through_the_sandwich(name, wfile, iters, p):
    """Both options off: every decision, round counter and marginal of the sandwich equals the oracle's term-by-term restatement."""
    B = 192
    og = oracle_literal_forms(name)
    assert og.forms == "literal" and not og.gnn_factored and not og.vn_shared_lse
    ex, ez = og.pauli_noise(SEED, p, 7000, B)
    sx, sz = og.syndrome(ex, ez)
    nl = len(iters)
    with _literal_kernels(name) as gg:
        w, gw = _weights(wfile, gg)
        o = og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), llr_const(0.05), return_llr=True)
        for shortcut in (False, True):
            gg.set_saturation_shortcut(shortcut)
            try:
                g = gg.sandwich_decode(to_gpu(sx), to_gpu(sz), iters, [gw] * (nl - 1), llr_const(0.05), return_llr=True, return_rounds=True)
            finally:
                gg.set_saturation_shortcut(True)
            assert np.array_equal(o["x_hat"], g["x_hat"].cpu().numpy()) and np.array_equal(o["z_hat"], g["z_hat"].cpu().numpy())
            assert np.array_equal(o["rounds"], g["rounds"].cpu().numpy())
            assert np.array_equal(o["llr"], g["llr"].cpu().numpy())
    # the library's default IS this restatement ...
    og_def = oracle_library_forms(name)
    assert (og_def.gnn_factored, og_def.vn_shared_lse) == (LIBRARY_GNN_FACTORED, LIBRARY_BP4_SHARED_LSE) == (False, False)
    # ... and the re-associated oracle is a different one: in the waterfall its marginals are not the literal ones
    if p >= 0.06:
        o2 = oracle_reassociated_forms(name).sandwich_decode(sx, sz, iters, [w] * (nl - 1), llr_const(0.05), return_llr=True)
        assert not np.array_equal(o["llr"], o2["llr"])


# windows of the global sample stream (first sample, in batches of 65 536): the first batch, the round-5 driver's timed batch
# (W = 5 warm-up steps: samples 327 680 .. 393 215, one solved sample at 1.03e-4 there), and six more
WINDOWS_882 = (0, 5, 1, 2, 3, 17, 100, 1000)


def test_reassociated_forms_against_the_default_at_the_benchmark_point_measured_rates():
    """BASELINE configs[2] at full size: p = 0.01, 65 536 codewords per window, (64, G, 16), fixed dataflow, eight windows of the sample
    stream.  Opt-in re-associated forms vs the default (literal) forms.  These are MEASURED RATES, NOT GUARANTEES: per window at most one
    sample with a different final decision and at most two SOLVED samples whose marginals are further apart than the north-star tolerance
    1e-4 (8.4 M samples: 3 differing decisions, 40 solved samples beyond 1e-4 — profiles/r4_forms_agreement_8M.json; the driver's round-5
    window holds one at 1.03e-4).  A universal "within 1e-4 on every solved sample" does not hold and is not asserted."""
    name, B = "ghp882", 65536
    gg = gpu_graph(name)
    _, gw = _weights(WEIGHTS_882, gg)
    gg.set_saturation_shortcut(False)
    tot = dict(decisions_differ=0, samples_gt_1e_4_solved=0, flagged_in_one_form_only=0)
    try:
        for win in WINDOWS_882:
            ex, ez = gg.pauli_noise(SEED, 0.01, win * B, B)
            sx, sz = gg.syndrome(ex, ez)
            r = gg.forms_agreement(sx, sz, [64, 16], [gw], llr_const(0.05))
            print(f"forms agreement at p = 0.01, window {win}:", {k: r[k] for k in ("decisions_differ", "max_abs_dllr_solved", "samples_gt_1e_4_solved",
                                                                                   "samples_gt_1e_4", "flagged_reassociated", "flagged_literal")})
            assert r["samples"] == B and r["decisions_differ"] <= 1, (win, r)
            assert r["samples_gt_1e_4_solved"] <= 2, (win, r)
            assert r["flagged_in_one_form_only"] <= 1 and max(r["flagged_reassociated"], r["flagged_literal"]) <= 8, (win, r)
            assert r["first_decoder"]["decisions_differ"] <= 1, (win, r)
            for k in tot:
                tot[k] += r[k]
    finally:
        gg.set_saturation_shortcut(True)
    assert (gg.gnn_factored, gg.bp4_shared_lse) == (LIBRARY_GNN_FACTORED, LIBRARY_BP4_SHARED_LSE)  # settings restored
    assert tot["decisions_differ"] <= 2 and tot["samples_gt_1e_4_solved"] <= 6, tot  # 524 288 samples; 8.4 M hold 3 and 40


def test_reassociated_forms_against_the_default_on_the_c4_code_at_p_001_measured_rates():
    """The same on the configs[3] shard shape ([[1270,28]], (64, G, 64), 32 768 codewords per window, p = 0.01), four windows; 8.4 M samples
    hold 6 differing decisions and 4 solved samples beyond 1e-4 (profiles/r4_forms_agreement_8M.json).  Measured rates, not guarantees."""
    name, B = "ghp1270", 32768
    gg = gpu_graph(name)
    _, gw = _weights(WEIGHTS_1270, gg)
    gg.set_saturation_shortcut(False)
    try:
        for win in (0, 3, 24, 500):
            ex, ez = gg.pauli_noise(SEED, 0.01, win * B, B)
            sx, sz = gg.syndrome(ex, ez)
            r = gg.forms_agreement(sx, sz, [64, 64], [gw], llr_const(0.05))
            assert r["decisions_differ"] <= 1 and r["samples_gt_1e_4_solved"] <= 2 and r["flagged_in_one_form_only"] <= 1, (win, r)
    finally:
        gg.set_saturation_shortcut(True)


def test_in_the_waterfall_the_two_forms_are_the_same_decoder_only_statistically():
    """p = 0.05, 65 536 codewords of [[882,24]]: the opt-in forms differ from the default by float32 rounding in the qubit update, BP's transient amplifies that
    on the samples that take long to converge, and a fraction of the samples ends on a different (equally valid) representative.
    Measured (profiles/r4_forms_agreement.json): 74 of 65 536 decisions differ (0.11 %), 213 samples beyond 1e-4, 12 vs 13 left flagged.
    The bounds below are what include/fgnn.h and README.md quote; the decoder's success is unchanged within binomial noise."""
    name, B = "ghp882", 65536
    gg = gpu_graph(name)
    _, gw = _weights(WEIGHTS_882, gg)
    ex, ez = gg.pauli_noise(SEED, 0.05, 0, B)
    sx, sz = gg.syndrome(ex, ez)
    r = gg.forms_agreement(sx, sz, [64, 16], [gw], llr_const(0.05))
    print("forms agreement at p = 0.05:", r)
    assert 0 < r["decisions_differ"] <= 0.005 * B, r          # measured 0.11 %: not identical, and not more than half a per cent
    assert r["samples_gt_1e_4"] <= 0.01 * B, r
    assert abs(r["flagged_reassociated"] - r["flagged_literal"]) <= 6 * max(1.0, r["flagged_in_one_form_only"]) ** 0.5 + 1, r

"""Host-side logic that needs no GPU: weight I/O, harness, sharding, the no-fallback rule."""
import io
import os
import pickle

import numpy as np
import pytest
import torch

import feedback_gnn_amd as F
from feedback_gnn_amd import weights_io
from feedback_gnn_amd.utils import PlotBER, count_block_errors, shard_range, sim_ber
from helpers import WEIGHTS_882, code


def test_bundled_weights_and_roundtrip(tmp_path):
    w = weights_io.read_weight_list(WEIGHTS_882)
    assert [a.shape for a in w] == [(40, 3), (3,), (4, 40), (40,), (40, 20), (20,), (4, 40), (40,), (40, 20), (20,), (43, 40), (40,)]
    assert all(a.dtype == np.float32 for a in w) and sum(a.size for a in w) == 3923
    p = tmp_path / "w.npz"
    weights_io.write_weight_list(w, p)
    w2 = weights_io.read_weight_list(str(p))
    assert all(np.array_equal(a, b) for a, b in zip(w, w2))
    # the reference's own file name resolves to the bundled conversion
    w3 = weights_io.read_weight_list("./sionna/fec/ldpc/weights/feedback_GNN_n882_k24_wt_4_60_iter_64_16_mixed.npy")
    assert all(np.array_equal(a, b) for a, b in zip(w, w3))


def test_restricted_unpickler_reads_tf_style_pickles_and_refuses_code(tmp_path):
    # a pickle shaped like the reference's: list of reduce(convert_to_tensor, (ndarray,))
    class FakeTensor:
        def __init__(self, a):
            self.a = a

        def __reduce__(self):
            return (_fake_convert, (self.a,))

    import sys
    import types
    mod = types.ModuleType("tensorflow.python.framework.ops")
    mod.convert_to_tensor = _fake_convert
    _fake_convert.__module__ = "tensorflow.python.framework.ops"
    _fake_convert.__qualname__ = _fake_convert.__name__ = "convert_to_tensor"
    for name in ("tensorflow", "tensorflow.python", "tensorflow.python.framework", "tensorflow.python.framework.ops"):
        sys.modules.setdefault(name, mod if name.endswith("ops") else types.ModuleType(name))
    try:
        arrays = [np.arange(6, dtype=np.float32).reshape(2, 3), np.ones(3, np.float32)]
        p = tmp_path / "tf_style.npy"
        with open(p, "wb") as f:
            pickle.dump([FakeTensor(a) for a in arrays], f)
    finally:
        for name in list(sys.modules):
            if name == "tensorflow" or name.startswith("tensorflow."):
                del sys.modules[name]
    got = weights_io.read_weight_list(str(p))
    assert all(np.array_equal(a, b) for a, b in zip(arrays, got))
    evil = tmp_path / "evil.npy"
    with open(evil, "wb") as f:
        pickle.dump([os.system], f)
    with pytest.raises(pickle.UnpicklingError):
        weights_io.read_weight_list(str(evil))


def _fake_convert(a):
    return a


def test_no_cpu_fallback():
    from feedback_gnn_amd.graph import TannerGraph
    from feedback_gnn_amd._lib import FgnnError
    with pytest.raises(FgnnError):
        TannerGraph(code("steane"), device=torch.device("cpu"))
    with pytest.raises(ValueError):
        F.QLDPCBPDecoder(code("steane"), cn_type="nonsense")


def test_public_names_mirror_the_reference():
    for name in ("QLDPCBPDecoder", "Feedback_GNN", "Sandwich_BP_GNN_Evaluation_Model", "load_weights", "save_weights",
                 "css_code", "create_QC_GHP_codes", "create_cyclic_permuting_matrix", "create_generalized_bicycle_codes",
                 "hypergraph_product", "create_rotated_surface_codes", "create_surface_codes", "hamming_code", "readAlist",
                 "sim_ber", "PlotBER", "count_block_errors", "int_mod_2"):
        assert hasattr(F, name), name
    # every QLDPC-path name sionna/fec/ldpc/__init__.py:9-20 exports (the 5G names — LDPC5GEncoder / Decoder, AllZeroEncoder, `codes` —
    # are the out-of-scope wireless stack)
    for name in ("LDPCBPDecoder", "First_Stage_BP_Model", "Second_Stage_GNN_BP_Model", "BP_BSC_Model", "GNN_BP4", "MLP", "OSD0_Decoder",
                 "BP4_OSD_Model", "BP2_OSD_Model", "create_checkerboard_toric_codes", "create_bivariate_QC_codes", "create_circulant_matrix",
                 "rep_code", "set_pcm_row"):
        assert hasattr(F, name), name


def test_mlp_helper_layer_and_set_pcm_row():
    """gnn.py:25-69: `MLP(units, activations, use_bias)` builds on the first call (glorot kernels, ones biases), applies the Dense chain to
    the last axis, and exposes its arrays in Keras' order; codes_q.py:147-150: `set_pcm_row` marks the four qubits of a plaquette."""
    m = F.MLP([40, 20], ["tanh", None], [True, False])
    x = torch.from_numpy(np.random.RandomState(0).randn(3, 5, 4).astype(np.float32))
    y = m(x)
    w = m.get_weights()
    assert [a.shape for a in w] == [(4, 40), (40,), (40, 20)] and np.all(w[1] == 1.0) and np.abs(w[0]).max() <= np.sqrt(6.0 / 44)
    ref = np.tanh(x.numpy().astype(np.float64) @ w[0] + w[1]) @ w[2]
    assert y.shape == (3, 5, 20) and np.abs(ref - y.numpy()).max() < 1e-5
    m2 = F.MLP([40, 20], ["tanh", None], [True, False])
    m2.set_weights(w)
    assert torch.equal(m2(x), y)
    with pytest.raises(ValueError):
        F.MLP([4], ["tanh", "tanh"], [True])
    with pytest.raises(NotImplementedError):
        F.MLP([4], ["gelu"], [True])
    pcm = np.zeros((2, 9), dtype=int)
    F.set_pcm_row(3, pcm, 1, 2, 2)  # wraps around: (2,2), (0,0), (0,2), (2,0)
    assert pcm[0].sum() == 0 and sorted(np.nonzero(pcm[1])[0]) == [0, 2, 6, 8]


def test_count_block_errors_and_sim_ber_qldpc():
    a = torch.tensor([[0, 0, 0], [0, 1, 0], [1, 1, 1], [0, 0, 0]])
    assert int(count_block_errors(torch.zeros_like(a), a)) == 2
    rng = np.random.RandomState(0)
    calls = []

    def mc_fun(batch_size, ebno_db):
        calls.append(ebno_db)
        fl = torch.from_numpy((rng.rand(batch_size, 5) < ebno_db / 5).astype(np.uint8))
        lg = torch.cat([fl, torch.from_numpy((rng.rand(batch_size, 2) < ebno_db / 10).astype(np.uint8))], dim=1)
        return fl, lg

    flagged, bler = sim_ber(mc_fun, [0.5, 0.2, 0.0], batch_size=200, max_mc_iter=50, num_target_block_errors=100,
                            verbose=False)
    st = sim_ber.last
    assert st["status"][0] == 4 and st["block_errors"][0] >= 100  # reached target block errors
    assert st["status"][2] == 2 and st["num_blocks"][2] == 50 * 200  # error-free point: early stop after max iter
    assert 0 < flagged[0] <= bler[0] < 1 and bler[2] == 0
    pb = PlotBER()
    pb.simulate(mc_fun, [0.4], 100, 5, legend="x", add_bler=True, num_target_block_errors=10, verbose=False)
    assert len(pb._bers) == 2 and pb._legends[1].endswith("(BLER)")
    with pytest.raises(NotImplementedError):
        sim_ber(mc_fun, [0.1], 10, 1, qldpc=False)


class _StreamModel:
    """Host stand-in with the model surface sim_ber's device-counter path needs: a counter-based sample stream (sample i is a
    pure function of i, like the Philox stream of the product), the per-batch call, mc_step and rewind."""

    def __init__(self):
        self._next = 0
        self.issued = 0

    def _rows(self, batch_size, p):
        idx = np.arange(self._next, self._next + batch_size, dtype=np.uint64)
        self._next += batch_size
        self.issued += 1
        h = (idx * np.uint64(0x9E3779B97F4A7C15) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
        fl = (h < p).astype(np.uint8)
        lg = (h < p / 3).astype(np.uint8)
        return torch.from_numpy(fl[:, None]), torch.from_numpy(np.stack([fl, lg], axis=1))

    def __call__(self, batch_size, ebno_db):
        return self._rows(batch_size, ebno_db)

    def mc_step(self, batch_size, p, counts=None):
        if counts is None:
            counts = torch.zeros(3, dtype=torch.int64)
        s, l = self._rows(batch_size, p)
        counts += torch.tensor([int(s.any(1).sum()), int(l.any(1).sum()), batch_size])
        return counts

    def rewind(self, batches, batch_size):
        self._next -= batches * batch_size


class _FusedStreamModel(_StreamModel):
    """... plus mc_steps: k batches as one draw, counters taken batch by batch (the product's fused launch)."""

    def __init__(self):
        super().__init__()
        self.fused_calls = 0

    def mc_steps(self, batch_size, p, k, counts, ring):
        self.fused_calls += 1
        s, l = self._rows(k * batch_size, p)
        for j in range(k):
            sl = slice(j * batch_size, (j + 1) * batch_size)
            counts += torch.tensor([int(s[sl].any(1).sum()), int(l[sl].any(1).sum()), batch_size])
            ring[j].copy_(counts)
        return counts


def test_sim_ber_fused_batches_equal_per_batch_path():
    """Deferred batches decoded several at a time (mc_steps) leave the same counters, status and stream position as the per-batch loop."""
    pts = [0.3, 0.05, 0.01, 0.002]
    for kw in (dict(num_target_block_errors=100), dict(num_target_bit_errors=40), dict(num_target_block_errors=100000)):
        a, b = _StreamModel(), _FusedStreamModel()
        sim_ber(a, pts, batch_size=64, max_mc_iter=300, verbose=False, early_stop=False, device_counters=False, **kw)
        ref = {k: np.array(v).copy() for k, v in sim_ber.last.items()}
        sim_ber(b, pts, batch_size=64, max_mc_iter=300, verbose=False, early_stop=False, device_counters=True, max_deferred=16,
                fuse_samples=64 * 5, **kw)
        for k in ("flag_errors", "block_errors", "num_blocks", "status"):
            assert np.array_equal(ref[k], sim_ber.last[k]), (kw, k, ref[k], sim_ber.last[k])
        assert a._next == b._next and b.fused_calls > 0
        c = _FusedStreamModel()
        sim_ber(c, pts, batch_size=64, max_mc_iter=300, verbose=False, early_stop=False, fuse_samples=0, **kw)
        assert c.fused_calls == 0 and c._next == a._next


def test_sim_ber_device_counter_path_equals_per_batch_path():
    """The deferred read-back must end every point after exactly the batch the per-batch loop ends it with, and leave the
    sample stream where that loop leaves it (so later points see the same samples): identical counters and status for every
    point, with far fewer host read-backs than batches."""
    pts = [0.3, 0.05, 0.01, 0.002]
    for kw in (dict(num_target_block_errors=100), dict(num_target_bit_errors=40), dict(num_target_block_errors=100000)):
        a, b = _StreamModel(), _StreamModel()
        sim_ber(a, pts, batch_size=64, max_mc_iter=300, verbose=False, early_stop=False, device_counters=False, **kw)
        ref = {k: np.array(v).copy() for k, v in sim_ber.last.items()}
        assert ref["device_counters"] is not None and not sim_ber.last["device_counters"]
        sim_ber(b, pts, batch_size=64, max_mc_iter=300, verbose=False, early_stop=False, device_counters=True,
                max_deferred=16, **kw)
        assert sim_ber.last["device_counters"]
        for k in ("flag_errors", "block_errors", "num_blocks", "status"):
            assert np.array_equal(ref[k], sim_ber.last[k]), (kw, k, ref[k], sim_ber.last[k])
        assert a._next == b._next  # the stream position after the run is the per-batch loop's
    assert b.issued >= a.issued  # discarded batches are the price; the counters are not affected


def test_shard_range_partitions_exactly():
    for total, world in ((10, 3), (65536, 8), (7, 8), (262144, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_training_helpers_on_cpu_tensors():
    """Adam (Keras update rule), clip_by_value and compute_bler are plain tensor code and run without a GPU."""
    import torch
    from feedback_gnn_amd.training import Adam, clip_by_value, compute_bler
    v = torch.tensor([1.0, -2.0, 3.0])
    g = torch.tensor([0.5, -0.25, 0.0])
    opt = Adam(learning_rate=lambda step: 0.1)
    ref, m, s = v.double().clone(), torch.zeros(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
    for t in range(1, 5):
        opt.apply_gradients([(g, v), (None, v)])
        m = 0.9 * m + 0.1 * g.double()
        s = 0.999 * s + 0.001 * g.double() ** 2
        ref = ref - 0.1 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (s.sqrt() + 1e-7)
    assert torch.allclose(v.double(), ref, atol=1e-6) and opt.iterations == 4
    assert clip_by_value(torch.tensor([-20.0, 3.0, 11.0]), -10, 10).tolist() == [-10.0, 3.0, 10.0]
    b = torch.zeros((4, 5), dtype=torch.uint8)
    bh = b.clone(); bh[1, 2] = 1; bh[3, 0] = 1; bh[3, 4] = 1
    assert compute_bler(b, bh) == 0.5


def test_plotber_bookkeeping_and_figure(tmp_path):
    """PlotBER.add / remove / reset / properties / __call__ (sionna/utils/plotting.py:207-310, :449-504)."""
    pytest.importorskip("matplotlib")
    from feedback_gnn_amd.utils import PlotBER
    pb = PlotBER("t")
    pb.add([0.1, 0.09], [1e-2, 1e-3], is_bler=True, legend="a")
    pb.add(np.array([0.1, 0.09]), np.array([2e-2, 3e-3]), legend="b")
    assert pb.legend == ["a", "b"] and pb.is_bler == [True, False] and len(pb.ber) == 2 and len(pb.snr) == 2
    pb.title = "u"
    assert pb.title == "u"
    with pytest.raises(AssertionError):
        pb.add([0.1], [1e-2, 1e-3])
    out = tmp_path / "fig.png"
    fig = pb(snr_db=[0.1, 0.08], ber=[5e-2, 5e-4], legend="c", is_bler=True, ylim=(1e-5, 1), save_fig=True, path=str(out))
    assert out.exists() and out.stat().st_size > 1000
    assert len(fig.axes[0].lines) == 3 and fig.axes[0].get_ylabel() == "BER / BLER"
    fig2 = pb(show_ber=False)
    assert len(fig2.axes[0].lines) == 1 and fig2.axes[0].get_ylabel() == "BLER"
    pb.remove(0)
    assert pb.legend == ["b"]
    pb.reset()
    assert pb.ber == [] and pb.snr == []


def test_every_kernel_that_evaluates_a_log_installs_the_lds_table_first():
    """fg_log / fg_log1p (hence fg_softplus, fg_lse2, fg_phi, fg_phi_gnn, fg_atanh) read a 64-entry table from LDS on the device; it is
    valid only after FG_LOG_TAB_SETUP() — which contains a barrier, so it must be the kernel's first statement, ahead of any divergent
    return.  A kernel that forgets it reads uninitialised LDS and silently produces wrong bits: this walks every __global__ function
    of csrc/ and, for each one whose body (or a device function it calls, transitively) reaches the log family, requires the setup
    macro before any other statement that is not a declaration."""
    import glob
    import os
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "feedback_gnn_amd", "csrc")
    text = {p: open(p).read() for p in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))}
    LOG_FAMILY = {"fg_log", "fg_log1p", "fg_softplus", "fg_lse2", "fg_phi", "fg_phi_gnn", "fg_atanh"}

    def strip(src):  # comments and string literals out
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        return re.sub(r'"(\\.|[^"\\])*"', '""', src)

    funcs = {}  # name -> [(file, is_kernel, body)]
    head = re.compile(r"\b(__global__|__device__|FG_FN)\b")
    for path, raw in text.items():
        src = strip(raw)
        for m in head.finditer(src):
            # walk to the '{' (definition) or ';' (declaration) at parenthesis depth 0; the function's name is the identifier in front
            # of the LAST top-level parenthesis group before it (attributes like __launch_bounds__(256) come earlier)
            depth, k, name, group_start = 0, m.end(), None, None
            while k < len(src):
                ch = src[k]
                if ch == "(":
                    if depth == 0:
                        group_start = k
                    depth += 1
                elif ch == ")":
                    depth -= 1
                    if depth == 0:
                        pre = re.search(r"([A-Za-z_]\w*)\s*$", src[m.end():group_start])
                        if pre and pre.group(1) not in ("__launch_bounds__", "__attribute__", "amdgpu_waves_per_eu"):
                            name = pre.group(1)
                elif depth == 0 and ch in "{;=":
                    break
                k += 1
            if k >= len(src) or src[k] != "{" or name is None:
                continue
            depth, e = 1, k + 1
            while depth and e < len(src):
                depth += {"{": 1, "}": -1}.get(src[e], 0)
                e += 1
            funcs.setdefault(name, []).append((os.path.basename(path), m.group(1) == "__global__", src[k + 1:e - 1]))
    assert sum(1 for v in funcs.values() for f in v if f[1]) >= 15, "kernel scan found too few __global__ functions"

    def calls(body):
        return set(re.findall(r"\b([A-Za-z_]\w*)\s*(?:<[^;(){}]*>)?\s*\(", body)) | set(re.findall(r"\b(?:MX|Mx<\w+>)::(\w+)", body))

    reach = {}

    def reaches_log(name, seen=()):
        if name in LOG_FAMILY:
            return True
        if name in reach:
            return reach[name]
        if name in seen or name not in funcs:
            return False
        r = any(reaches_log(c, seen + (name,)) for _, _, body in funcs[name] for c in calls(body))
        reach[name] = r
        return r

    # Mx<false>::softplus / lse2 / phi forward to the fg_ routines
    funcs.setdefault("softplus", []).append(("fgnn_bp4.hip", False, "fg_softplus("))
    funcs.setdefault("lse2", []).append(("fgnn_bp4.hip", False, "fg_lse2("))
    funcs.setdefault("phi", []).append(("fgnn_bp4.hip", False, "fg_phi("))
    checked = []
    for name, defs in funcs.items():
        for fname, is_kernel, body in defs:
            if not is_kernel or not any(reaches_log(c) for c in calls(body)):
                continue
            stmts = [st.strip() for st in re.split(r";", body) if st.strip()]
            first_real = None
            for st in stmts:
                if st.startswith("FG_LOG_TAB_SETUP"):
                    first_real = "setup"
                    break
                if re.match(r"(static_assert|using|constexpr|const|extern|typedef)\b", st):
                    continue  # declarations and compile-time statements may precede the macro; anything that executes may not
                first_real = st
                break
            assert first_real == "setup", f"{fname}: kernel {name} reaches the log family but its first statement is `{str(first_real)[:60]}`"
            checked.append(name)
    assert {"bp4_kernel", "bp2_kernel"} <= set(checked) and len(checked) >= 5, checked

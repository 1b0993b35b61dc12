"""Feedback_GNN and the BP/GNN sandwich evaluation model on MI355X.

Drop-ins for `sionna.fec.ldpc.Feedback_GNN` and `Sandwich_BP_GNN_Evaluation_Model`
(/root/reference sionna/fec/ldpc/feedback_gnn.py:20-188, :232-361), `First_Stage_BP_Model` / `Second_Stage_GNN_BP_Model`
(:364-463, forward only) and `sionna.channel.Pauli` (sionna/channel/pauli.py:80-108).  All device work goes through
libfgnn_hip.so (include/fgnn.h); tensors are torch tensors on the HIP device.
"""
import numpy as np
import torch

from .graph import ACTIVATIONS, REDUCE_OPS, SHIPPED_GNN_CONFIG, GnnWeights, TannerGraph, gnn_weight_shapes
from .weights_io import read_weight_list

_W_SHAPES = [(40, 3), (3,), (4, 40), (40,), (40, 20), (20,), (4, 40), (40,), (40, 20), (20,), (43, 40), (40,)]


def _glorot_uniform(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


class Feedback_GNN:
    """One CN->VN message-passing layer that maps BP marginals + soft syndromes to new channel LLRs.

    Constructor as feedback_gnn.py:21-28.  The architecture the reference trains and ships (n882.py:45-51:
    num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True) runs on the
    MFMA kernel and has a reverse pass; any other setting with num_msg_dims <= 32, num_hidden_units <= 96,
    num_mlp_layers <= 4, reduce_op in sum/mean/max/min, activation in tanh/relu/sigmoid/linear runs on the runtime-shaped
    kernel (forward only).  ``get_weights()`` / ``set_weights()`` / ``load_weights`` use Keras' array order for the setting.

    Call: ``G((h_vn[bs,n,3], logit_hx[m_x,bs], logit_hz[m_z,bs], syndrome_x[m_x,bs], syndrome_z[m_z,bs]))``
    → ``[bs,n,3]`` (new llrx, llry, llrz).
    """

    def __init__(self, code, num_msg_dims, num_hidden_units, num_mlp_layers, reduce_op="mean", activation="tanh",
                 use_bias=False, device=None, graph=None, seed=0):
        cfg = (int(num_msg_dims), int(num_hidden_units), int(num_mlp_layers), reduce_op, activation, bool(use_bias))
        if reduce_op not in REDUCE_OPS:
            raise ValueError("unknown reduce operation")  # feedback_gnn.py:148
        if activation not in ACTIVATIONS:
            raise NotImplementedError(f"activation {activation!r}: the HIP kernels implement tanh, relu, sigmoid and linear")
        if not (1 <= cfg[0] <= 32 and 1 <= cfg[2] <= 4 and (cfg[2] == 1 or 1 <= cfg[1] <= 96)):
            raise NotImplementedError("the runtime-shaped kernel takes num_msg_dims <= 32, num_hidden_units <= 96, "
                                      f"1 <= num_mlp_layers <= 4; got {cfg[:3]}")
        self._config = cfg
        self._num_msg_dims, self._num_hidden_units, self._num_mlp_layers = cfg[:3]
        self._reduce_op, self._activation, self._use_bias = cfg[3:]
        self.graph = graph if graph is not None else TannerGraph(code, stage_one=True, device=device)
        self._num_vn = self.graph.n
        self._num_cn_x, self._num_cn_z = self.graph.m_x, self.graph.m_z
        self._num_edges_x, self._num_edges_z = self.graph.E_x, self.graph.E_z
        # Keras initialisers of feedback_gnn.py:115-128 / gnn.py:55-61: Dense kernels glorot-uniform,
        # biases ones, the output layer's kernel zeros.
        rng = np.random.RandomState(seed)
        self._shapes = gnn_weight_shapes(*cfg[:3], use_bias=cfg[5])
        w = []
        for shp in self._shapes:
            w.append(np.ones(shp, np.float32) if len(shp) == 1 else _glorot_uniform(rng, shp))
        w[0] = np.zeros(self._shapes[0], np.float32)
        self._weights = self._vars = self._var_versions = None
        self.set_weights(w)

    @property
    def is_shipped_architecture(self):
        """True for num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, mean, tanh, bias: the streaming / MFMA kernels and their
        specialised reverse pass; every other setting runs the runtime-shaped kernels, forward and (round 4) reverse."""
        return self._config == SHIPPED_GNN_CONFIG

    def get_weights(self):
        self._sync()
        return [a.copy() for a in self._weights.arrays]

    def set_weights(self, weights):
        self._weights = GnnWeights(list(weights), self.graph.device, self._config)
        self._vars = self._var_versions = None

    @property
    def trainable_weights(self):
        """The 12 arrays as float32 device tensors in Keras order (Layer.trainable_weights).  An optimizer updates them in
        place; the device tables of the kernels are rebuilt from them before the next call."""
        if self._vars is None:
            self._vars = [torch.from_numpy(a.copy()).to(self.graph.device) for a in self._weights.arrays]
            self._var_versions = [v._version for v in self._vars]
        return self._vars

    trainable_variables = trainable_weights

    def _sync(self):
        if self._vars is not None and [v._version for v in self._vars] != self._var_versions:
            self._weights = GnnWeights([v.detach().cpu().numpy() for v in self._vars], self.graph.device, self._config)
            self._var_versions = [v._version for v in self._vars]

    @property
    def device_weights(self):
        self._sync()
        return self._weights

    def count_params(self):
        return int(sum(int(np.prod(s)) for s in self._shapes))

    def __call__(self, inputs):
        h_vn, logit_hx, logit_hz, syndrome_x, syndrome_z = inputs
        g = self.graph
        dev = g.device
        h_vn = torch.as_tensor(h_vn, device=dev, dtype=torch.float32)
        if h_vn.dim() != 3 or h_vn.shape[1:] != (g.n, 3):
            raise ValueError(f"h_vn must have shape [batch_size, {g.n}, 3], got {tuple(h_vn.shape)}")
        llr = h_vn.permute(0, 2, 1).contiguous()

        def cn(t, rows, dtype):
            t = torch.as_tensor(t, device=dev)
            if t.dim() != 2 or t.shape[0] != rows or t.shape[1] != h_vn.shape[0]:
                raise ValueError(f"expected shape [{rows}, {h_vn.shape[0]}], got {tuple(t.shape)}")
            if dtype == torch.uint8:
                return (t.to(torch.int64) & 1).to(torch.uint8).t().contiguous()
            return t.to(torch.float32).t().contiguous()

        out = g.feedback_gnn(self.device_weights, llr, cn(logit_hx, g.m_x, torch.float32), cn(logit_hz, g.m_z, torch.float32),
                             cn(syndrome_x, g.m_x, torch.uint8), cn(syndrome_z, g.m_z, torch.uint8))
        return out.permute(0, 2, 1).contiguous()

    call = __call__

    def build(self, input_shape=None):
        """Keras builds the Dense layers on the first call (feedback_gnn.py:110-128); here the weights exist after construction."""


def load_weights(system, model_path):
    """`load_weights(G, path)` of gnn.py:774-791: reads the reference's pickle (restricted
    unpickler, no TensorFlow) or this package's .npz and calls ``system.set_weights``."""
    system.set_weights(read_weight_list(model_path))


class Pauli:
    """i.i.d. Pauli channel (sionna/channel/pauli.py:98-108) and, with ``wt=True``, the fixed-weight channel of :80-97.

    Native call (what the models here use): ``Pauli(graph, seed=, wt=)(batch_size, p[, first_sample])`` → (noise_x, noise_z) uint8
    [bs, n].  The stream is counter-based (Philox4x32-10 keyed by seed and global sample index) so any sharding of a batch over
    GPUs sees the same samples.

    The reference's call (pauli.py:72-117) is accepted too: ``Pauli(wt=False)([cx, cz, px, py, pz])`` or, with ``wt=True``,
    ``([cx, cz, wt])`` — ``cx`` gives the shape ``[bs, n]`` — returning bool tensors ``(noise_x, noise_z)`` when ``cz`` is None and
    ``(y_x, y_z, noise_x, noise_z)`` otherwise.  Any triple ``(px, py, pz)`` is taken as the reference takes it (pauli.py:98-108:
    ``noise_x = u < px``, ``noise_z = (u >= px - py) & (u < px + pz - py)`` in float32, nothing validated); the depolarizing split
    of every caller in the reference is ``px = pz = 2p/3, py = p/3`` (feedback_gnn.py:298, bp_osd.py:107).  Successive
    reference-style calls draw successive samples of the stream.

    SEEDING.  The reference draws from TensorFlow's global generator and never repeats; here the stream is a pure function of
    ``(seed, sample index)``.  A channel built WITHOUT an explicit ``seed=`` takes ``0x5EED + k`` where k counts the seed-less
    channels constructed in this process, so two channels side by side (or one built inside a loop) do not replay each other's
    samples; pass ``seed=`` for a reproducible stream."""

    _unseeded = 0  # seed-less instances constructed so far in this process

    def __init__(self, graph=None, seed=None, wt=False, dtype=None, device=None, **kwargs):
        if isinstance(graph, bool):  # Pauli(True) would be dtype in the reference; be lenient with a positional wt
            graph, wt = None, graph
        self.graph = graph
        if seed is None:
            seed = 0x5EED + Pauli._unseeded
            Pauli._unseeded += 1
        self.seed = int(seed)
        self.wt = bool(wt)
        self._device = device
        self._next = 0  # sample-stream position of the reference-style calls

    def _noise(self, B, n, p, first_sample, out, device):
        """``p``: error rate (depolarizing split), a triple (px, py, pz), or with wt=True the weight."""
        triple = isinstance(p, (tuple, list))
        if self.graph is not None:
            if self.wt:
                return self.graph.pauli_noise_wt(self.seed, int(p), first_sample, int(B), out=out)
            if triple:
                return self.graph.pauli_noise_xyz(self.seed, p[0], p[1], p[2], first_sample, int(B), out=out)
            return self.graph.pauli_noise(self.seed, p, first_sample, int(B), out=out)
        # graph-less: the byte kernels need only n and a device
        from . import _lib
        from .graph import _ptr, _stream, check
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ex = torch.empty((B, n), dtype=torch.uint8, device=dev)
        ez = torch.empty((B, n), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            if self.wt:
                check(_lib.lib().fgnn_pauli_noise_wt(self.seed, int(p), int(first_sample), int(B), int(n), _ptr(ex), _ptr(ez), _stream(dev)))
            elif triple:
                check(_lib.lib().fgnn_pauli_noise_xyz(self.seed, float(np.float32(p[0])), float(np.float32(p[1])), float(np.float32(p[2])),
                                                      int(first_sample), int(B), int(n), _ptr(ex), _ptr(ez), _stream(dev)))
            else:
                check(_lib.lib().fgnn_pauli_noise(self.seed, float(np.float32(p)), int(first_sample), int(B), int(n), _ptr(ex), _ptr(ez),
                                                  _stream(dev)))
        return ex, ez

    def __call__(self, batch_size, p=None, first_sample=0, out=None):
        """``p`` = physical error rate, or with ``wt=True`` the exact number of erroneous qubits per sample
        (pauli.py:80-97: positions uniform without replacement, X/Y/Z with probability 1/3 each).  ``out=(noise_x, noise_z)``
        writes into given [batch_size, n] buffers.  A list / tuple as first argument is the reference's ``inputs``."""
        if isinstance(batch_size, (list, tuple)):
            return self._reference_call(batch_size)
        if self.graph is None:
            raise ValueError("Pauli(batch_size, p) needs the channel to be built on a graph: Pauli(graph, ...)")
        return self._noise(batch_size, self.graph.n, p, first_sample, out, None)

    def _reference_call(self, inputs):
        if self.wt:
            cx, cz, level = inputs
            level = int(level)
        else:
            cx, cz, px, py, pz = inputs
            px, py, pz = float(px), float(py), float(pz)
            level = 3.0 * py
            # the depolarizing split px = pz = 2p/3, py = p/3 of the reference's own callers (feedback_gnn.py:298: formed from a float32
            # p inside the graph) takes the one-parameter entry point, whose thresholds are formed from p in float32 the same way —
            # the stream of the native call; any other triple goes through as given (pauli.py:98-108)
            if abs(px - 2.0 * level / 3.0) > 1e-6 * max(level, 1e-30) or abs(pz - px) > 1e-6 * max(level, 1e-30):
                level = (px, py, pz)
        cx_t = torch.as_tensor(cx)
        if cx_t.dim() != 2:
            raise ValueError("cx must have shape [batch_size, n]")
        B, n = int(cx_t.shape[0]), int(cx_t.shape[1])
        dev = self.graph.device if self.graph is not None else (cx_t.device if cx_t.is_cuda else self._device)
        ex, ez = self._noise(B, n, level, self._next, None, dev)
        self._next += B
        noise_x, noise_z = ex.bool(), ez.bool()
        if cx is not None and cz is not None:
            return (torch.as_tensor(cx, device=ex.device).bool() ^ noise_x, torch.as_tensor(cz, device=ex.device).bool() ^ noise_z,
                    noise_x, noise_z)
        return noise_x, noise_z

    call = __call__


class Sandwich_BP_GNN_Evaluation_Model:
    """BP, then (GNN, BP) x (num_layers-1) with per-sample masking, on depolarizing noise.

    Constructor as feedback_gnn.py:265: ``(code, decoders, feedbacks, num_layers=4, wt=False, p0=0.05)``.
    ``model(batch_size, p)`` → ``(s_hat[bs, m_z+m_x], ls_hat[bs, rows(hx_perp)+rows(hz_perp)])``: residual
    syndrome and residual logical syndrome of noise XOR estimate (feedback_gnn.py:343-361); a sample is
    "flagged" iff its s_hat row is non-zero and a block error iff its ls_hat row is non-zero.

    MI355X-native extras (keyword-only): ``seed``; ``compact`` (run the feedback rounds only on samples
    still flagged — identical decisions, hence identical ``s_hat`` / ``ls_hat`` / counters; the intermediate marginals of a
    sample that left the flagged set are those of the last decoder that ran on it, see `TannerGraph.sandwich_decode`); ``rank``/``world_size`` shard the global sample stream;
    ``model.mc_step(batch_size, p, counts)`` accumulates (#flagged, #block errors, #samples) into a
    device int64[3] without any host synchronisation.
    """

    def __init__(self, code, decoders, feedbacks, num_layers=4, wt=False, p0=0.05, *, seed=0x5EED, compact=False,
                 rank=0, world_size=1, output_dtype=torch.uint8, streams=1):
        if wt and p0 is None:
            raise ValueError("wt=True needs an explicit p0: the second argument of call() is then an error weight, not a rate")
        if len(decoders) < num_layers or len(feedbacks) < num_layers - 1:
            raise ValueError("need num_layers decoders and num_layers-1 feedbacks")
        self.k, self.n = code.K, code.N
        self.hx, self.hz, self.lx, self.lz = code.hx, code.hz, code.lx, code.lz
        self.hx_perp, self.hz_perp = code.hx_perp, code.hz_perp
        self.code_name = code.name
        self.num_checks = code.hx.shape[0] + code.hz.shape[0]
        self.decoders, self.feedbacks, self.num_layers = decoders, feedbacks, int(num_layers)
        self.wt, self.p0 = wt, p0
        self.graph = decoders[0].graph
        if not self.graph.stage_one:
            raise ValueError("the sandwich needs stage_one decoders (decoding_q.py:792-793)")
        self.channel = Pauli(self.graph, seed=seed, wt=wt)
        self.compact = bool(compact)
        self.rank, self.world_size = int(rank), int(world_size)
        self.output_dtype = output_dtype
        self._next_sample = 0
        # ``streams`` > 1: `mc_step` issues consecutive (independent) batches alternately on that many side streams, each with its own
        # workspace, so that one batch's kernels fill the SIMDs that the prologues, epilogues and kernel tails of the other leave idle —
        # at the reference's batch size of 5 000 the bare loop goes from 0.90 to 0.99 of the chip's large-batch rate
        # (profiles/r4z_batch_sizes.txt).  The counters are updated atomically on the device; `join()` orders the caller's stream
        # behind all side streams (call it before reading the counters).
        self.streams = max(1, int(streams))
        self._side_streams = [torch.cuda.Stream(device=self.graph.device) for _ in range(self.streams)] if self.streams > 1 else []
        self._slot = 0
        self._workspaces = [None] * self.streams
        self._ws_batches = [-1] * self.streams

    def _llr_const(self, p):
        p0 = np.float32(p if self.p0 is None else self.p0)
        return float(np.log(np.float32(3.0) * (np.float32(1.0) - p0) / p0, dtype=np.float32))  # (:311-312)

    def _take_samples(self, batch_size):
        """Global sample indices of this rank's next batch: consecutive blocks of world_size*batch_size."""
        first = self._next_sample + self.rank * batch_size
        self._next_sample += self.world_size * batch_size
        return first

    def next_sample_range(self, batch_size):
        """``(first, last)``: the half-open range of global Philox sample indices this rank's NEXT batch of ``batch_size`` will draw
        (rank r of a world of W takes block r of every W consecutive blocks) — without advancing the stream."""
        first = self._next_sample + self.rank * int(batch_size)
        return first, first + int(batch_size)

    def decode(self, batch_size, p, first_sample=None, noise=None, _slot=None):
        """Noise -> syndromes -> sandwich.  Returns dict(noise_x, noise_z, x_hat, z_hat).  ``noise=(noise_x, noise_z)``: decode
        given device arrays instead of drawing ``batch_size`` samples."""
        B = int(batch_size)
        g = self.graph
        if _slot is None:  # a direct call runs on the caller's stream: behind whatever `mc_step` has in flight on the side streams
            self.join()
        if noise is None:
            first = self._take_samples(B) if first_sample is None else int(first_sample)
            ex, ez = self.channel(B, p, first)
        else:
            ex, ez = noise
        sx, sz = g.syndrome(ex, ez)
        k = _slot if _slot is not None else 0
        if self._ws_batches[k] < B:  # the largest workspace seen serves every smaller batch
            self._workspaces[k] = g.sandwich_workspace(B)
            self._ws_batches[k] = B
        L = self.num_layers
        out = g.sandwich_decode(sx, sz, [d.num_iter for d in self.decoders[:L]],
                                [f.device_weights for f in self.feedbacks[:L - 1]], self._llr_const(p),
                                factors=[d.normalization_factor for d in self.decoders[:L]],
                                cn_types=[d.cn_type for d in self.decoders[:L]], compact=self.compact,
                                workspace=self._workspaces[k])
        out["noise_x"], out["noise_z"] = ex, ez
        return out

    def __call__(self, batch_size, ebno_db=None, **kw):
        p = kw.get("p", ebno_db)
        o = self.decode(batch_size, p)
        s_hat, ls_hat, _ = self.graph.residual(o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"], want_arrays=True)
        if self.output_dtype != torch.uint8:
            s_hat, ls_hat = s_hat.to(self.output_dtype), ls_hat.to(self.output_dtype)
        return s_hat, ls_hat

    call = __call__

    def failures(self, batch_size, p):
        """Noise of the samples the whole sandwich fails to bring back to the syndrome — what the reference's
        notebook model `BP4_Error_Model` returns (examples/Generate_dataset.ipynb cells 1, 5, 10): ``(noise_x[err], noise_z[err])``."""
        o = self.decode(batch_size, p)
        _, _, flags = self.graph.residual(o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"], want_arrays=False)
        err = (flags & 1).bool()
        return o["noise_x"][err], o["noise_z"][err]

    def rewind(self, batches, batch_size):
        """Give back the last ``batches`` batches of ``batch_size`` samples per rank of the global sample stream (sim_ber's
        deferred read-back discards batches issued beyond the one a point ends with)."""
        self._next_sample -= int(batches) * self.world_size * int(batch_size)
        if self._next_sample < 0:
            raise ValueError("rewind beyond the start of the sample stream")

    def mc_step(self, batch_size, p, counts=None):
        """One Monte-Carlo batch with on-device counting (sim_ber's qldpc branch, misc.py:647-669): ``counts`` (device int64[3],
        created zeroed when None) += (#flagged, #block errors, #samples).  No host synchronisation."""
        if counts is None:
            counts = torch.zeros(3, dtype=torch.int64, device=self.graph.device)
        if self.streams > 1:
            side = self._side_streams[self._slot]
            side.wait_stream(torch.cuda.current_stream(self.graph.device))  # whatever the caller queued (e.g. zeroing counts) comes first
            with torch.cuda.stream(side):
                o = self.decode(batch_size, p, _slot=self._slot)
                _, _, flags = self.graph.residual(o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"], want_arrays=False)
                self.graph.count_flags(flags, counts)  # atomic adds: batches in flight on several streams share the counters
            self._slot = (self._slot + 1) % self.streams
            return counts
        o = self.decode(batch_size, p)
        _, _, flags = self.graph.residual(o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"], want_arrays=False)
        return self.graph.count_flags(flags, counts)

    def mc_graph(self, batch_size, p, steps, counts):
        """``steps`` consecutive Monte-Carlo batches (noise -> syndromes -> sandwich -> residual -> counters, what ``steps`` calls of
        `mc_step` enqueue) captured ONCE into a hipGraph; returns ``replay()``, a callable that runs them again on the NEXT
        ``steps * world_size * batch_size`` samples of the stream with one graph launch and no other host work — the reference's
        ``while`` loop of misc.py:636-738 with the host out of it.  The stream position lives in an 8-byte device counter that every
        captured noise launch reads (fgnn_pauli_noise_dev) and the graph's last node advances; the host's own position is advanced
        by `replay()`, so `mc_step` / `decode` calls in between continue where the graph left off.  ``counts`` (device int64[3]) is
        the persistent accumulator the captured counting kernels add to.  For batches so small that a step is launch-bound
        (BASELINE configs[0]: 256 codewords, five launches of 5-110 us); needs ``compact=False`` and ``streams=1``.  The graph owns its
        workspace and holds the weight objects it captured (``replay.workspace`` / ``replay.weights``): later calls on the model with
        larger batches cannot pull memory from under it; weights loaded into a feedback AFTER the capture are not seen by the graph."""
        if self.compact or self.streams > 1 or self.channel.wt:
            raise ValueError("mc_graph captures the fixed i.i.d. dataflow: compact=False, streams=1, wt=False")
        B, steps, g = int(batch_size), int(steps), self.graph
        if steps < 1 or B < 1:
            raise ValueError("steps and batch_size must be >= 1")
        if counts.device != g.device or counts.dtype != torch.int64 or tuple(counts.shape) != (3,):
            raise ValueError(f"counts must be int64[3] on {g.device}")
        ctr = torch.tensor([self._next_sample], dtype=torch.int64, device=g.device)
        span = self.world_size * B
        L = self.num_layers
        iters = [d.num_iter for d in self.decoders[:L]]
        weights = [f.device_weights for f in self.feedbacks[:L - 1]]
        factors = [d.normalization_factor for d in self.decoders[:L]]
        cn_types = [d.cn_type for d in self.decoders[:L]]
        # The graph bakes raw device pointers in: it gets a workspace of its OWN (the model's is re-bound whenever a later decode / mc_step /
        # mc_steps asks for a larger batch, and the old tensor would go back to the caching allocator while every replay still writes
        # to it), and the closure below keeps that workspace and the weight objects alive for as long as the replay callable lives
        ws = g.sandwich_workspace(B)

        def body(acc):
            for j in range(steps):
                ex, ez = g.pauli_noise(self.channel.seed, p, j * span + self.rank * B, B, first_dev=ctr)
                sx, sz = g.syndrome(ex, ez)
                o = g.sandwich_decode(sx, sz, iters, weights, self._llr_const(p), factors=factors, cn_types=cn_types, compact=False,
                                      workspace=ws)
                _, _, flags = g.residual(ex, ez, o["x_hat"], o["z_hat"], want_arrays=False)
                g.count_flags(flags, acc)

        # once eagerly (a capture must not be the first launch of a kernel: code objects load lazily) into a scratch accumulator,
        # on a side stream as torch's capture protocol asks; the device counter is not advanced, so the capture starts at the same samples
        side = torch.cuda.Stream(device=g.device)
        side.wait_stream(torch.cuda.current_stream(g.device))
        with torch.cuda.stream(side):
            body(torch.zeros(3, dtype=torch.int64, device=g.device))
        torch.cuda.current_stream(g.device).wait_stream(side)
        torch.cuda.synchronize(g.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body(counts)
            ctr.add_(steps * span)

        expected = [self._next_sample]

        def replay():
            if self._next_sample != expected[0]:  # mc_step / decode / rewind moved the stream in between: tell the device counter
                ctr.fill_(self._next_sample)
            graph.replay()
            self._next_sample += steps * span
            expected[0] = self._next_sample

        replay.graph, replay.steps, replay.counter = graph, steps, ctr
        # what the captured kernels point at (advisor, round 5): pinned on the callable.  The feedbacks' weight handles are fixed at capture
        # time — `load_weights` / `set_weights` on a Feedback_GNN afterwards uploads NEW device arrays that this graph does not see
        replay.workspace, replay.weights, replay.counts = ws, weights, counts
        return replay

    def join(self):
        """``streams`` > 1: make the caller's current stream wait for every batch `mc_step` has issued on the side streams (a
        device-side wait, no host synchronisation).  Call it before the counters are read or zeroed."""
        cur = torch.cuda.current_stream(self.graph.device)
        for s in self._side_streams:
            cur.wait_stream(s)


    def mc_steps(self, batch_size, p, num_batches, counts, ring):
        """``num_batches`` consecutive Monte-Carlo batches of ``batch_size`` decoded as ONE launch over ``num_batches * batch_size``
        samples — the very samples (global Philox indices) that ``num_batches`` calls of `mc_step` would draw on this rank — with the
        counters taken batch by batch: ``ring[j]`` (device int64 [num_batches, 3]) = the counters after batch j, ``counts`` = after the
        last.  A harness that keeps the reference's batch size (n882.py:45: 5 000) then runs at the rate of a full-chip batch while
        its stopping rule still sees every batch boundary (`sim_ber`).  No host synchronisation."""
        k, bs = int(num_batches), int(batch_size)
        g = self.graph
        self.join()  # (streams > 1) batches issued by mc_step on the side streams share the workspaces this launch uses
        if self.world_size == 1:
            noise = None
            first = self._take_samples(k * bs)
        else:  # this rank's batches are bs-sized blocks world_size * bs apart in the global stream
            ex = torch.empty((k * bs, g.n), dtype=torch.uint8, device=g.device)
            ez = torch.empty((k * bs, g.n), dtype=torch.uint8, device=g.device)
            for j in range(k):
                self.channel(bs, p, self._take_samples(bs), out=(ex[j * bs:(j + 1) * bs], ez[j * bs:(j + 1) * bs]))
            noise, first = (ex, ez), None
        o = self.decode(k * bs, p, first_sample=first, noise=noise)
        _, _, flags = g.residual(o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"], want_arrays=False)
        return g.count_flags_batches(flags, bs, counts, ring)


class First_Stage_BP_Model:
    """``First_Stage_BP_Model(code, decoder, p0=0.05)``; ``model(noise_x, noise_z)`` → ``(h_vn[bs,n,3], logit_hx_perp, logit_hz_perp)``:
    the first BP block of the training pipeline on GIVEN noise (feedback_gnn.py:364-392)."""

    def __init__(self, code, decoder, p0=0.05):
        self.hx, self.hz, self.decoder, self.p0 = code.hx, code.hz, decoder, p0

    def __call__(self, noise_x, noise_z):
        g = self.decoder.graph
        ex = torch.as_tensor(noise_x, device=g.device).to(torch.uint8).contiguous()
        ez = torch.as_tensor(noise_z, device=g.device).to(torch.uint8).contiguous()
        sx, sz = g.syndrome(ex, ez)
        p0 = np.float32(self.p0)
        L = float(np.log(np.float32(3.0) * (np.float32(1.0) - p0) / p0, dtype=np.float32))
        d = self.decoder
        o = g.bp4_decode(sx, sz, d.num_iter, d.cn_type, d.normalization_factor, llr_const=L)
        return o["llr"].permute(0, 2, 1).contiguous(), o["x_logit"].t(), o["z_logit"].t()

    call = __call__


class Second_Stage_GNN_BP_Model:
    """The training objective (feedback_gnn.py:395-463): GNN → BP with per-iteration soft syndromes →
    ``loss = sum_{i=loss_from}^{num_iter-1} BCE(1-syndrome_z, x_logit_i) + BCE(1-syndrome_x, z_logit_i)`` plus the residual
    check.  ``model(noise_x, noise_z, h_vn, logit_hx_perp, logit_hz_perp)`` → ``(s_hat, ls_hat, loss)``.

    The reference obtains the weight gradients with ``tf.GradientTape`` around this call; here
    ``model.value_and_grad(...)`` → ``(s_hat, ls_hat, loss, grads)`` runs the hand-written reverse pass
    (fgnn_bp4_backward, fgnn_feedback_gnn_backward) and returns the 12 gradients in the order of
    ``model.trainable_weights`` — see ``feedback_gnn_amd.training`` for the optimizer and the loop of Feedback_GNN.ipynb."""

    def __init__(self, code, feedback, decoder, num_iter=16, trainable=True, loss_from=8):
        self.feedback, self.decoder, self.num_iter, self.loss_from, self.trainable = feedback, decoder, int(num_iter), int(loss_from), trainable

    @property
    def trainable_weights(self):
        return self.feedback.trainable_weights if self.trainable else []

    trainable_variables = trainable_weights

    def _inputs(self, noise_x, noise_z, h_vn, logit_hx_perp, logit_hz_perp):
        g = self.decoder.graph
        ex = torch.as_tensor(noise_x, device=g.device).to(torch.uint8).contiguous()
        ez = torch.as_tensor(noise_z, device=g.device).to(torch.uint8).contiguous()
        sx, sz = g.syndrome(ex, ez)
        return g, ex, ez, sx, sz

    def __call__(self, noise_x, noise_z, h_vn, logit_hx_perp, logit_hz_perp):
        g, ex, ez, sx, sz = self._inputs(noise_x, noise_z, h_vn, logit_hx_perp, logit_hz_perp)
        new_llr = self.feedback((h_vn, logit_hz_perp, logit_hx_perp, sx.t(), sz.t()))  # the swap of :436
        llr_hat, x_hat, z_hat = self.decoder((new_llr.permute(0, 2, 1).contiguous(), sx.t(), sz.t()))
        gt_x = (1 - sz).to(torch.float32)  # labels flipped for BCE (:431-432)
        gt_z = (1 - sx).to(torch.float32)
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        loss = torch.zeros((), dtype=torch.float32, device=g.device)
        for i in range(self.loss_from, self.num_iter):  # (:439-442)
            loss = loss + bce(llr_hat[2 * i + 2].t(), gt_x) + bce(llr_hat[2 * i + 3].t(), gt_z)
        s_hat, ls_hat, _ = g.residual(ex, ez, x_hat.to(torch.uint8).contiguous(), z_hat.to(torch.uint8).contiguous())
        return s_hat, ls_hat, loss

    call = __call__

    def value_and_grad(self, noise_x, noise_z, h_vn, logit_hx_perp, logit_hz_perp):
        """``(s_hat, ls_hat, loss, grads)``: what ``with tf.GradientTape() as tape: ... = model(...)`` followed by
        ``tape.gradient(loss, model.trainable_variables)`` yields in Feedback_GNN.ipynb cell 8."""
        g, ex, ez, sx, sz = self._inputs(noise_x, noise_z, h_vn, logit_hx_perp, logit_hz_perp)
        dev = g.device
        llr_in = torch.as_tensor(h_vn, device=dev, dtype=torch.float32).permute(0, 2, 1).contiguous()
        lhx = torch.as_tensor(logit_hz_perp, device=dev, dtype=torch.float32).t().contiguous()  # rows of hx
        lhz = torch.as_tensor(logit_hx_perp, device=dev, dtype=torch.float32).t().contiguous()  # rows of hz
        W = self.feedback.device_weights
        T, factor = self.num_iter, float(self.decoder.normalization_factor)
        new_llr = g.feedback_gnn(W, llr_in, lhx, lhz, sx, sz)
        tr = g.bp4_logit_trace(new_llr, sx, sz, T, factor)
        xl = tr["x_logit"].clone().requires_grad_(True)
        zl = tr["z_logit"].clone().requires_grad_(True)
        gt_x, gt_z = (1 - sz).to(torch.float32), (1 - sx).to(torch.float32)
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        for i in range(self.loss_from, T):
            loss = loss + bce(xl[i + 1], gt_x) + bce(zl[i + 1], gt_z)
        if self.loss_from < T:
            loss.backward()
            gx, gz = xl.grad, zl.grad
        else:
            gx, gz = torch.zeros_like(xl), torch.zeros_like(zl)
        d_new = g.bp4_backward(new_llr, sx, sz, tr["tape_x"], tr["tape_z"], gx, gz, factor)
        grads = g.feedback_gnn_backward(W, llr_in, lhx, lhz, sx, sz, d_new)
        s_hat, ls_hat, _ = g.residual(ex, ez, tr["x_hat"], tr["z_hat"])
        return s_hat, ls_hat, loss.detach(), grads

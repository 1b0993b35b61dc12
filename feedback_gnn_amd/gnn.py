"""GNN_BP4 — the syndrome-only "full GNN" decoder on MI355X (BASELINE.json configs[4]).

Drop-in for `sionna.fec.ldpc.GNN_BP4` (/root/reference sionna/fec/ldpc/gnn.py:71-423).  The configuration of SURVEY.md §8d
(num_embed_dims=20, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True) runs
the MFMA kernel; every other constructor setting the reference's classes accept (embed dims <= 32, hidden units <= 96, 1..4 layers,
sum / mean / max / min, tanh / relu / sigmoid / linear, with or without bias, trainable node / edge attributes) a runtime-shaped kernel.  The reference's `call` raises as shipped (it unpacks five values from
`cal_logit`, which returns four, gnn.py:408 vs :314) and no trained weights exist; this class implements the repaired
semantics of the oracle (oracle/fgnn_oracle.c: og_gnn_bp4) and initialises weights like Keras would (glorot-uniform
kernels, ones biases, zero kernel for `_llr_inv_embed`).
"""
import numpy as np
import torch

from .graph import ACTIVATIONS, REDUCE_OPS, GnnBp4Weights, TannerGraph, gnnbp4_weight_shapes


class MLP:
    """``MLP(units, activations, use_bias)`` — the reference's helper layer (gnn.py:25-69): a chain of Dense layers, built on the first
    call from the input's last dimension (glorot-uniform kernels, ``bias_initializer='ones'``, gnn.py:55-60), ``layer(inputs)`` applies
    it to the last axis.  A convenience for scripts that use the class on its own: it runs as torch ops on the input's device.  The
    decoders do NOT go through it — the MLPs of `Feedback_GNN` and `GNN_BP4` are evaluated inside the HIP kernels from the same weight
    arrays (``get_weights`` / ``set_weights`` use Keras' order: kernel, bias per layer)."""

    _ACT = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, "linear": None, None: None}

    def __init__(self, units, activations, use_bias, seed=0):
        if not (len(units) == len(activations) == len(use_bias)):
            raise ValueError("units, activations and use_bias must have one entry per layer")
        for a in activations:
            if a not in self._ACT:
                raise NotImplementedError(f"activation {a!r}: tanh, relu, sigmoid and linear are implemented")
        self._num_units, self._activations, self._use_bias = [int(u) for u in units], list(activations), [bool(b) for b in use_bias]
        self._seed, self._weights = seed, None

    def build(self, input_shape):
        rng = np.random.RandomState(self._seed)
        fan_in, w = int(input_shape[-1]), []
        for units, bias in zip(self._num_units, self._use_bias):
            lim = np.sqrt(6.0 / (fan_in + units))
            w.append(rng.uniform(-lim, lim, size=(fan_in, units)).astype(np.float32))
            if bias:
                w.append(np.ones((units,), np.float32))
            fan_in = units
        self._weights = w

    def get_weights(self):
        return [a.copy() for a in (self._weights or [])]

    def set_weights(self, weights):
        self._weights = [np.asarray(a, dtype=np.float32) for a in weights]

    def __call__(self, inputs):
        x = torch.as_tensor(inputs, dtype=torch.float32)
        if self._weights is None:
            self.build(tuple(x.shape))
        it = iter(torch.from_numpy(a).to(x.device) for a in self._weights)
        for act, bias in zip(self._activations, self._use_bias):
            x = x @ next(it)
            if bias:
                x = x + next(it)
            if self._ACT[act] is not None:
                x = self._ACT[act](x)
        return x

    call = __call__


class GNN_BP4:
    """``decoder((syndrome_x[bs,m_x], syndrome_z[bs,m_z]))`` → ``(llr_hat, x_hat[n,bs], z_hat[n,bs])`` where ``llr_hat`` is a
    list with one ``(x_perp_logit[m_z+k,bs], z_perp_logit[m_x+k,bs])`` pair per iteration (gnn.py:409)."""

    def __init__(self, code, num_embed_dims, num_msg_dims, num_hidden_units, num_mlp_layers, num_iter, reduce_op="mean",
                 activation="tanh", clip_llr_to=None, use_attributes=False, node_attribute_dims=0, msg_attribute_dims=0,
                 use_bias=False, input_embed=False, loss_type="boxplus-phi", device=None, graph=None, seed=0):
        if loss_type != "boxplus-phi":
            # 'sine' (gnn.py:410-412) leaves hx_logit / hz_logit undefined for the next check update: the reference's call raises
            raise NotImplementedError("loss_type 'boxplus-phi' only (the reference's 'sine' branch cannot run more than one iteration)")
        self.config = (int(num_embed_dims), int(num_hidden_units), int(num_mlp_layers), reduce_op, activation, bool(use_bias),
                       bool(use_attributes), int(node_attribute_dims) if use_attributes else 0,
                       int(msg_attribute_dims) if use_attributes else 0)
        if reduce_op not in REDUCE_OPS:
            raise ValueError("unknown reduce operation")  # gnn.py:568
        if activation not in ACTIVATIONS:
            raise NotImplementedError(f"activation {activation!r}: the HIP kernels implement {sorted(k for k in ACTIVATIONS if k)}")
        # num_msg_dims is accepted but irrelevant: the reference overwrites units[-1] with num_embed_dims in the list the message
        # MLPs share (gnn.py:548, :690), so messages have num_embed_dims components.  clip_llr_to and input_embed are stored by the
        # reference and never read by call.
        self._num_msg_dims = int(num_msg_dims)
        self._clip_llr_to, self._input_embed = clip_llr_to, input_embed
        self._num_iter = int(num_iter)
        self.graph = graph if graph is not None else TannerGraph(code, stage_one=True, device=device)
        rng = np.random.RandomState(seed)
        shapes = gnnbp4_weight_shapes(self.graph, self.config)
        nw = (7 * self.config[2] + 1) * (2 if use_bias else 1)  # Dense arrays; the rest are attributes
        w = []
        for i, shp in enumerate(shapes):
            if i >= nw:
                w.append(np.zeros(shp, np.float32))  # trainable attributes start at zero (gnn.py:527-532, :673-677)
            elif len(shp) == 1:
                w.append(np.ones(shp, np.float32))   # bias_initializer='ones' (gnn.py:47-50, :248)
            else:
                lim = np.sqrt(6.0 / (shp[0] + shp[1]))
                w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
        w[nw - (2 if use_bias else 1)] = np.zeros(shapes[nw - (2 if use_bias else 1)], np.float32)  # _llr_inv_embed kernel: zeros (:249)
        self._weights = None
        self.set_weights(w)

    @property
    def num_iter(self):
        return self._num_iter

    @num_iter.setter
    def num_iter(self, value):
        self._num_iter = int(value)

    def get_weights(self):
        return [a.copy() for a in self._weights.arrays]

    def set_weights(self, weights):
        self._weights = GnnBp4Weights(list(weights), self.graph.device, config=self.config, graph=self.graph)

    def __call__(self, inputs):
        syndrome_x, syndrome_z = inputs
        g = self.graph

        def prep(s, rows):
            s = torch.as_tensor(s, device=g.device)
            if s.dim() != 2 or s.shape[1] != rows:
                raise ValueError(f"syndrome must have shape [batch_size, {rows}], got {tuple(s.shape)}")
            return (s.to(torch.int64) & 1).to(torch.uint8).contiguous()

        out = g.gnn_bp4_decode(self._weights, prep(syndrome_x, g.m_x), prep(syndrome_z, g.m_z), self._num_iter)
        llr_hat = [(out["x_logit_all"][i].t(), out["z_logit_all"][i].t()) for i in range(self._num_iter)]
        return llr_hat, out["x_hat"].t().to(torch.int64), out["z_hat"].t().to(torch.float64)

    call = __call__

    def build(self, input_shape=None):
        """Keras builds lazily (gnn.py:296-312); here the weights exist after construction."""

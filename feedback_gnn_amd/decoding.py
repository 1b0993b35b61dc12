"""LDPCBPDecoder (binary, syndrome mode) and BP_BSC_Model on MI355X — SURVEY.md §8(f) rank 1.

Drop-ins for the QLDPC use of `sionna.fec.ldpc.LDPCBPDecoder` (/root/reference sionna/fec/ldpc/decoding.py:15-1048 with the
fork's `is_syndrome` and `normalization_factor` additions) and `BP_BSC_Model` (sionna/fec/ldpc/feedback_gnn.py:190-229), as
used in examples/QLDPC.ipynb cell 7 (binary decoding of the [[882,24]] hx matrix over a BSC).
"""
import types

import numpy as np
import torch

from ._lib import CN_TYPES
from .graph import TannerGraph


def _binary_graph(pcm, logical_pcm, device):
    pcm = np.asarray(pcm).astype(np.int64)
    n = pcm.shape[1]
    lp = np.zeros((1, n), np.int64) if logical_pcm is None else np.asarray(logical_pcm).astype(np.int64)
    zero = np.zeros((1, n), np.int64)
    code = types.SimpleNamespace(hx=pcm, hz=pcm, hx_perp=lp, hz_perp=lp, lx=zero, lz=zero)
    return TannerGraph(code, stage_one=True, device=device)


class LDPCBPDecoder:
    """Flooding BP on one parity-check matrix with syndrome input.

    ``LDPCBPDecoder(pcm, trainable=False, cn_type='boxplus-phi', hard_out=True, track_exit=False, num_iter=32,
    normalization_factor=1.0, stateful=False, is_syndrome=False, output_dtype=torch.float32)`` (decoding.py:260-270).
    Call ``decoder((llr_ch[bs,n], syndrome[m,bs]))`` if ``is_syndrome`` else ``decoder(llr_ch)``; ``llr_ch`` are logits
    (log p(1)/p(0)); returns hard decisions (0/1 floats) or soft logits ``[bs,n]``.  Trainable weights, EXIT tracking and the
    stateful mode of upstream Sionna are outside the QLDPC path and raise NotImplementedError.
    """

    def __init__(self, pcm, trainable=False, cn_type='boxplus-phi', hard_out=True, track_exit=False, num_iter=32,
                 normalization_factor=1.0, stateful=False, is_syndrome=False, output_dtype=torch.float32, device=None, graph=None,
                 **kwargs):
        if cn_type not in CN_TYPES:
            raise ValueError('Unknown node type.')
        if trainable or track_exit or stateful:
            raise NotImplementedError("trainable / track_exit / stateful are not part of the syndrome-decoding path")
        if not isinstance(num_iter, (int, np.integer)) or num_iter < 0:
            raise AssertionError('num_iter cannot be negative.')
        pcm = np.asarray(pcm.toarray() if hasattr(pcm, "toarray") else pcm)
        if not np.array_equal(pcm, pcm.astype(bool)):
            raise AssertionError('PC matrix must be binary.')
        self._cn_type, self._hard_out, self._num_iter = cn_type, bool(hard_out), int(num_iter)
        self._normalization_factor, self._is_syndrome, self._output_dtype = float(normalization_factor), bool(is_syndrome), output_dtype
        self.graph = graph if graph is not None else _binary_graph(pcm, None, device)
        self._num_vns, self._num_cns = self.graph.n, self.graph.m_x
        self._pcm = pcm

    # ---- the reference class' read-only surface (decoding.py:420-494) ----
    pcm = property(lambda self: self._pcm)
    num_cns = property(lambda self: self._num_cns)
    num_vns = property(lambda self: self._num_vns)
    num_edges = property(lambda self: int(self.graph.E_x))
    has_weights = property(lambda self: False)     # trainable=True is refused by the constructor
    output_dtype = property(lambda self: self._output_dtype)
    llr_max = property(lambda self: 20.0)          # the input clip of decoding.py:918-920, fixed in the kernel (fgnn_bp2_decode)

    @property
    def num_iter(self):
        return self._num_iter

    @num_iter.setter
    def num_iter(self, value):
        if not isinstance(value, (int, np.integer)) or value < 0:
            raise AssertionError('num_iter cannot be negative.')
        self._num_iter = int(value)

    @property
    def edge_weights(self):
        raise NotImplementedError("no trainable edge weights on the syndrome-decoding path (has_weights is False)")

    @property
    def ie_c(self):
        raise NotImplementedError("EXIT tracking (track_exit) is not part of the syndrome-decoding path")

    ie_v = ie_c

    def show_weights(self, size=7):
        raise NotImplementedError("no trainable edge weights on the syndrome-decoding path")

    def build(self, input_shape=None):
        """Keras builds lazily; this class is ready after construction."""

    def __call__(self, inputs):
        g = self.graph
        if self._is_syndrome:
            llr_ch, syndrome = inputs
            syndrome = torch.as_tensor(syndrome, device=g.device)
            if syndrome.dim() != 2 or syndrome.shape[0] != self._num_cns:
                raise ValueError(f"syndrome must have shape [{self._num_cns}, batch_size]")
            synd = (syndrome.to(torch.int64) & 1).to(torch.uint8).t().contiguous()
        else:
            llr_ch, synd = inputs, None
        llr_ch = torch.as_tensor(llr_ch, device=g.device)
        if llr_ch.dtype != self._output_dtype:
            raise TypeError('Invalid input dtype.')
        if llr_ch.shape[-1] != self._num_vns:
            raise ValueError('Last dimension must be of length n.')
        shape = llr_ch.shape
        flat = llr_ch.reshape(-1, self._num_vns).to(torch.float32).contiguous()
        soft, hard = g.bp2_decode(synd, self._num_iter, self._cn_type, self._normalization_factor, llr_ch=flat,
                                  want_soft=not self._hard_out, want_hard=self._hard_out)
        out = hard.to(self._output_dtype) if self._hard_out else soft.to(self._output_dtype)
        return out.reshape(shape)

    call = __call__


class BP_BSC_Model:
    """``BP_BSC_Model(pcm, decoder, logical_pcm=None, p0=None)``; ``model(batch_size, p)`` draws BSC(p) noise on the all-zero
    word, decodes its syndrome and returns ``(noise, noise_hat)`` or, with ``logical_pcm``, ``(s_hat[bs,m], ls_hat[bs,rows])``
    (feedback_gnn.py:207-229)."""

    def __init__(self, pcm, decoder, logical_pcm=None, p0=None, *, seed=0x5EED, rank=0, world_size=1):
        self.pcm, self.logical_pcm, self.decoder, self.p0 = pcm, logical_pcm, decoder, p0
        self.n = np.asarray(pcm).shape[1]
        self.graph = _binary_graph(pcm, logical_pcm, decoder.graph.device)
        self.seed, self.rank, self.world_size, self._next = int(seed), int(rank), int(world_size), 0

    def __call__(self, batch_size, ebno_db=None, **kw):
        p = float(kw.get("p", ebno_db))
        B = int(batch_size)
        g = self.graph
        first = self._next + self.rank * B
        self._next += self.world_size * B
        p0 = np.float32(p if self.p0 is None else self.p0)
        llr_const = float(-np.log((np.float32(1.0) - p0) / p0, dtype=np.float32))  # (:210-211)
        noise = g.bsc_noise(self.seed, p, first, B)
        zeros = torch.zeros_like(noise)
        synd, _ = g.syndrome(zeros, noise)  # syndrome_x = hx * noise_z with hx = pcm (:216-217)
        d = self.decoder
        _, noise_hat = g.bp2_decode(synd, d._num_iter, d._cn_type, d._normalization_factor, llr_const=llr_const, B=B,
                                    want_soft=False)
        if self.logical_pcm is None:
            return noise.to(torch.float32), noise_hat.to(torch.float32)
        # s_hat = pcm (noise xor noise_hat), ls_hat = logical_pcm (noise xor noise_hat) (:222-229): the x-halves of the
        # CSS residual with hz = pcm, hx_perp = logical_pcm
        s_hat, ls_hat, _ = g.residual(noise, zeros, noise_hat, zeros, want_arrays=True)
        return s_hat[:, :g.m_z], ls_hat[:, :g.rows_hxp]

    call = __call__

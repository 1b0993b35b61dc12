"""QLDPCBPDecoder — syndrome-based quaternary BP for CSS codes on MI355X.

Drop-in for `sionna.fec.ldpc.QLDPCBPDecoder` (/root/reference sionna/fec/ldpc/decoding_q.py:14-797):
same constructor arguments, same call contract and return tuples, torch tensors instead of TF
tensors.  The work is done by `fgnn_bp4_decode` in libfgnn_hip.so (LDS-resident HIP kernel);
this class only validates inputs and converts between the reference's tensor shapes and the
library's codeword-major layouts.
"""
import numpy as np
import torch

from ._lib import CN_TYPES
from .graph import TannerGraph


class QLDPCBPDecoder:
    """Iterative BP4 decoder.

    Parameters follow decoding_q.py:15-27: ``code`` (css_code), ``trainable``, ``cn_type`` in
    {'boxplus', 'boxplus-phi', 'minsum'}, ``hard_out``/``track_exit``/``loss_type`` (stored, unused —
    as in the reference), ``num_iter``, ``normalization_factor``, ``output_dtype``, ``stage_one``,
    ``stage_two``.  Extra keywords: ``device``; ``graph`` (share another decoder's TannerGraph); ``reference_dtypes`` (default True) returns
    ``x_hat`` as int64 and ``z_hat`` as float64 exactly like decoding_q.py:788-790, False keeps uint8.

    Call: ``decoder((llr_ch[bs,3,n] float32, syndrome_x[m_x,bs], syndrome_z[m_z,bs]))`` →
      stage_one:              (llrx, llry, llrz [bs,n], x_hat, z_hat [bs,n], x_logit, z_logit [rows,bs])
      trainable or stage_two: (llr_hat[2*num_iter+2, rows, bs], x_hat, z_hat)
      otherwise:              (x_hat, z_hat)
    """

    def __init__(self, code, trainable=False, cn_type='boxplus', hard_out=True, track_exit=False, num_iter=32,
                 normalization_factor=0.625, output_dtype=torch.float32, loss_type='boxplus-phi', stage_one=False,
                 stage_two=False, device=None, reference_dtypes=True, graph=None, **kwargs):
        if cn_type not in CN_TYPES:
            raise ValueError('Unknown node type.')  # decoding_q.py:107
        self._trainable = bool(trainable)
        self._cn_type = cn_type
        self._hard_out = hard_out
        self._track_exit = track_exit
        self._num_iter = int(num_iter)
        self._normalization_factor = float(normalization_factor)
        self._output_dtype = output_dtype
        self._loss_type = loss_type
        self._stage_one = bool(stage_one)
        self._stage_two = bool(stage_two)
        self._reference_dtypes = bool(reference_dtypes)
        self._code = code
        # pcm_x_perp/pcm_z_perp := hz/hx for the two-stage modes (decoding_q.py:35-37)
        want_stage = self._stage_one or self._stage_two
        if graph is not None and graph.stage_one != want_stage:
            raise ValueError("shared graph was built for a different stage_one/stage_two setting")
        # decoders of one sandwich can share one device graph (``graph=``): the tables are immutable
        self.graph = graph if graph is not None else TannerGraph(code, stage_one=want_stage, device=device)
        self._num_vns = self.graph.n
        self._num_cns_x = self.graph.m_x
        self._num_cns_z = self.graph.m_z
        self._num_edges_x = self.graph.E_x
        self._num_edges_z = self.graph.E_z

    # properties used by the sandwich driver
    @property
    def num_iter(self):
        return self._num_iter

    @property
    def cn_type(self):
        return self._cn_type

    @property
    def normalization_factor(self):
        return self._normalization_factor

    @property
    def dtype(self):
        return self._output_dtype

    def build(self, input_shape=None):
        """Keras builds lazily (decoding_q.py:646-659 only validates shapes); this class is ready after construction."""

    def show_weights(self, size=7):
        raise NotImplementedError("no trainable edge weights: the reference's QLDPCBPDecoder never applies them either (SURVEY.md a10)")

    def cal_logit(self, llrx, llry, llrz):
        """Soft syndromes of given marginals (decoding_q.py:455-471): ``llrx, llry, llrz [n, bs]`` (the layout `call` holds them in)
        → ``(x_perp_logit[rows_x, bs], z_perp_logit[rows_z, bs])``.  Evaluated by the decoder kernel's own epilogue: a zero-iteration
        decode whose channel LLRs are the given marginals returns them unchanged together with their soft syndromes.  The epilogue
        (`_cn_update_phi_loss`, :433-453) is the same code for every check-node rule; the call names 'minsum' so that it always runs on
        the shared float32 routines — bit-equal to the oracle's restatement — even when the opt-in FGNN_OPT_HW_TRANSCENDENTALS is on
        (that option only ever applies to 'boxplus-phi' launches) and whatever this decoder's own cn_type / normalization factor are."""
        g = self.graph
        llr = torch.stack([torch.as_tensor(t, device=g.device).to(torch.float32) for t in (llrx, llry, llrz)], dim=0)  # [3, n, bs]
        if llr.dim() != 3 or llr.shape[1] != self._num_vns:
            raise ValueError('llrx, llry, llrz must have shape [n, batch_size].')
        llr = llr.permute(2, 0, 1).contiguous()  # [bs, 3, n]
        B = llr.shape[0]
        sx = torch.zeros((B, self._num_cns_x), dtype=torch.uint8, device=g.device)
        sz = torch.zeros((B, self._num_cns_z), dtype=torch.uint8, device=g.device)
        out = g.bp4_decode(sx, sz, 0, "minsum", 1.0, llr_ch=llr, want_logits=True)
        return out["x_logit"].t(), out["z_logit"].t()

    def _syndrome_in(self, s, rows):
        s = torch.as_tensor(s, device=self.graph.device)
        if s.dim() != 2 or s.shape[0] != rows:
            raise ValueError(f"syndrome must have shape [{rows}, batch_size], got {tuple(s.shape)}")
        return (s.to(torch.int64) & 1).to(torch.uint8).t().contiguous()

    def _hard_out_dtypes(self, x_hat, z_hat):
        if self._reference_dtypes:
            return x_hat.to(torch.int64), z_hat.to(torch.float64)
        return x_hat, z_hat

    def __call__(self, inputs):
        llr_ch, syndrome_x, syndrome_z = inputs
        llr_ch = torch.as_tensor(llr_ch, device=self.graph.device)
        if llr_ch.dtype != self._output_dtype:
            raise TypeError('Invalid input dtype.')  # tf.debugging.assert_type, decoding_q.py:690
        if llr_ch.shape[-1] != self._num_vns:
            raise ValueError('Last dimension must be of length n.')  # decoding_q.py:701-703
        if llr_ch.dim() != 3 or llr_ch.shape[1] != 3:
            raise ValueError('llr_ch must have shape [batch_size, 3, n].')
        llr_ch = llr_ch.to(torch.float32).contiguous()  # internal calculations in float32 (:693)
        sx = self._syndrome_in(syndrome_x, self._num_cns_x)
        sz = self._syndrome_in(syndrome_z, self._num_cns_z)
        if sx.shape[0] != llr_ch.shape[0] or sz.shape[0] != llr_ch.shape[0]:
            raise ValueError('batch sizes of llr_ch and the syndromes differ.')
        g = self.graph
        if self._trainable or self._stage_two:
            return self._call_with_logit_trace(llr_ch, sx, sz)
        out = g.bp4_decode(sx, sz, self._num_iter, self._cn_type, self._normalization_factor, llr_ch=llr_ch,
                           want_logits=self._stage_one)
        x_hat, z_hat = self._hard_out_dtypes(out["x_hat"], out["z_hat"])
        if self._stage_one:
            llr = out["llr"]
            return (llr[:, 0, :], llr[:, 1, :], llr[:, 2, :], x_hat, z_hat, out["x_logit"].t(), out["z_logit"].t())
        return x_hat, z_hat

    def _call_with_logit_trace(self, llr_ch, sx, sz):
        """trainable / stage_two return mode (decoding_q.py:730,743-746,779-781,794-795): the soft syndromes recorded after the VN
        update of every iteration are the soft syndromes of the marginals after 0, 1, ..., num_iter full iterations; the kernel
        records them itself in one launch (fgnn_bp4_decode_trace) and ``llr_hat[2k]`` / ``llr_hat[2k+1]`` are transposed views of
        its two trace buffers."""
        g = self.graph
        rows_x, rows_z = g.rows_xp, g.rows_zp
        if rows_x != rows_z:
            raise ValueError("llr_hat stacking needs pcm_x_perp and pcm_z_perp with equal row counts")
        B = llr_ch.shape[0]
        out = g.bp4_decode_trace(sx, sz, self._num_iter, self._cn_type, self._normalization_factor, llr_ch=llr_ch)
        hat = torch.stack((out["x_logit"], out["z_logit"]), dim=1).reshape(2 * self._num_iter + 2, B, rows_x).transpose(1, 2)
        x_hat, z_hat = self._hard_out_dtypes(out["x_hat"], out["z_hat"])
        return hat, x_hat, z_hat

    call = __call__

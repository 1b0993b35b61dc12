"""Device-resident Tanner graphs of a CSS code and thin torch-tensor wrappers over the C ABI.

`TannerGraph` owns one `fgnn_graph` handle (include/fgnn.h).  PyTorch-ROCm is used only for
device memory and streams: every method takes/returns torch tensors that live on the graph's
device, passes their `data_ptr()` and the current HIP stream to the library, and never touches
the data on the host.  Layouts are the library's codeword-major ones (fgnn.h "Conventions").
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import CN_TYPES, check


def _coo(mat):
    r, c = np.nonzero(np.asarray(mat))
    return np.ascontiguousarray(r, dtype=np.int32), np.ascontiguousarray(c, dtype=np.int32)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _resolve_device(device):
    """torch.device with an explicit index: None and a bare "cuda" mean torch's CURRENT device (not device 0)."""
    device = torch.device("cuda" if device is None else device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


REDUCE_OPS = {"sum": 0, "mean": 1, "max": 2, "min": 3}
ACTIVATIONS = {None: 0, "linear": 0, "tanh": 1, "relu": 2, "sigmoid": 3}
SHIPPED_GNN_CONFIG = (20, 40, 2, "mean", "tanh", True)


def gnn_weight_shapes(num_msg_dims, num_hidden_units, num_mlp_layers, use_bias=True):
    """Shapes of Feedback_GNN.get_weights() (feedback_gnn.py:110-128): _llr_inv_embed, vn_msg_mlp_x, vn_msg_mlp_z, vn_embed_mlp."""
    D, H, L = int(num_msg_dims), int(num_hidden_units), int(num_mlp_layers)
    dense = [(H if L > 1 else 2 * D + 3, 3)]
    for _ in range(2):
        dense += [(4 if l == 0 else H, D if l == L - 1 else H) for l in range(L)]
    dense += [(2 * D + 3 if l == 0 else H, H) for l in range(L - 1)]
    shapes = []
    for k, j in dense:
        shapes.append((k, j))
        if use_bias:
            shapes.append((j,))
    return shapes


class GnnWeights:
    """Device copy of one feedback GNN's weight arrays (fgnn_weights).  ``config`` = (num_msg_dims, num_hidden_units,
    num_mlp_layers, reduce_op, activation, use_bias); the shipped configuration takes the MFMA kernel, any other one the
    runtime-shaped kernel (fgnn_weights_create_general)."""

    def __init__(self, arrays, device, config=SHIPPED_GNN_CONFIG, force_general=False):
        arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        D, H, L, red, act, bias = config
        if red not in REDUCE_OPS:
            raise ValueError("unknown reduce operation")  # feedback_gnn.py:148
        if act not in ACTIVATIONS:
            raise NotImplementedError(f"activation {act!r}: the HIP kernels implement {sorted(k for k in ACTIVATIONS if k)}")
        shapes = gnn_weight_shapes(D, H, L, bias)
        if [a.shape for a in arrays] != shapes:
            raise ValueError(f"feedback-GNN weights must have shapes {shapes}, got {[a.shape for a in arrays]}")
        self.arrays = arrays
        self.config = (int(D), int(H), int(L), red, act, bool(bias))
        self.device = _resolve_device(device)
        self.general = force_general or self.config != SHIPPED_GNN_CONFIG
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        h = C.c_void_p()
        if self.general:
            cfg = (C.c_int * 6)(int(D), int(H), int(L), REDUCE_OPS[red], ACTIVATIONS[act], int(bool(bias)))
            check(_lib.lib().fgnn_weights_create_general(cfg, ptrs, len(arrays), self.device.index, C.byref(h)))
        else:
            check(_lib.lib().fgnn_weights_create(ptrs, self.device.index, C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().fgnn_weights_destroy(self.handle)
        except Exception:
            pass


class TannerGraph:
    """hx/hz Tanner graphs + soft-syndrome row sets + hx_perp/hz_perp of one CSS code on one GPU.

    stage_one=True installs pcm_x_perp = hz, pcm_z_perp = hx as the soft-syndrome rows
    (reference decoding_q.py:35-37), otherwise code.hx_perp / code.hz_perp (:33-34).
    """

    def __init__(self, code, stage_one=True, device=None):
        if device is not None and torch.device(device).type != "cuda":
            raise _lib.FgnnError("the BP4/feedback-GNN decoder runs on a HIP device only (no CPU fallback)")
        if not torch.cuda.is_available():
            raise _lib.FgnnError("no HIP device: the BP4/feedback-GNN decoder has no CPU fallback")
        self.device = _resolve_device(device)
        L = _lib.lib()
        self.code = code
        hx, hz = np.asarray(code.hx), np.asarray(code.hz)
        self.n = int(hx.shape[1])
        self.m_x, self.m_z = int(hx.shape[0]), int(hz.shape[0])
        rx, cx = _coo(hx)
        rz, cz = _coo(hz)
        self.E_x, self.E_z = len(rx), len(rz)
        h = C.c_void_p()
        check(L.fgnn_graph_create(self.n, self.m_x, self.m_z, self.E_x, _np_ptr(rx), _np_ptr(cx), self.E_z, _np_ptr(rz),
                                  _np_ptr(cz), self.device.index, C.byref(h)))
        self.handle = h
        self.gnn_factored = False  # the library default: the reference's association, one Dense per edge (FGNN_OPT_GNN_FACTORED is opt-in)
        self.gnn_stream = True  # the library default (FGNN_OPT_GNN_STREAM)
        self.bp4_shared_lse = False  # the library default: one log-sum-exp per edge (FGNN_OPT_BP4_SHARED_LSE is opt-in)
        self.stage_one = bool(stage_one)
        xp, zp = (hz, hx) if stage_one else (np.asarray(code.hx_perp), np.asarray(code.hz_perp))
        self.rows_xp, self.rows_zp = int(xp.shape[0]), int(zp.shape[0])
        self.rows_hxp, self.rows_hzp = int(code.hx_perp.shape[0]), int(code.hz_perp.shape[0])
        self.rows_lx, self.rows_lz = int(np.asarray(code.lx).shape[0]), int(np.asarray(code.lz).shape[0])
        for which, mat in ((0, xp), (1, zp), (2, code.hx_perp), (3, code.hz_perp), (4, code.lx), (5, code.lz)):
            r, c = _coo(mat)
            check(L.fgnn_graph_set_rows(self.handle, which, int(np.asarray(mat).shape[0]), len(r), _np_ptr(r), _np_ptr(c)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().fgnn_graph_destroy(self.handle)
        except Exception:
            pass

    # ---- introspection ------------------------------------------------------------------------
    def info(self):
        buf = (C.c_int32 * 16)()
        check(_lib.lib().fgnn_graph_info(self.handle, buf))
        keys = ("n", "m_x", "m_z", "E_x", "E_z", "threads_per_codeword", "codewords_per_block", "lds_bytes_per_block",
                "regular", "device", "dv_x", "dv_z", "dc")
        return dict(zip(keys, list(buf)))

    def set_launch(self, threads_per_codeword=0, codewords_per_block=0):
        check(_lib.lib().fgnn_graph_set_launch(self.handle, int(threads_per_codeword), int(codewords_per_block)))

    def set_saturation_shortcut(self, on=True):
        """Exact wave-uniform shortcut for saturated nodes in the regular BP4 kernel (default on; same results)."""
        check(_lib.lib().fgnn_graph_set_option(self.handle, 1, int(bool(on))))

    def set_fixed_point_exit(self, on=True):
        """Exact early exit of converged codewords (FGNN_OPT_FIXED_POINT_EXIT; only acts with the saturation shortcut on)."""
        check(_lib.lib().fgnn_graph_set_option(self.handle, 2, int(bool(on))))

    def set_hw_transcendentals(self, on=True):
        """OPT-IN, not bit-exact (FGNN_OPT_HW_TRANSCENDENTALS): boxplus-phi decodes run on v_exp_f32 / v_log_f32 in the fixed
        dataflow, phi's clip points pinned to the reference's known-answer values (same saturated fixed point as the exact kernel,
        different bits in the transient).  Off by default; no parity test and no headline number uses it."""
        check(_lib.lib().fgnn_graph_set_option(self.handle, 3, int(bool(on))))

    def set_gnn_factored(self, on=True):
        """OPT-IN re-association (FGNN_OPT_GNN_FACTORED, default off = feedback_gnn.py:175-184 term by term): the X/Y/Z part of the first
        Dense once per qubit and side, one last Dense on the edge-summed activations.  Same real-number function, float32 rounding differs
        (<= 5e-7 on the output): statistically the same decoder, not the reference's operation sequence."""
        check(_lib.lib().fgnn_graph_set_option(self.handle, 4, int(bool(on))))
        self.gnn_factored = bool(on)

    def set_gnn_stream(self, on=True):
        """Feedback GNN of a regular graph on the streaming VALU kernel (FGNN_OPT_GNN_STREAM): True (default) = wherever it is the
        faster kernel (launches of 4 096 codewords or more), "always" = every launch, False = never
        (MFMA-tile kernel).  The same float operations in the same order: bit-identical results.  "always" also moves `gnn_bp4_decode`
        on a (3,3,6)-regular graph to its streaming kernel, which is bit-identical and ~35 % SLOWER than its MFMA tiles (the tested
        second implementation, include/fgnn.h option 6); True / False leave GNN_BP4 on the MFMA tiles."""
        value = 2 if on == "always" else int(bool(on))
        check(_lib.lib().fgnn_graph_set_option(self.handle, 6, value))
        self.gnn_stream = "always" if value == 2 else bool(value)

    def set_bp4_shared_lse(self, on=True):
        """OPT-IN re-association (FGNN_OPT_BP4_SHARED_LSE, default off = decoding_q.py:254-273 term by term): the (a - b)-dependent part
        of the qubit update's log-sum-exp formed once per qubit and side instead of once per edge — same real-number function, 4 instead
        of 8 exp/log pairs per qubit and iteration, v->c messages move by <= 4 ulp of the totals per update: statistically the same
        decoder, not the reference's operation sequence."""
        check(_lib.lib().fgnn_graph_set_option(self.handle, 5, int(bool(on))))
        self.bp4_shared_lse = bool(on)

    def force_generic(self, on=True):
        """Testing hook: run the runtime-degree kernel even on a degree-regular graph."""
        check(_lib.lib().fgnn_graph_force_generic(self.handle, int(bool(on))))

    def profile_enable(self, max_launches):
        """Record HIP events around every BP4 launch (0 disables)."""
        check(_lib.lib().fgnn_profile_enable(self.handle, int(max_launches)))
        self._prof_cap = int(max_launches)

    def profile_read(self):
        """[(ms, num_iter, batch), ...] of the BP4 launches since the last read."""
        cap = max(1, getattr(self, "_prof_cap", 0))
        ms = np.zeros(cap, np.float32)
        it = np.zeros(cap, np.int32)
        bt = np.zeros(cap, np.int32)
        cnt = C.c_int32(0)
        check(_lib.lib().fgnn_profile_read(self.handle, _np_ptr(ms), _np_ptr(it), _np_ptr(bt), cap, C.byref(cnt)))
        return [(float(ms[i]), int(it[i]), int(bt[i])) for i in range(cnt.value)]

    def edges(self, side):
        """Canonical (qubit, check)-sorted edge list of hx (side 0) or hz (side 1): (chk, var)."""
        E = self.E_x if side == 0 else self.E_z
        chk = np.empty(E, np.int32)
        var = np.empty(E, np.int32)
        check(_lib.lib().fgnn_graph_edges(self.handle, side, _np_ptr(chk), _np_ptr(var)))
        return chk, var

    # ---- helpers ----------------------------------------------------------------------------------
    def _chk(self, t, shape, dtype, name):
        if t.device != self.device:
            raise ValueError(f"{name} must live on {self.device}, got {t.device}")
        if t.dtype != dtype:
            raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        return t.contiguous()

    def _chk_out(self, t, shape, dtype, name):
        """A buffer the library writes in place: validated like an input, but a non-contiguous view is an error (a silent copy
        would receive the result instead of the caller's tensor)."""
        self._chk(t, shape, dtype, name)
        if not t.is_contiguous():
            raise ValueError(f"{name} is written in place and must be contiguous")
        return t

    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ---- QLDPCBPDecoder.call ---------------------------------------------------------------------
    def bp4_decode(self, synd_x, synd_z, num_iter, cn_type="boxplus-phi", factor=1.0, llr_ch=None, llr_const=0.0,
                   msg_init=None, return_msgs=False, want_logits=True):
        if cn_type not in CN_TYPES:
            raise ValueError("Unknown node type.")
        B = int(synd_x.shape[0])
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        if llr_ch is not None:
            llr_ch = self._chk(llr_ch, (B, 3, self.n), torch.float32, "llr_ch")
        mix = miz = None
        if msg_init is not None:
            mix = self._chk(msg_init[0], (B, self.E_x), torch.float32, "msg_init_x")
            miz = self._chk(msg_init[1], (B, self.E_z), torch.float32, "msg_init_z")
        llr = self._new((B, 3, self.n), torch.float32)
        xh = self._new((B, self.n), torch.uint8)
        zh = self._new((B, self.n), torch.uint8)
        xl = self._new((B, self.rows_xp), torch.float32) if want_logits else None
        zl = self._new((B, self.rows_zp), torch.float32) if want_logits else None
        mox = self._new((B, self.E_x), torch.float32) if return_msgs else None
        moz = self._new((B, self.E_z), torch.float32) if return_msgs else None
        check(_lib.lib().fgnn_bp4_decode(self.handle, CN_TYPES[cn_type], int(num_iter), float(factor), _ptr(llr_ch),
                                         float(llr_const), _ptr(synd_x), _ptr(synd_z), B, _ptr(mix), _ptr(miz), _ptr(llr),
                                         _ptr(xh), _ptr(zh), _ptr(xl), _ptr(zl), _ptr(mox), _ptr(moz), _stream(self.device)))
        out = dict(llr=llr, x_hat=xh, z_hat=zh, x_logit=xl, z_logit=zl)
        if return_msgs:
            out["msg_x"], out["msg_z"] = mox, moz
        return out

    def bp4_decode_trace(self, synd_x, synd_z, num_iter, cn_type="boxplus-phi", factor=1.0, llr_ch=None, llr_const=0.0,
                         msg_init=None, want_tape=False):
        """One launch of `num_iter` iterations that records the soft syndromes after 0, 1, ..., num_iter iterations
        (fgnn_bp4_decode_trace: the reference's trainable / stage_two return mode, decoding_q.py:743-746).  Returns
        dict(llr, x_hat, z_hat, x_logit [T+1,B,rows0], z_logit [T+1,B,rows1]) and, with ``want_tape``, tape_x [T+1,B,E_x] /
        tape_z [T+1,B,E_z] (the c->v messages before every iteration, the last slot after the last one)."""
        if cn_type not in CN_TYPES:
            raise ValueError("Unknown node type.")
        B, T = int(synd_x.shape[0]), int(num_iter)
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        if llr_ch is not None:
            llr_ch = self._chk(llr_ch, (B, 3, self.n), torch.float32, "llr_ch")
        mix = miz = None
        if msg_init is not None:
            mix = self._chk(msg_init[0], (B, self.E_x), torch.float32, "msg_init_x")
            miz = self._chk(msg_init[1], (B, self.E_z), torch.float32, "msg_init_z")
        llr = self._new((B, 3, self.n), torch.float32)
        xh = self._new((B, self.n), torch.uint8)
        zh = self._new((B, self.n), torch.uint8)
        xl = self._new((T + 1, B, self.rows_xp), torch.float32)
        zl = self._new((T + 1, B, self.rows_zp), torch.float32)
        tx = self._new((T + 1, B, self.E_x), torch.float32) if want_tape else None
        tz = self._new((T + 1, B, self.E_z), torch.float32) if want_tape else None
        check(_lib.lib().fgnn_bp4_decode_trace(self.handle, CN_TYPES[cn_type], T, float(factor), _ptr(llr_ch), float(llr_const),
                                               _ptr(synd_x), _ptr(synd_z), B, _ptr(mix), _ptr(miz), _ptr(llr), _ptr(xh), _ptr(zh),
                                               _ptr(xl), _ptr(zl), _ptr(tx), _ptr(tz), _stream(self.device)))
        out = dict(llr=llr, x_hat=xh, z_hat=zh, x_logit=xl, z_logit=zl)
        if want_tape:
            out["tape_x"], out["tape_z"] = tx, tz
        return out

    # ---- Feedback_GNN.call -------------------------------------------------------------------------
    def feedback_gnn(self, weights, llr, logit_hx, logit_hz, synd_x, synd_z):
        B = int(llr.shape[0])
        llr = self._chk(llr, (B, 3, self.n), torch.float32, "llr")
        logit_hx = self._chk(logit_hx, (B, self.m_x), torch.float32, "logit_hx")
        logit_hz = self._chk(logit_hz, (B, self.m_z), torch.float32, "logit_hz")
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        out = self._new((B, 3, self.n), torch.float32)
        check(_lib.lib().fgnn_feedback_gnn(self.handle, weights.handle, _ptr(llr), _ptr(logit_hx), _ptr(logit_hz),
                                           _ptr(synd_x), _ptr(synd_z), B, _ptr(out), _stream(self.device)))
        return out

    # ---- reverse pass of the second training stage (Second_Stage_GNN_BP_Model + tf.GradientTape) ----------
    def bp4_logit_trace(self, llr_ch, synd_x, synd_z, num_iter, factor=1.0, chained=False):
        """Forward of the stage_two decoder with a tape (``chained``: as T chained one-iteration launches, the round-1/2 form kept
        as the cross-check of the one-launch kernel).  Returns
        dict(tape_x [T+1,B,E_x], tape_z [T+1,B,E_z], x_logit [T+1,B,rows0], z_logit [T+1,B,rows1], x_hat, z_hat, llr)."""
        if not chained:  # one launch: the kernel records the soft syndromes and the tape itself (fgnn_bp4_decode_trace)
            return self.bp4_decode_trace(synd_x, synd_z, num_iter, "boxplus-phi", factor, llr_ch=llr_ch, want_tape=True)
        B, T = int(llr_ch.shape[0]), int(num_iter)
        tx = torch.zeros((T + 1, B, self.E_x), dtype=torch.float32, device=self.device)
        tz = torch.zeros((T + 1, B, self.E_z), dtype=torch.float32, device=self.device)
        xl = self._new((T + 1, B, self.rows_xp), torch.float32)
        zl = self._new((T + 1, B, self.rows_zp), torch.float32)
        out = self.bp4_decode(synd_x, synd_z, 0, "boxplus-phi", factor, llr_ch=llr_ch)
        xl[0], zl[0] = out["x_logit"], out["z_logit"]
        for k in range(T):
            out = self.bp4_decode(synd_x, synd_z, 1, "boxplus-phi", factor, llr_ch=llr_ch, msg_init=(tx[k], tz[k]),
                                  return_msgs=True)
            tx[k + 1], tz[k + 1] = out["msg_x"], out["msg_z"]
            xl[k + 1], zl[k + 1] = out["x_logit"], out["z_logit"]
        return dict(tape_x=tx, tape_z=tz, x_logit=xl, z_logit=zl, x_hat=out["x_hat"], z_hat=out["z_hat"], llr=out["llr"])

    def bp4_backward(self, llr_ch, synd_x, synd_z, tape_x, tape_z, grad_x_logit, grad_z_logit, factor=1.0):
        """d loss / d llr_ch [B,3,n] from d loss / d soft syndromes [T+1,B,rows] (fgnn_bp4_backward)."""
        T, B = int(tape_x.shape[0]) - 1, int(llr_ch.shape[0])
        llr_ch = self._chk(llr_ch, (B, 3, self.n), torch.float32, "llr_ch")
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        tape_x = self._chk(tape_x, (T + 1, B, self.E_x), torch.float32, "tape_x")
        tape_z = self._chk(tape_z, (T + 1, B, self.E_z), torch.float32, "tape_z")
        gx = self._chk(grad_x_logit, (T + 1, B, self.rows_xp), torch.float32, "grad_x_logit")
        gz = self._chk(grad_z_logit, (T + 1, B, self.rows_zp), torch.float32, "grad_z_logit")
        has = ((gx != 0).flatten(1).any(1) | (gz != 0).flatten(1).any(1)).to(torch.uint8).contiguous()
        out = self._new((B, 3, self.n), torch.float32)
        check(_lib.lib().fgnn_bp4_backward(self.handle, T, float(factor), _ptr(llr_ch), _ptr(synd_x), _ptr(synd_z), B,
                                           _ptr(tape_x), _ptr(tape_z), _ptr(gx), _ptr(gz), _ptr(has), _ptr(out),
                                           _stream(self.device)))
        return out

    def feedback_gnn_backward(self, weights, llr, logit_hx, logit_hz, synd_x, synd_z, grad_out):
        """Weight gradients (12 tensors, Keras order) of sum(grad_out * Feedback_GNN(...)): the HIP kernel leaves the
        (activation, delta) pairs of the four Dense layers in HBM, the reductions over batch x edges are library GEMMs."""
        B = int(llr.shape[0])
        llr = self._chk(llr, (B, 3, self.n), torch.float32, "llr")
        logit_hx = self._chk(logit_hx, (B, self.m_x), torch.float32, "logit_hx")
        logit_hz = self._chk(logit_hz, (B, self.m_z), torch.float32, "logit_hz")
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        grad_out = self._chk(grad_out, (B, 3, self.n), torch.float32, "grad_out")
        if weights.general:
            # any constructor setting (fgnn_feedback_gnn_backward_general): one (input activations, pre-activation delta) pair per Dense
            # layer in execution order; the gradients come back in the order of get_weights() (_llr_inv_embed first)
            D, H, L, _, _, bias = weights.config
            shapes = [(4 if l == 0 else H, D if l == L - 1 else H) for _ in range(2) for l in range(L)]
            shapes += [(2 * D + 3 if l == 0 else H, H) for l in range(L - 1)]
            shapes += [(H if L > 1 else 2 * D + 3, 3)]
            rows = [B * self.E_x] * L + [B * self.E_z] * L + [B * self.n] * L
            acts = [self._new((r, k), torch.float32) for r, (k, _) in zip(rows, shapes)]
            deltas = [self._new((r, j), torch.float32) for r, (_, j) in zip(rows, shapes)]
            pa = (C.c_void_p * len(acts))(*[t.data_ptr() for t in acts])
            pd = (C.c_void_p * len(deltas))(*[t.data_ptr() for t in deltas])
            check(_lib.lib().fgnn_feedback_gnn_backward_general(self.handle, weights.handle, _ptr(llr), _ptr(logit_hx), _ptr(logit_hz),
                                                                _ptr(synd_x), _ptr(synd_z), B, _ptr(grad_out), pa, pd, len(acts),
                                                                _stream(self.device)))
            grads = []
            for f in range(3 * L):
                li = 3 * L - 1 if f == 0 else f - 1
                grads.append(acts[li].t() @ deltas[li])
                if bias:
                    grads.append(deltas[li].sum(0))
            return grads
        node_in = self._new((B * self.n, 44), torch.float32)
        node_h2 = self._new((B * self.n, 40), torch.float32)
        node_d2 = self._new((B * self.n, 40), torch.float32)
        Es = (self.E_x, self.E_z)
        feat = [self._new((B * e, 4), torch.float32) for e in Es]
        h1 = [self._new((B * e, 40), torch.float32) for e in Es]
        d1 = [self._new((B * e, 40), torch.float32) for e in Es]
        dm = [self._new((B * e, 20), torch.float32) for e in Es]
        pp = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])  # noqa: E731
        check(_lib.lib().fgnn_feedback_gnn_backward(self.handle, weights.handle, _ptr(llr), _ptr(logit_hx), _ptr(logit_hz),
                                                    _ptr(synd_x), _ptr(synd_z), B, _ptr(grad_out), _ptr(node_in),
                                                    _ptr(node_h2), _ptr(node_d2), pp(feat), pp(h1), pp(d1), pp(dm),
                                                    _stream(self.device)))
        G = grad_out.permute(0, 2, 1).reshape(B * self.n, 3)
        grads = [node_h2.t() @ G, G.sum(0)]
        for s in range(2):
            grads += [feat[s].t() @ d1[s], d1[s].sum(0), h1[s].t() @ dm[s], dm[s].sum(0)]
        grads += [node_in[:, :43].t() @ node_d2, node_d2.sum(0)]
        return grads

    # ---- channel / syndromes / flags / residual ---------------------------------------------------------
    def pauli_noise(self, seed, p, first_sample, B, out=None, first_dev=None):
        """``out=(ex, ez)``: write into the given contiguous uint8 [B, n] tensors (row slices of a larger batch).  ``first_dev`` (device
        int64 / uint64 [1]): the stream position is read on the device, sample b = ``first_dev[0] + first_sample + b``
        (fgnn_pauli_noise_dev: Monte-Carlo loops captured in a hipGraph)."""
        if out is None:
            ex = self._new((B, self.n), torch.uint8)
            ez = self._new((B, self.n), torch.uint8)
        else:
            ex = self._chk_out(out[0], (B, self.n), torch.uint8, "noise_x")
            ez = self._chk_out(out[1], (B, self.n), torch.uint8, "noise_z")
        with torch.cuda.device(self.device):  # graph-less entry points run on the current device
            if first_dev is None:
                check(_lib.lib().fgnn_pauli_noise(int(seed), float(np.float32(p)), int(first_sample), B, self.n, _ptr(ex), _ptr(ez),
                                                  _stream(self.device)))
            else:
                if first_dev.device != self.device or first_dev.dtype not in (torch.int64, torch.uint64) or first_dev.numel() != 1:
                    raise ValueError(f"first_dev must be one int64 on {self.device}")
                check(_lib.lib().fgnn_pauli_noise_dev(int(seed), float(np.float32(p)), _ptr(first_dev), int(first_sample), B, self.n,
                                                      _ptr(ex), _ptr(ez), _stream(self.device)))
        return ex, ez

    def pauli_noise_xyz(self, seed, px, py, pz, first_sample, B, out=None):
        """Pauli.call for any triple (px, py, pz), pauli.py:98-108 (fgnn_pauli_noise_xyz): the same Philox uniforms as `pauli_noise`."""
        if out is None:
            ex = self._new((B, self.n), torch.uint8)
            ez = self._new((B, self.n), torch.uint8)
        else:
            ex = self._chk_out(out[0], (B, self.n), torch.uint8, "noise_x")
            ez = self._chk_out(out[1], (B, self.n), torch.uint8, "noise_z")
        with torch.cuda.device(self.device):
            check(_lib.lib().fgnn_pauli_noise_xyz(int(seed), float(np.float32(px)), float(np.float32(py)), float(np.float32(pz)),
                                                  int(first_sample), B, self.n, _ptr(ex), _ptr(ez), _stream(self.device)))
        return ex, ez

    def pauli_noise_wt(self, seed, wt, first_sample, B, out=None):
        if out is None:
            ex = self._new((B, self.n), torch.uint8)
            ez = self._new((B, self.n), torch.uint8)
        else:
            ex = self._chk_out(out[0], (B, self.n), torch.uint8, "noise_x")
            ez = self._chk_out(out[1], (B, self.n), torch.uint8, "noise_z")
        with torch.cuda.device(self.device):
            check(_lib.lib().fgnn_pauli_noise_wt(int(seed), int(wt), int(first_sample), B, self.n, _ptr(ex), _ptr(ez),
                                                 _stream(self.device)))
        return ex, ez

    def syndrome(self, ex, ez):
        B = int(ex.shape[0])
        ex = self._chk(ex, (B, self.n), torch.uint8, "noise_x")
        ez = self._chk(ez, (B, self.n), torch.uint8, "noise_z")
        sx = self._new((B, self.m_x), torch.uint8)
        sz = self._new((B, self.m_z), torch.uint8)
        check(_lib.lib().fgnn_syndrome(self.handle, _ptr(ex), _ptr(ez), B, _ptr(sx), _ptr(sz), _stream(self.device)))
        return sx, sz

    def flag_update(self, x_hat, z_hat, synd_x, synd_z, errors):
        B = int(x_hat.shape[0])
        x_hat = self._chk(x_hat, (B, self.n), torch.uint8, "x_hat")
        z_hat = self._chk(z_hat, (B, self.n), torch.uint8, "z_hat")
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        errors = self._chk_out(errors, (B,), torch.uint8, "errors")
        check(_lib.lib().fgnn_flag_update(self.handle, _ptr(x_hat), _ptr(z_hat), _ptr(synd_x), _ptr(synd_z), B, _ptr(errors),
                                          _stream(self.device)))
        return errors

    def merge(self, errors, x_upd, z_upd, x_hat, z_hat):
        B = int(x_hat.shape[0])
        errors = self._chk(errors, (B,), torch.uint8, "errors")
        x_upd = self._chk(x_upd, (B, self.n), torch.uint8, "x_upd")
        z_upd = self._chk(z_upd, (B, self.n), torch.uint8, "z_upd")
        x_hat = self._chk_out(x_hat, (B, self.n), torch.uint8, "x_hat")
        z_hat = self._chk_out(z_hat, (B, self.n), torch.uint8, "z_hat")
        with torch.cuda.device(self.device):  # fgnn_merge takes no graph: it runs on the current device
            check(_lib.lib().fgnn_merge(_ptr(errors), _ptr(x_upd), _ptr(z_upd), B, self.n, _ptr(x_hat), _ptr(z_hat),
                                        _stream(self.device)))

    def residual(self, ex, ez, x_hat, z_hat, want_arrays=True):
        B = int(ex.shape[0])
        ex = self._chk(ex, (B, self.n), torch.uint8, "noise_x")
        ez = self._chk(ez, (B, self.n), torch.uint8, "noise_z")
        x_hat = self._chk(x_hat, (B, self.n), torch.uint8, "x_hat")
        z_hat = self._chk(z_hat, (B, self.n), torch.uint8, "z_hat")
        s_hat = self._new((B, self.m_z + self.m_x), torch.uint8) if want_arrays else None
        ls_hat = self._new((B, self.rows_hxp + self.rows_hzp), torch.uint8) if want_arrays else None
        flags = self._new((B,), torch.uint8)
        check(_lib.lib().fgnn_residual(self.handle, _ptr(ex), _ptr(ez), _ptr(x_hat), _ptr(z_hat), B, _ptr(s_hat), _ptr(ls_hat),
                                       _ptr(flags), _stream(self.device)))
        return s_hat, ls_hat, flags

    def count_flags(self, flags, counts):
        """counts (uint64/int64 [3] on device) += (#flagged, #block errors, #samples)."""
        flags = self._chk(flags, (int(flags.shape[0]),), torch.uint8, "flags")
        if counts.device != self.device or counts.dtype not in (torch.int64, torch.uint64) or tuple(counts.shape) != (3,) \
                or not counts.is_contiguous():
            raise ValueError(f"counts must be a contiguous int64[3] on {self.device}")
        with torch.cuda.device(self.device):  # fgnn_count_flags takes no graph: it runs on the current device
            check(_lib.lib().fgnn_count_flags(_ptr(flags), int(flags.shape[0]), _ptr(counts), _stream(self.device)))
        return counts

    def count_flags_batches(self, flags, batch, counts, ring):
        """``flags`` holds ``k = len(flags) // batch`` consecutive batches decoded as one launch: ``ring[j]`` (int64 [k, 3] rows of a
        device ring) = the counters after batch j, ``counts`` = the last row (fgnn_count_flags_batches)."""
        B, batch = int(flags.shape[0]), int(batch)
        if batch <= 0 or B % batch:
            raise ValueError("flags must hold a whole number of batches")
        k = B // batch
        flags = self._chk(flags, (B,), torch.uint8, "flags")
        for t, shape, name in ((counts, (3,), "counts"), (ring, (k, 3), "ring")):
            if t.device != self.device or t.dtype not in (torch.int64, torch.uint64) or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous int64{list(shape)} on {self.device}")
        scratch = self._new((2 * k,), torch.int32)
        with torch.cuda.device(self.device):
            check(_lib.lib().fgnn_count_flags_batches(_ptr(flags), k, batch, _ptr(counts), _ptr(ring), _ptr(scratch),
                                                      _stream(self.device)))
        return counts

    # ---- Sandwich body -----------------------------------------------------------------------------------
    def sandwich_workspace(self, B):
        nbytes = _lib.lib().fgnn_sandwich_workspace_bytes(self.handle, int(B))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def sandwich_decode(self, synd_x, synd_z, iters, weights_list, llr_const, factors=None, cn_types=None, compact=False,
                        workspace=None, return_llr=False, return_rounds=False):
        """fgnn_sandwich_decode.  ``compact=True`` runs each feedback round only on the samples still flagged: ``x_hat`` /
        ``z_hat`` / ``rounds`` are identical to the full run, but ``llr`` (``return_llr``) of a sample that left the flagged
        set holds the marginals of the LAST decoder that ran on it, where the full run holds those of the last decoder of the
        stack (which the reference computes for every sample and then ignores, feedback_gnn.py:336-340)."""
        num_layers = len(iters)
        if len(weights_list) != num_layers - 1:
            raise ValueError("need num_layers-1 feedback GNNs")
        if not self.stage_one:
            raise ValueError("the sandwich needs a stage_one graph")
        factors = [1.0] * num_layers if factors is None else list(factors)
        cn_types = ["boxplus-phi"] * num_layers if cn_types is None else list(cn_types)
        it = np.asarray(iters, dtype=np.int32)
        fa = np.asarray(factors, dtype=np.float32)
        ct = np.asarray([CN_TYPES[c] for c in cn_types], dtype=np.int32)
        wh = (C.c_void_p * max(1, num_layers - 1))(*[w.handle for w in weights_list])
        B = int(synd_x.shape[0])
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        if workspace is None:
            workspace = self.sandwich_workspace(B)
        xh = self._new((B, self.n), torch.uint8)
        zh = self._new((B, self.n), torch.uint8)
        llr = self._new((B, 3, self.n), torch.float32) if return_llr else None
        rounds = self._new((B,), torch.uint8) if return_rounds else None
        check(_lib.lib().fgnn_sandwich_decode(self.handle, num_layers, _np_ptr(it), _np_ptr(fa), _np_ptr(ct), wh,
                                              float(llr_const), _ptr(synd_x), _ptr(synd_z), B, int(bool(compact)), _ptr(xh),
                                              _ptr(zh), _ptr(llr), _ptr(rounds), _ptr(workspace), workspace.numel(),
                                              _stream(self.device)))
        out = dict(x_hat=xh, z_hat=zh)
        if return_llr:
            out["llr"] = llr
        if return_rounds:
            out["rounds"] = rounds
        return out

    def forms_agreement(self, synd_x, synd_z, iters, weights_list, llr_const, chunk=16384, factors=None, cn_types=None):
        """The sandwich on the same syndromes under the two OPT-IN re-associations (options 4 and 5 = 1) and under the library's default,
        the reference's formulas term by term (options 4 and 5 = 0: one Dense per edge, feedback_gnn.py:175-184; one log-sum-exp per
        edge, decoding_q.py:254-273), compared per sample: how many samples end on different decisions, how far the marginals of the
        last decoder (and of the first decoder alone) are apart — over all samples and over the samples both forms solve (a sample
        BP does not converge on is chaotic under ANY change of float32 rounding).  Both runs are this library's
        kernels, each bit-equal to the oracle's restatement of its form; the settings in force are restored."""
        B = int(synd_x.shape[0])
        factors = [1.0] * len(iters) if factors is None else list(factors)
        cn_types = ["boxplus-phi"] * len(iters) if cn_types is None else list(cn_types)
        prev = (self.gnn_factored, self.bp4_shared_lse)
        res = dict(samples=B, decisions_differ=0, max_abs_dllr=0.0, samples_gt_1e_4=0, max_abs_dllr_solved=0.0,
                   samples_gt_1e_4_solved=0, flagged_reassociated=0, flagged_literal=0, flagged_in_one_form_only=0,
                   first_decoder=dict(decisions_differ=0, max_abs_dllr=0.0, samples_gt_1e_4=0))
        try:
            for s in range(0, B, chunk):
                sx, sz = synd_x[s:s + chunk].contiguous(), synd_z[s:s + chunk].contiguous()
                ones = torch.ones(sx.shape[0], dtype=torch.uint8, device=self.device)
                outs = []
                for reassociated in (True, False):
                    self.set_gnn_factored(reassociated)
                    self.set_bp4_shared_lse(reassociated)
                    o = self.sandwich_decode(sx, sz, iters, weights_list, llr_const, factors=factors, cn_types=cn_types, return_llr=True)
                    o["flag"] = self.flag_update(o["x_hat"], o["z_hat"], sx, sz, ones.clone()) != 0
                    o["first"] = self.bp4_decode(sx, sz, iters[0], cn_types[0], factors[0], llr_const=llr_const, want_logits=False)
                    outs.append(o)
                a, b = outs
                differ = (a["x_hat"] != b["x_hat"]).any(1) | (a["z_hat"] != b["z_hat"]).any(1)
                d = (a["llr"] - b["llr"]).abs().flatten(1).max(1).values
                solved = ~(a["flag"] | b["flag"])
                res["decisions_differ"] += int(differ.sum())
                res["max_abs_dllr"] = max(res["max_abs_dllr"], float(d.max()))
                res["samples_gt_1e_4"] += int((d > 1e-4).sum())
                if bool(solved.any()):
                    res["max_abs_dllr_solved"] = max(res["max_abs_dllr_solved"], float(d[solved].max()))
                res["samples_gt_1e_4_solved"] += int((d[solved] > 1e-4).sum())
                res["flagged_reassociated"] += int(a["flag"].sum())
                res["flagged_literal"] += int(b["flag"].sum())
                res["flagged_in_one_form_only"] += int((a["flag"] ^ b["flag"]).sum())
                fa, fb, f = a["first"], b["first"], res["first_decoder"]
                d1 = (fa["llr"] - fb["llr"]).abs().flatten(1).max(1).values
                f["decisions_differ"] += int(((fa["x_hat"] != fb["x_hat"]).any(1) | (fa["z_hat"] != fb["z_hat"]).any(1)).sum())
                f["max_abs_dllr"] = max(f["max_abs_dllr"], float(d1.max()))
                f["samples_gt_1e_4"] += int((d1 > 1e-4).sum())
        finally:
            self.set_gnn_factored(prev[0])
            self.set_bp4_shared_lse(prev[1])
        return res

    # ---- OSD-0 (bp_osd.py) -----------------------------------------------------------------------------------------
    def set_basis(self, side, pivot_rows):
        piv = np.ascontiguousarray(pivot_rows, dtype=np.int32)
        check(_lib.lib().fgnn_graph_set_basis(self.handle, int(side), len(piv), _np_ptr(piv)))

    def compact(self, mask, bit=1):
        """Device index list of the samples with (mask & bit) != 0 and their count (one 4-byte device->host read)."""
        B = int(mask.shape[0])
        mask = self._chk(mask, (B,), torch.uint8, "mask")
        index = self._new((max(B, 1),), torch.int32)
        count = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().fgnn_compact(_ptr(mask), int(bit), B, _ptr(index), _ptr(count), _stream(self.device)))
        return index, int(count.item())

    def osd0(self, side, synd, e_hat, marg=None, llr_bin=None, index=None, nact=0):
        """Overwrite e_hat[b] (uint8 [B,n]) for the listed samples with the OSD-0 solution of side 0 (hx) / 1 (hz)."""
        B = int(synd.shape[0])
        synd = self._chk(synd, (B, self.m_x if side == 0 else self.m_z), torch.uint8, "synd")
        e_hat = self._chk_out(e_hat, (B, self.n), torch.uint8, "e_hat")
        if marg is not None:
            marg = self._chk(marg, tuple(marg.shape), torch.float32, "marg")
        if llr_bin is not None:
            llr_bin = self._chk(llr_bin, tuple(llr_bin.shape), torch.float32, "llr_bin")
        if index is not None:
            index = self._chk(index, tuple(index.shape), torch.int32, "index")
        check(_lib.lib().fgnn_osd0(self.handle, int(side), _ptr(marg), _ptr(llr_bin), _ptr(synd), B, _ptr(index), int(nact),
                                   _ptr(e_hat), _stream(self.device)))
        return e_hat

    def residual_rows(self, rows_x, rows_z, ex, ez, x_hat, z_hat):
        B = int(ex.shape[0])
        ex = self._chk(ex, (B, self.n), torch.uint8, "noise_x")
        ez = self._chk(ez, (B, self.n), torch.uint8, "noise_z")
        x_hat = self._chk(x_hat, (B, self.n), torch.uint8, "x_hat")
        z_hat = self._chk(z_hat, (B, self.n), torch.uint8, "z_hat")
        ls_hat = self._new((B, self._row_count(rows_x) + self._row_count(rows_z)), torch.uint8)
        flags = self._new((B,), torch.uint8)
        check(_lib.lib().fgnn_residual_rows(self.handle, int(rows_x), int(rows_z), _ptr(ex), _ptr(ez), _ptr(x_hat), _ptr(z_hat), B, None,
                                            _ptr(ls_hat), _ptr(flags), _stream(self.device)))
        return ls_hat, flags

    def _row_count(self, which):
        return [self.rows_xp, self.rows_zp, self.rows_hxp, self.rows_hzp, self.rows_lx, self.rows_lz][which]

    # ---- binary syndrome BP on the hx graph (LDPCBPDecoder, is_syndrome=True) ----------------------------------
    def bp2_decode(self, synd, num_iter, cn_type="boxplus-phi", factor=1.0, llr_ch=None, llr_const=0.0, B=None, want_soft=True,
                   want_hard=True):
        if cn_type not in CN_TYPES:
            raise ValueError("Unknown node type.")
        if synd is not None:
            B = int(synd.shape[0])
            synd = self._chk(synd, (B, self.m_x), torch.uint8, "syndrome")
        if llr_ch is not None:
            B = int(llr_ch.shape[0]) if B is None else B
            llr_ch = self._chk(llr_ch, (B, self.n), torch.float32, "llr_ch")
        soft = self._new((B, self.n), torch.float32) if want_soft else None
        hard = self._new((B, self.n), torch.uint8) if want_hard else None
        check(_lib.lib().fgnn_bp2_decode(self.handle, CN_TYPES[cn_type], int(num_iter), float(factor), _ptr(llr_ch), float(llr_const),
                                         _ptr(synd), B, _ptr(soft), _ptr(hard), _stream(self.device)))
        return soft, hard

    def bsc_noise(self, seed, p, first_sample, B):
        e = self._new((B, self.n), torch.uint8)
        with torch.cuda.device(self.device):
            check(_lib.lib().fgnn_bsc_noise(int(seed), float(np.float32(p)), int(first_sample), B, self.n, _ptr(e),
                                            _stream(self.device)))
        return e

    # ---- GNN_BP4 -----------------------------------------------------------------------------------------
    def gnn_bp4_decode(self, weights, synd_x, synd_z, num_iter, return_logits=True, workspace=None):
        B = int(synd_x.shape[0])
        synd_x = self._chk(synd_x, (B, self.m_x), torch.uint8, "synd_x")
        synd_z = self._chk(synd_z, (B, self.m_z), torch.uint8, "synd_z")
        nbytes = _lib.lib().fgnn_gnnbp4_weights_workspace_bytes(self.handle, weights.handle, B)
        if workspace is None:
            workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        xh = self._new((B, self.n), torch.uint8)
        zh = self._new((B, self.n), torch.uint8)
        llr = self._new((B, 3, self.n), torch.float32)
        xl = self._new((num_iter, B, self.m_z + self.rows_lz), torch.float32) if return_logits else None
        zl = self._new((num_iter, B, self.m_x + self.rows_lx), torch.float32) if return_logits else None
        check(_lib.lib().fgnn_gnnbp4_decode(self.handle, weights.handle, int(num_iter), _ptr(synd_x), _ptr(synd_z), B, _ptr(xh),
                                            _ptr(zh), _ptr(llr), _ptr(xl), _ptr(zl), _ptr(workspace), workspace.numel(),
                                            _stream(self.device)))
        return dict(x_hat=xh, z_hat=zh, llr=llr, x_logit_all=xl, z_logit_all=zl)


GNNBP4_SHAPES = ([(40, 40), (40,), (40, 20), (20,)] * 2 + [(41, 40), (40,), (40, 20), (20,)] * 2 + [(40, 40), (40,), (40, 20), (20,)] * 2
                 + [(60, 40), (40,), (40, 20), (20,)] + [(20, 3), (3,)])


SHIPPED_GNNBP4_CONFIG = (20, 40, 2, "mean", "tanh", True, False, 0, 0)


def gnnbp4_weight_shapes(graph_or_code, config):
    """Shapes of a GNN_BP4 weight list (order of fgnn.h: 7 MLPs x L Dense, _llr_inv_embed, then the 7 attribute arrays) for
    config = (num_embed_dims, num_hidden_units, num_mlp_layers, reduce_op, activation, use_bias, use_attributes,
    node_attribute_dims, msg_attribute_dims)."""
    D, H, L, _, _, bias, attr, An, Am = config
    if not attr:
        An = Am = 0
    nin = [2 * D + Am] * 2 + [2 * D + An + 1] * 2 + [2 * D + Am] * 2 + [3 * D + An]
    shapes = []
    for q in range(7):
        for k in range(L):
            K, J = (nin[q] if k == 0 else H), (D if k == L - 1 else H)
            shapes.append((K, J))
            if bias:
                shapes.append((J,))
    shapes.append((D, 3))
    if bias:
        shapes.append((3,))
    if attr:
        g = graph_or_code
        if hasattr(g, "hx"):
            hx, hz = np.asarray(g.hx), np.asarray(g.hz)
            n, mx, mz, ex, ez = hx.shape[1], hx.shape[0], hz.shape[0], int(hx.sum()), int(hz.sum())
        else:
            n, mx, mz, ex, ez = g.n, g.m_x, g.m_z, g.E_x, g.E_z
        shapes += [(mx, An), (mz, An), (ex, Am), (ez, Am), (n, An), (ex, Am), (ez, Am)]
    return shapes


class GnnBp4Weights:
    """Device copy of one GNN_BP4 parameter set (order of fgnn.h).  The benchmark configuration (20, 40, 2, mean, tanh, bias, no
    attributes: 30 arrays) runs the MFMA kernel; any other constructor setting — pass ``config`` and the ``graph`` (edge attributes
    are re-ordered for it) — a runtime-shaped kernel (fgnn_gnnbp4_weights_create_general)."""

    def __init__(self, arrays, device, config=SHIPPED_GNNBP4_CONFIG, graph=None, force_general=False):
        arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        D, H, L, red, act, bias, attr, An, Am = config
        if red not in REDUCE_OPS:
            raise ValueError("unknown reduce operation")  # gnn.py:568
        if act not in ACTIVATIONS:
            raise NotImplementedError(f"activation {act!r}: the HIP kernels implement {sorted(k for k in ACTIVATIONS if k)}")
        self.config = (int(D), int(H), int(L), red, act, bool(bias), bool(attr), int(An) if attr else 0, int(Am) if attr else 0)
        self.general = force_general or self.config != SHIPPED_GNNBP4_CONFIG
        self.arrays = arrays
        self.device = _resolve_device(device)
        h = C.c_void_p()
        if not self.general:
            if [a.shape for a in arrays] != GNNBP4_SHAPES:
                raise ValueError(f"GNN_BP4 weights must have shapes {GNNBP4_SHAPES}")
            ptrs = (C.c_void_p * 30)(*[a.ctypes.data for a in arrays])
            check(_lib.lib().fgnn_gnnbp4_weights_create(ptrs, 20, 40, self.device.index, C.byref(h)))
        else:
            if graph is None:
                raise ValueError("a runtime-shaped GNN_BP4 weight set is built for a graph: pass graph=")
            shapes = gnnbp4_weight_shapes(graph, self.config)
            if [a.shape for a in arrays] != shapes:
                raise ValueError(f"GNN_BP4 weights must have shapes {shapes}, got {[a.shape for a in arrays]}")
            cfg = (C.c_int * 9)(self.config[0], self.config[1], self.config[2], REDUCE_OPS[red], ACTIVATIONS[act], int(bool(bias)),
                                int(bool(attr)), self.config[7], self.config[8])
            ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
            check(_lib.lib().fgnn_gnnbp4_weights_create_general(graph.handle, cfg, ptrs, len(arrays), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().fgnn_gnnbp4_weights_destroy(self.handle)
        except Exception:
            pass

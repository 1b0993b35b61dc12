"""ctypes binding of the CPU oracle (oracle/fgnn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (feedback_gnn_amd) never imports this module.
All arrays are NumPy, codeword-major (batch first), exactly the layouts of the C functions.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libfgnn_oracle.so")

CN_TYPES = {"boxplus": 0, "boxplus-phi": 1, "minsum": 2}


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    # always ask make: it is a no-op when the library is newer than fgnn_oracle.c and the two shared headers, and a stale oracle
    # (built before an edit of fgnn_math.h) would otherwise keep checking the kernels against yesterday's arithmetic
    subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.environ.get("FGNN_ORACLE_LIB_PATH", _LIB_PATH))  # override: the sanitizer build of tests/test_oracle_sanitizers.py
        _lib.og_graph_create.restype = C.c_void_p
        _lib.og_graph_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p]
        _lib.og_graph_set_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.og_graph_destroy.argtypes = [C.c_void_p]
        _lib.og_graph_set_gnn_order.argtypes = [C.c_void_p, C.c_int]
        _lib.og_set_num_threads.argtypes = [C.c_int]
        _lib.og_gnn_bp4_general.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 5
        _lib.og_graph_set_vn_shared_lse.argtypes = [C.c_void_p, C.c_int]
        _lib.og_bp4_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_int] + [C.c_void_p] * 9
        _lib.og_feedback_gnn.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_void_p]
        _lib.og_feedback_gnn_general.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]
        _lib.og_pauli_noise.argtypes = [C.c_uint64, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.og_pauli_noise_xyz.argtypes = [C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.og_syndrome.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.og_residual.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3
        _lib.og_sandwich_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        _lib.og_math_apply.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        _lib.og_math_checksums.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.og_num_threads.restype = C.c_int
        _lib.og_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.og_bp2_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.og_bsc_noise.argtypes = [C.c_uint64, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        _lib.og_osd0.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.og_pauli_noise_wt.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.og_gnn_bp4.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _coo(mat):
    r, c = np.nonzero(np.asarray(mat))
    return np.ascontiguousarray(r, dtype=np.int32), np.ascontiguousarray(c, dtype=np.int32)


def num_threads():
    return lib().og_num_threads()


def set_num_threads(n):
    lib().og_set_num_threads(int(n))


def host_cpu_share():
    """CPUs this process may really use: the scheduler affinity capped by the cgroup CPU quota (cpu.max / cfs_quota_us) — on a GPU box
    that shows 256 hardware threads but grants a 16-CPU quota, 16 threads beat 128."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except Exception:
            pass
    return n


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, np.uint32)
    lib().og_philox(_p(c), _p(k), _p(o))
    return o


MATH_FUNCTIONS = {"exp": 0, "log": 1, "log1p": 2, "softplus": 3, "phi": 4, "tanh": 5, "atanh": 6, "phi_gnn": 7, "lse2_corr": 8,
                  "sigmoid": 9, "div3": 10, "rcp_unit": 11, "div_atanh": 12, "lse2_1": 13,
                  # integer-valued probes of fgnn_rng.h (checksums only, not math_apply): a Philox4x32-10 block, the uint32 -> [0,1) map,
                  # the depolarizing thresholds of p
                  "philox": 14, "u32_to_unit": 15, "pauli_thr": 16}


def math_apply(name, x):
    fn = MATH_FUNCTIONS[name]
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().og_math_apply(fn, _p(x), _p(y), x.size)
    return y


def math_checksums(name, lo, hi, chunk_log2=22, ulp=False):
    """Exhaustive probe of one shared-math function over the float32 bit patterns lo..hi: uint64 [windows, 2] checksums of the result
    bits per aligned window of 2^chunk_log2 inputs (og_math_checksums), and with ``ulp`` the largest error against float64 libm in
    float32 ulps and the input bits where it occurs."""
    lo, hi = int(lo), int(hi)
    nwin = (hi >> chunk_log2) - (lo >> chunk_log2) + 1
    out = np.zeros((nwin, 2), dtype=np.uint64)
    worst, at = C.c_double(-1.0), C.c_uint32(0)
    lib().og_math_checksums(MATH_FUNCTIONS[name], lo, hi, int(chunk_log2), _p(out), C.byref(worst) if ulp else None, C.byref(at) if ulp else None)
    return (out, float(worst.value), int(at.value)) if ulp else out


class OracleGraph:
    """Tanner graphs of a CSS code + the row sets the decoder needs.

    ``stage_one=True`` uses pcm_x_perp = hz, pcm_z_perp = hx for the soft syndrome
    (decoding_q.py:35-37); otherwise code.hx_perp / code.hz_perp.
    """

    FORMS = ("library-default", "literal", "reassociated")

    def __init__(self, code, stage_one=True, *, forms):
        """``forms`` (required, so that every caller says which restatement it checks against): "literal" = the reference's formulas
        term by term (one log-sum-exp per edge, decoding_q.py:254-273; one Dense per edge, feedback_gnn.py:175-184, gnn.py:573-610);
        "library-default" = what libfgnn_hip runs out of the box — since round 6 that IS the literal restatement; "reassociated" = the
        library's two opt-in re-associations (FGNN_OPT_BP4_SHARED_LSE, FGNN_OPT_GNN_FACTORED).  All are restated in fgnn_oracle.c;
        set_gnn_order / set_vn_shared_lse switch them one by one afterwards."""
        if forms not in self.FORMS:
            raise ValueError(f"forms must be one of {self.FORMS}")
        L = lib()
        self.code = code
        self.n = int(code.hx.shape[1])
        self.m_x, self.m_z = int(code.hx.shape[0]), int(code.hz.shape[0])
        rx, cx = _coo(code.hx)
        rz, cz = _coo(code.hz)
        self.E_x, self.E_z = len(rx), len(rz)
        self.h = L.og_graph_create(self.n, self.m_x, self.m_z, self.E_x, _p(rx), _p(cx), self.E_z, _p(rz), _p(cz))
        self.forms = forms
        self.set_gnn_order(forms == "reassociated")
        self.set_vn_shared_lse(forms == "reassociated")
        xp, zp = (code.hz, code.hx) if stage_one else (code.hx_perp, code.hz_perp)
        self.rows_xp, self.rows_zp = int(xp.shape[0]), int(zp.shape[0])
        self.rows_lx, self.rows_lz = int(np.asarray(code.lx).shape[0]), int(np.asarray(code.lz).shape[0])
        for which, mat in ((0, xp), (1, zp), (2, code.hx_perp), (3, code.hz_perp), (4, code.lx), (5, code.lz)):
            r, c = _coo(mat)
            L.og_graph_set_rows(self.h, which, int(mat.shape[0]), len(r), _p(r), _p(c))
        self.rows_hxp, self.rows_hzp = int(code.hx_perp.shape[0]), int(code.hz_perp.shape[0])

    def set_gnn_order(self, factored):
        """0 = literal association of feedback_gnn.py:175-184, 1 = factored (same as the library's FGNN_OPT_GNN_FACTORED)."""
        lib().og_graph_set_gnn_order(self.h, int(bool(factored)))
        self.gnn_factored = bool(factored)

    def set_vn_shared_lse(self, shared):
        """0 = one log-sum-exp per edge (decoding_q.py:266, :271 term by term), 1 = its (a - b)-dependent part once per qubit and
        side (same as the library's FGNN_OPT_BP4_SHARED_LSE)."""
        lib().og_graph_set_vn_shared_lse(self.h, int(bool(shared)))
        self.vn_shared_lse = bool(shared)

    def __del__(self):
        try:
            lib().og_graph_destroy(self.h)
        except Exception:
            pass

    # -- QLDPCBPDecoder.call ---------------------------------------------------------------
    def bp4_decode(self, synd_x, synd_z, num_iter, cn_type="boxplus-phi", factor=1.0, llr_ch=None, llr_const=0.0,
                   msg_init=None, return_msgs=False):
        synd_x = np.ascontiguousarray(synd_x, dtype=np.uint8)
        synd_z = np.ascontiguousarray(synd_z, dtype=np.uint8)
        B = synd_x.shape[0]
        assert synd_x.shape == (B, self.m_x) and synd_z.shape == (B, self.m_z)
        if llr_ch is not None:
            llr_ch = np.ascontiguousarray(llr_ch, dtype=np.float32)
            assert llr_ch.shape == (B, 3, self.n)
        mix = miz = None
        if msg_init is not None:
            mix = np.ascontiguousarray(msg_init[0], dtype=np.float32)
            miz = np.ascontiguousarray(msg_init[1], dtype=np.float32)
            assert mix.shape == (B, self.E_x) and miz.shape == (B, self.E_z)
        llr = np.empty((B, 3, self.n), np.float32)
        xh = np.empty((B, self.n), np.uint8)
        zh = np.empty((B, self.n), np.uint8)
        xl = np.empty((B, self.rows_xp), np.float32)
        zl = np.empty((B, self.rows_zp), np.float32)
        mox = np.empty((B, self.E_x), np.float32) if return_msgs else None
        moz = np.empty((B, self.E_z), np.float32) if return_msgs else None
        rc = lib().og_bp4_decode(self.h, CN_TYPES[cn_type], int(num_iter), float(factor), _p(llr_ch), float(llr_const),
                                 _p(synd_x), _p(synd_z), B, _p(mix), _p(miz), _p(llr), _p(xh), _p(zh), _p(xl), _p(zl),
                                 _p(mox), _p(moz))
        assert rc == 0
        out = dict(llr=llr, x_hat=xh, z_hat=zh, x_logit=xl, z_logit=zl)
        if return_msgs:
            out["msg_x"], out["msg_z"] = mox, moz
        return out

    # -- Feedback_GNN.call -------------------------------------------------------------------
    def feedback_gnn(self, weights, llr, logit_hx, logit_hz, synd_x, synd_z):
        w = [np.ascontiguousarray(a, dtype=np.float32) for a in weights]
        wp = (C.c_void_p * 12)(*[a.ctypes.data for a in w])
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        B = llr.shape[0]
        logit_hx = np.ascontiguousarray(logit_hx, dtype=np.float32)
        logit_hz = np.ascontiguousarray(logit_hz, dtype=np.float32)
        synd_x = np.ascontiguousarray(synd_x, dtype=np.uint8)
        synd_z = np.ascontiguousarray(synd_z, dtype=np.uint8)
        assert llr.shape == (B, 3, self.n) and logit_hx.shape == (B, self.m_x) and logit_hz.shape == (B, self.m_z)
        out = np.empty((B, 3, self.n), np.float32)
        rc = lib().og_feedback_gnn(self.h, wp, _p(llr), _p(logit_hx), _p(logit_hz), _p(synd_x), _p(synd_z), B, _p(out))
        assert rc == 0
        return out

    def feedback_gnn_general(self, cfg, weights, llr, logit_hx, logit_hz, synd_x, synd_z):
        """cfg = (num_msg_dims, num_hidden_units, num_mlp_layers, reduce_op 0..3, activation 0..3, use_bias)."""
        w = [np.ascontiguousarray(a, dtype=np.float32) for a in weights]
        wp = (C.c_void_p * len(w))(*[a.ctypes.data for a in w])
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        B = llr.shape[0]
        logit_hx = np.ascontiguousarray(logit_hx, dtype=np.float32)
        logit_hz = np.ascontiguousarray(logit_hz, dtype=np.float32)
        synd_x = np.ascontiguousarray(synd_x, dtype=np.uint8)
        synd_z = np.ascontiguousarray(synd_z, dtype=np.uint8)
        out = np.empty((B, 3, self.n), np.float32)
        rc = lib().og_feedback_gnn_general(self.h, *[int(c) for c in cfg], wp, _p(llr), _p(logit_hx), _p(logit_hz), _p(synd_x),
                                           _p(synd_z), B, _p(out))
        assert rc == 0
        return out

    # -- channel / syndromes / residual -------------------------------------------------------
    def pauli_noise(self, seed, p, first_sample, B):
        ex = np.empty((B, self.n), np.uint8)
        ez = np.empty((B, self.n), np.uint8)
        lib().og_pauli_noise(int(seed), float(np.float32(p)), int(first_sample), B, self.n, _p(ex), _p(ez))
        return ex, ez

    def pauli_noise_xyz(self, seed, px, py, pz, first_sample, B):
        """Pauli.call for any triple (pauli.py:98-108)."""
        ex = np.empty((B, self.n), np.uint8)
        ez = np.empty((B, self.n), np.uint8)
        lib().og_pauli_noise_xyz(int(seed), float(np.float32(px)), float(np.float32(py)), float(np.float32(pz)), int(first_sample), B,
                                 self.n, _p(ex), _p(ez))
        return ex, ez

    def syndrome(self, ex, ez):
        ex = np.ascontiguousarray(ex, dtype=np.uint8)
        ez = np.ascontiguousarray(ez, dtype=np.uint8)
        B = ex.shape[0]
        sx = np.empty((B, self.m_x), np.uint8)
        sz = np.empty((B, self.m_z), np.uint8)
        lib().og_syndrome(self.h, _p(ex), _p(ez), B, _p(sx), _p(sz))
        return sx, sz

    def residual(self, ex, ez, xh, zh):
        B = ex.shape[0]
        s_hat = np.empty((B, self.m_z + self.m_x), np.uint8)
        ls_hat = np.empty((B, self.rows_hxp + self.rows_hzp), np.uint8)
        flags = np.empty((B,), np.uint8)
        lib().og_residual(self.h, _p(np.ascontiguousarray(ex)), _p(np.ascontiguousarray(ez)),
                          _p(np.ascontiguousarray(xh)), _p(np.ascontiguousarray(zh)), B, _p(s_hat), _p(ls_hat), _p(flags))
        return s_hat, ls_hat, flags

    # -- Sandwich_BP_GNN_Evaluation_Model.call (on given syndromes) -------------------------------
    def sandwich_decode(self, synd_x, synd_z, iters, weights_list, llr_const, factors=None, cn_types=None,
                        return_llr=False):
        num_layers = len(iters)
        assert len(weights_list) == num_layers - 1
        factors = [1.0] * num_layers if factors is None else list(factors)
        cn_types = ["boxplus-phi"] * num_layers if cn_types is None else list(cn_types)
        it = np.asarray(iters, dtype=np.int32)
        fa = np.asarray(factors, dtype=np.float32)
        ct = np.asarray([CN_TYPES[c] for c in cn_types], dtype=np.int32)
        keep = []
        outer = (C.c_void_p * max(1, num_layers - 1))()
        for i, wl in enumerate(weights_list):
            w = [np.ascontiguousarray(a, dtype=np.float32) for a in wl]
            inner = (C.c_void_p * 12)(*[a.ctypes.data for a in w])
            keep.append((w, inner))
            outer[i] = C.cast(inner, C.c_void_p).value
        synd_x = np.ascontiguousarray(synd_x, dtype=np.uint8)
        synd_z = np.ascontiguousarray(synd_z, dtype=np.uint8)
        B = synd_x.shape[0]
        xh = np.empty((B, self.n), np.uint8)
        zh = np.empty((B, self.n), np.uint8)
        llr = np.empty((B, 3, self.n), np.float32) if return_llr else None
        rounds = np.empty((B,), np.uint8)
        rc = lib().og_sandwich_decode(self.h, num_layers, _p(it), _p(fa), _p(ct), outer, float(llr_const), _p(synd_x),
                                      _p(synd_z), B, _p(xh), _p(zh), _p(llr), _p(rounds))
        assert rc == 0, rc
        out = dict(x_hat=xh, z_hat=zh, rounds=rounds)
        if return_llr:
            out["llr"] = llr
        return out

    # -- GNN_BP4.call -------------------------------------------------------------------------------------
    def gnn_bp4(self, weights, synd_x, synd_z, num_iter, D=20, H=40):
        w = [np.ascontiguousarray(a, dtype=np.float32) for a in weights]
        assert len(w) == 30
        wp = (C.c_void_p * 30)(*[a.ctypes.data for a in w])
        synd_x = np.ascontiguousarray(synd_x, dtype=np.uint8)
        synd_z = np.ascontiguousarray(synd_z, dtype=np.uint8)
        B = synd_x.shape[0]
        xh = np.empty((B, self.n), np.uint8)
        zh = np.empty((B, self.n), np.uint8)
        llr = np.empty((B, 3, self.n), np.float32)
        xl = np.empty((num_iter, B, self.m_z + self.rows_lz), np.float32)
        zl = np.empty((num_iter, B, self.m_x + self.rows_lx), np.float32)
        rc = lib().og_gnn_bp4(self.h, wp, D, H, int(num_iter), _p(synd_x), _p(synd_z), B, _p(xh), _p(zh), _p(llr), _p(xl), _p(zl))
        assert rc == 0
        return dict(x_hat=xh, z_hat=zh, llr=llr, x_logit_all=xl, z_logit_all=zl)

    def gnn_bp4_general(self, cfg, weights, synd_x, synd_z, num_iter):
        """cfg = (D, H, L, reduce_op 0..3, activation 0..3, use_bias, use_attributes, An, Am); weights in the order of
        og_gnn_bp4_general (7 MLPs x L Dense, _llr_inv_embed, then the 7 attribute arrays if use_attributes)."""
        w = [np.ascontiguousarray(a, dtype=np.float32) for a in weights]
        wp = (C.c_void_p * len(w))(*[a.ctypes.data for a in w])
        c = (C.c_int * 9)(*[int(x) for x in cfg])
        synd_x = np.ascontiguousarray(synd_x, dtype=np.uint8)
        synd_z = np.ascontiguousarray(synd_z, dtype=np.uint8)
        B = synd_x.shape[0]
        xh = np.empty((B, self.n), np.uint8)
        zh = np.empty((B, self.n), np.uint8)
        llr = np.empty((B, 3, self.n), np.float32)
        xl = np.empty((num_iter, B, self.m_z + self.rows_lz), np.float32)
        zl = np.empty((num_iter, B, self.m_x + self.rows_lx), np.float32)
        rc = lib().og_gnn_bp4_general(self.h, c, wp, len(w), int(num_iter), _p(synd_x), _p(synd_z), B, _p(xh), _p(zh), _p(llr), _p(xl),
                                      _p(zl))
        assert rc == 0, rc
        return dict(x_hat=xh, z_hat=zh, llr=llr, x_logit_all=xl, z_logit_all=zl)

    # -- binary syndrome BP (LDPCBPDecoder, is_syndrome=True) on the hx graph ---------------------------------
    def bp2_decode(self, synd, num_iter, cn_type="boxplus-phi", factor=1.0, llr_ch=None, llr_const=0.0):
        B = synd.shape[0]
        synd = np.ascontiguousarray(synd, dtype=np.uint8)
        if llr_ch is not None:
            llr_ch = np.ascontiguousarray(llr_ch, dtype=np.float32)
        soft = np.empty((B, self.n), np.float32)
        hard = np.empty((B, self.n), np.uint8)
        rc = lib().og_bp2_decode(self.h, CN_TYPES[cn_type], int(num_iter), float(factor), _p(llr_ch), float(llr_const), _p(synd), B,
                                 _p(soft), _p(hard))
        assert rc == 0
        return soft, hard

    def bsc_noise(self, seed, p, first_sample, B):
        e = np.empty((B, self.n), np.uint8)
        lib().og_bsc_noise(int(seed), float(np.float32(p)), int(first_sample), B, self.n, _p(e))
        return e

    # -- OSD-0 (bp_osd.py:14-77) -------------------------------------------------------------------------------
    def osd0(self, side, pivot_rows, synd, marg=None, llr_bin=None, index=None, e_hat=None):
        piv = np.ascontiguousarray(pivot_rows, dtype=np.int32)
        synd = np.ascontiguousarray(synd, dtype=np.uint8)
        B = synd.shape[0]
        if marg is not None:
            marg = np.ascontiguousarray(marg, dtype=np.float32)
        if llr_bin is not None:
            llr_bin = np.ascontiguousarray(llr_bin, dtype=np.float32)
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        if e_hat is None:
            e_hat = np.zeros((B, self.n), np.uint8)
        rc = lib().og_osd0(self.h, int(side), len(piv), _p(piv), _p(marg), _p(llr_bin), _p(synd), B, _p(idx),
                           0 if idx is None else len(idx), _p(e_hat))
        assert rc == 0
        return e_hat

    def pauli_noise_wt(self, seed, wt, first_sample, B):
        ex = np.empty((B, self.n), np.uint8)
        ez = np.empty((B, self.n), np.uint8)
        rc = lib().og_pauli_noise_wt(int(seed), int(wt), int(first_sample), B, self.n, _p(ex), _p(ez))
        assert rc == 0
        return ex, ez

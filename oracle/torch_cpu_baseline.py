"""TF-CPU-shaped baseline: the reference's BP4 + feedback-GNN sandwich restated op for op on torch CPU tensors.

TEST / BENCH INFRASTRUCTURE ONLY (bench.py's `cpu_baseline_tf_like` leg and tests/): never imported by the product.

TensorFlow cannot run on either box, so the "reference CPU path" cannot be timed itself.  The C oracle
(fgnn_oracle.c, OpenMP over codewords, scalar loops that keep one codeword in L1) is a much better CPU program than the
reference's; this module is the closest analogue of how /root/reference sionna/fec/ldpc/decoding_q.py:732-767 actually executes on
a host: batch-minor float32 tensors [E, B], one framework op at a time, every intermediate materialised —

    per iteration   totals      ragged reduce_sum over a qubit's edges   -> index_add_         (:244-248)
                    v->c        gather totals per edge, softplus, stacked reduce_logsumexp     (:254-273)
                    to CN order gather with the CN permutation                                  (:752-753)
                    c->v        sign / reduce_prod, phi, ragged reduce_sum, phi, multiply       (:376-431)
                    back        gather with the inverse permutation                             (:766-767)
    epilogue        marginals, argmin decision, cal_logit                                       (:771-790, :455-471)
    Feedback_GNN    gathers that materialise [B,E,4] -> Dense(40,tanh) -> Dense(20) -> ragged reduce_mean -> Dense(40,tanh) -> Dense(3)
                                                                                                (feedback_gnn.py:161-188)
    sandwich        flag test by int64 matmul mod 2, masked merge                               (feedback_gnn.py:321-340)

with torch's own exp / log / log1p / tanh and `torch.set_num_threads(all cores)`.  Results agree with the C oracle the way two
faithful float32 implementations do (identical decisions on converged samples); bench.py checks that on every run.
"""
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore", message="index_reduce")

F = torch.float32
_THR = float(np.float32(np.log(np.finfo(np.float32).eps, dtype=np.float32) + np.float32(2.0)))  # tf2xla Softplus threshold (negative)
PHI_MIN, PHI_MAX = 8.5e-8, 16.635532


def softplus(t):
    e = torch.exp(torch.clamp(t, max=30.0))
    return torch.where(t > -_THR, t, torch.where(t < _THR, e, torch.log1p(e)))


def lse2(a, b):
    m = torch.maximum(a, b)
    return torch.log(torch.exp(a - m) + torch.exp(b - m)) + m


def phi(x):
    x = torch.clamp(x, PHI_MIN, PHI_MAX)
    return softplus(x) - torch.log(torch.exp(x) - 1.0)


class Graph:
    """Edge tables of decoding_q.py:53-94 as index tensors (VN-major edges, CN permutation and its inverse)."""

    def __init__(self, code):
        self.n = int(code.hx.shape[1])
        self.sides = []
        for pcm in (np.asarray(code.hx), np.asarray(code.hz)):
            chk, var = np.nonzero(pcm)
            o = np.lexsort((chk, var))
            chk, var = chk[o], var[o]
            to_cn = np.lexsort((var, chk))
            self.sides.append(dict(m=pcm.shape[0], var=torch.from_numpy(var.astype(np.int64)), chk=torch.from_numpy(chk.astype(np.int64)),
                                   to_cn=torch.from_numpy(to_cn.astype(np.int64)), inv=torch.from_numpy(np.argsort(to_cn).astype(np.int64)),
                                   cn_of=torch.from_numpy(chk[to_cn].astype(np.int64)),
                                   deg=torch.from_numpy(np.bincount(var, minlength=self.n).astype(np.float32))))
        # stage-one soft-syndrome rows: x_logit over hz rows, z_logit over hx rows (decoding_q.py:35-37)
        self.rows = []
        for pcm in (np.asarray(code.hz), np.asarray(code.hx)):
            r, c = np.nonzero(pcm)
            self.rows.append(dict(rows=pcm.shape[0], row=torch.from_numpy(r.astype(np.int64)), col=torch.from_numpy(c.astype(np.int64))))
        self.hx = torch.from_numpy(np.asarray(code.hx).astype(np.int64))
        self.hz = torch.from_numpy(np.asarray(code.hz).astype(np.int64))


def _seg_sum(vals, index, size):
    out = torch.zeros((size,) + tuple(vals.shape[1:]), dtype=vals.dtype)
    return out.index_add_(0, index, vals)


def _seg_sign(neg, index, size):
    """ragged reduce_prod of +-1 signs: the parity of the negative entries."""
    cnt = _seg_sum(neg.to(F), index, size)
    return 1.0 - 2.0 * torch.remainder(cnt, 2.0)


def _cn_phi(nu_cn, side, sigma, factor):
    neg = nu_cn < 0
    sgn = torch.where(neg, -1.0, 1.0)
    prod = _seg_sign(neg, side["cn_of"], side["m"]) * sigma
    a = phi(torch.abs(nu_cn))
    T = _seg_sum(a, side["cn_of"], side["m"])
    return sgn * prod[side["cn_of"]] * phi(T[side["cn_of"]] - a) * factor


def _cn_tanh(nu_cn, side, sigma, factor):
    """cn_type='boxplus', decoding_q.py:313-363: tanh(msg / 2), exact zeros -> 1e-12, ragged reduce_prod times the syndrome sign,
    own edge divided out (msg**-1), |.| < 1e-7 -> 0, clip to +-(1 - 1e-7), 2 atanh."""
    t = torch.tanh(nu_cn / 2)
    t = torch.where(t == 0, torch.full_like(t, 1e-12), t)
    prod = torch.ones((side["m"], t.shape[1]), dtype=F).index_reduce_(0, side["cn_of"], t, "prod") * sigma
    q = t ** -1 * prod[side["cn_of"]]
    q = torch.where(torch.abs(q) < 1e-7, torch.zeros_like(q), q)
    q = torch.clamp(q, -(1.0 - 1e-7), 1.0 - 1e-7)
    return 2 * torch.atanh(q) * factor


CN_RULES = {"boxplus-phi": _cn_phi, "boxplus": _cn_tanh}


def _logits(rows, llr):
    v = llr[rows["col"]]
    prod = _seg_sign(v < 0, rows["row"], rows["rows"])
    return prod * phi(_seg_sum(phi(torch.abs(v)), rows["row"], rows["rows"]))


def bp4_decode(g, synd_x, synd_z, num_iter, llr_ch, factor=1.0, cn_type="boxplus-phi"):
    """synd_* [m,B] (0/1 float), llr_ch [3,n,B] -> X, Y, Z [n,B], x_hat, z_hat [n,B] int64, x_logit [m_z,B], z_logit [m_x,B]."""
    B = synd_x.shape[1]
    cn_rule = CN_RULES[cn_type]
    sig = [1.0 - 2.0 * synd_x, 1.0 - 2.0 * synd_z]
    msg = [torch.zeros((s["var"].numel(), B), dtype=F) for s in g.sides]

    def totals():
        Sx = _seg_sum(msg[0], g.sides[0]["var"], g.n)
        Sz = _seg_sum(msg[1], g.sides[1]["var"], g.n)
        return Sz + llr_ch[0], (Sz + Sx) + llr_ch[1], Sx + llr_ch[2]

    for _ in range(num_iter):
        X, Y, Z = totals()
        vx, vz = g.sides[0]["var"], g.sides[1]["var"]
        nux = softplus(-X)[vx] - lse2(-(Z[vx] - msg[0]), -(Y[vx] - msg[0]))
        nuz = softplus(-Z)[vz] - lse2(-(X[vz] - msg[1]), -(Y[vz] - msg[1]))
        for s, nu in ((0, nux), (1, nuz)):
            side = g.sides[s]
            msg[s] = cn_rule(nu[side["to_cn"]], side, sig[s], factor)[side["inv"]]
    X, Y, Z = totals()
    dec = torch.argmin(torch.stack([torch.zeros_like(X), X, Z, Y], 0), dim=0)
    x_hat = dec % 2
    z_hat = (dec - x_hat) // 2
    llr_z = softplus(-X) - lse2(-Z, -Y)
    llr_x = softplus(-Z) - lse2(-X, -Y)
    return X, Y, Z, x_hat, z_hat, _logits(g.rows[0], llr_x), _logits(g.rows[1], llr_z)


def feedback_gnn(g, w, X, Y, Z, logit_hx, logit_hz, synd_x, synd_z):
    """Feedback_GNN.call with the reference's materialised gathers; returns the new (L_X, L_Y, L_Z) as [3,n,B]."""
    h_vn = torch.stack([X, Y, Z], -1).transpose(0, 1).contiguous()  # [B,n,3]
    means = []
    for s, (logit, synd, k) in enumerate(((logit_hx, synd_x, 2), (logit_hz, synd_z, 6))):
        side = g.sides[s]
        h_cn = (logit * (1.0 - 2.0 * synd)).t().unsqueeze(-1)  # [B,m,1]
        feat = torch.cat([h_cn[:, side["chk"], :], h_vn[:, side["var"], :]], -1)  # [B,E,4]
        m = torch.tanh(feat @ w[k] + w[k + 1]) @ w[k + 2] + w[k + 3]  # [B,E,20]
        ssum = torch.zeros((m.shape[0], g.n, m.shape[2]), dtype=F).index_add_(1, side["var"], m)
        means.append(ssum / side["deg"][None, :, None])
    z = torch.cat([means[0], means[1], h_vn], -1)
    o = torch.tanh(z @ w[10] + w[11]) @ w[0] + w[1]  # [B,n,3]
    return o.permute(2, 1, 0).contiguous()


def sandwich_decode(g, weights, synd_x, synd_z, iters, llr_const, factors=None, cn_types=None):
    """Sandwich_BP_GNN_Evaluation_Model.call from the syndromes on (feedback_gnn.py:311-340).  synd_* uint8 [B,m] codeword-major
    (as the oracle takes them); returns x_hat, z_hat uint8 [B,n]."""
    w = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in weights] if weights is not None else None
    sx = torch.from_numpy(np.ascontiguousarray(synd_x)).t().contiguous()
    sz = torch.from_numpy(np.ascontiguousarray(synd_z)).t().contiguous()
    sxf, szf = sx.to(F), sz.to(F)
    B = sx.shape[1]
    factors = [1.0] * len(iters) if factors is None else factors
    cn_types = ["boxplus-phi"] * len(iters) if cn_types is None else cn_types
    llr = torch.full((3, g.n, B), float(llr_const), dtype=F)
    X, Y, Z, xh, zh, xl, zl = bp4_decode(g, sxf, szf, iters[0], llr, factors[0], cn_types[0])
    errors = torch.ones(B, dtype=torch.bool)
    for i in range(1, len(iters)):
        flag = ((g.hz @ xh) % 2 != sz.to(torch.int64)).any(0) | ((g.hx @ zh) % 2 != sx.to(torch.int64)).any(0)
        errors = errors & flag
        llr = feedback_gnn(g, w, X, Y, Z, zl, xl, sxf, szf)  # the logit swap of feedback_gnn.py:335
        X, Y, Z, xn, zn, xl, zl = bp4_decode(g, sxf, szf, iters[i], llr, factors[i], cn_types[i])
        xh = torch.where(errors[None, :], xn, xh)
        zh = torch.where(errors[None, :], zn, zh)
    return xh.t().contiguous().to(torch.uint8).numpy(), zh.t().contiguous().to(torch.uint8).numpy()

"""Differentiable float64 PyTorch restatement of the training objective (TEST INFRASTRUCTURE ONLY).

Third independent statement of the reference arithmetic, written with torch ops so that autograd supplies the exact
gradients of Second_Stage_GNN_BP_Model.call (/root/reference sionna/fec/ldpc/feedback_gnn.py:423-463) — GNN → 16 BP
iterations with per-iteration soft syndromes → sum of BCE losses — with respect to the GNN weights and the channel LLRs.
The hand-written backward kernels (feedback_gnn_amd/csrc/fgnn_backward.hip) are checked against these gradients.

Gradient conventions follow TensorFlow's registered gradients: softplus' = sigmoid; clip_by_value passes the gradient inside
[lo, hi] and blocks it outside; sign() and the stop_gradient'ed sign products carry none (decoding_q.py:392-409, :428);
reduce_logsumexp' = softmax.
"""
import numpy as np
import torch

DT = torch.float64
PHI_MIN, PHI_MAX = 8.5e-8, 16.635532


def phi(x):
    xc = torch.clamp(x, PHI_MIN, PHI_MAX)
    return torch.nn.functional.softplus(xc) - torch.log(torch.expm1(xc))


class Graph:
    def __init__(self, code):
        self.n = code.hx.shape[1]
        self.sides = []
        for pcm in (code.hx, code.hz):
            chk, var = np.nonzero(np.asarray(pcm))
            o = np.lexsort((chk, var))  # canonical VN-major order (qubit, check)
            self.sides.append(dict(chk=torch.from_numpy(chk[o]), var=torch.from_numpy(var[o]), m=pcm.shape[0]))


def _scatter_sum(vals, idx, size):
    out = torch.zeros((vals.shape[0], size) + vals.shape[2:], dtype=vals.dtype)
    return out.index_add(1, idx, vals)


def bp4_logit_trace(g, llr_ch, synd_x, synd_z, num_iter, factor=1.0):
    """llr_ch [B,3,n] (requires_grad ok), syndromes [B,m] 0/1.  Returns (x_logits, z_logits) lists of length num_iter+1
    (soft syndromes over the hz / hx rows after 0..num_iter iterations, stage_two layout) and the final marginals."""
    B = llr_ch.shape[0]
    sig = [1.0 - 2.0 * synd_x.to(DT), 1.0 - 2.0 * synd_z.to(DT)]
    msg = [torch.zeros((B, s["chk"].numel()), dtype=DT) for s in g.sides]
    Lx, Ly, Lz = llr_ch[:, 0], llr_ch[:, 1], llr_ch[:, 2]
    xs, zs = [], []
    sp = torch.nn.functional.softplus

    def lse(a, b):
        return torch.logsumexp(torch.stack([a, b], -1), -1)

    def totals():
        Sx = _scatter_sum(msg[0], g.sides[0]["var"], g.n)
        Sz = _scatter_sum(msg[1], g.sides[1]["var"], g.n)
        return Sz + Lx, (Sz + Sx) + Ly, Sx + Lz

    def logits(X, Y, Z):
        llr_z = sp(-X) - lse(-Z, -Y)
        llr_x = sp(-Z) - lse(-X, -Y)
        out = []
        for s, llr in ((1, llr_x), (0, llr_z)):  # x_logit over hz rows with llr_x; z_logit over hx rows with llr_z
            side = g.sides[s]
            v = llr[:, side["var"]]
            neg = _scatter_sum((v < 0).to(DT), side["chk"], side["m"])
            sgn = 1.0 - 2.0 * torch.remainder(neg, 2.0)
            T = _scatter_sum(phi(v.abs()), side["chk"], side["m"])
            out.append(sgn.detach() * phi(T))
        return out

    for it in range(num_iter + 1):
        X, Y, Z = totals()
        xl, zl = logits(X, Y, Z)
        xs.append(xl)
        zs.append(zl)
        if it == num_iter:
            break
        new = []
        for s, (A, Bt) in enumerate(((X, Z), (Z, X))):
            side = g.sides[s]
            v, c = side["var"], side["chk"]
            nu = sp(-A)[:, v] - lse(-(Bt[:, v] - msg[s]), -(Y[:, v] - msg[s]))
            neg = (nu < 0).to(DT)
            par = torch.remainder(_scatter_sum(neg, c, side["m"]), 2.0)
            S = sig[s] * (1.0 - 2.0 * par)
            a = phi(nu.abs())
            T = _scatter_sum(a, c, side["m"])
            out = ((1.0 - 2.0 * neg) * S[:, c]).detach() * phi(T[:, c] - a)
            new.append(out * factor)
        msg = new
    return xs, zs, (X, Y, Z)


def feedback_gnn(g, w, llr, logit_hx, logit_hz, synd_x, synd_z):
    """w: list of 12 torch tensors (requires_grad ok); llr [B,3,n]; returns [B,3,n]."""
    h_vn = llr.permute(0, 2, 1)
    ms = []
    for s, (logit, synd, k) in enumerate(((logit_hx, synd_x, 2), (logit_hz, synd_z, 6))):
        side = g.sides[s]
        h_cn = (logit * (1.0 - 2.0 * synd.to(DT)))[:, :, None]
        feat = torch.cat([h_cn[:, side["chk"], :], h_vn[:, side["var"], :]], -1)
        m = torch.tanh(feat @ w[k] + w[k + 1]) @ w[k + 2] + w[k + 3]
        deg = torch.bincount(side["var"], minlength=g.n).to(DT)
        ms.append(_scatter_sum(m, side["var"], g.n) / deg[None, :, None])
    z = torch.cat([ms[0], ms[1], h_vn], -1)
    out = torch.tanh(z @ w[10] + w[11]) @ w[0] + w[1]
    return out.permute(0, 2, 1)


def feedback_gnn_general(g, cfg, w, llr, logit_hx, logit_hz, synd_x, synd_z):
    """Any constructor setting of Feedback_GNN (feedback_gnn.py:110-150): cfg = (D, H, L, reduce_op, activation, use_bias) with
    reduce_op 0 sum / 1 mean / 2 max / 3 min and activation 0 linear / 1 tanh / 2 relu / 3 sigmoid; w in get_weights() order."""
    D, H, L, red, act, bias = cfg
    f = {0: lambda t: t, 1: torch.tanh, 2: torch.relu, 3: torch.sigmoid}[act]
    st = 2 if bias else 1

    def dense(x, idx, a):
        y = x @ w[idx * st]
        if bias:
            y = y + w[idx * st + 1]
        return a(y)

    h_vn = llr.permute(0, 2, 1)
    ms = []
    for s, (logit, synd) in enumerate(((logit_hx, synd_x), (logit_hz, synd_z))):
        side = g.sides[s]
        h_cn = (logit * (1.0 - 2.0 * synd.to(DT)))[:, :, None]
        x = torch.cat([h_cn[:, side["chk"], :], h_vn[:, side["var"], :]], -1)
        for k in range(L):
            x = dense(x, 1 + s * L + k, (lambda t: t) if k == L - 1 else f)
        cols = []
        for v in range(g.n):  # small test graphs only
            sel = x[:, side["var"] == v, :]
            if sel.shape[1] == 0:
                cols.append(torch.zeros((x.shape[0], D), dtype=DT))
            elif red == 0:
                cols.append(sel.sum(1))
            elif red == 1:
                cols.append(sel.mean(1))
            elif red == 2:
                cols.append(sel.max(1).values)
            else:
                cols.append(sel.min(1).values)
        ms.append(torch.stack(cols, 1))
    z = torch.cat([ms[0], ms[1], h_vn], -1)
    for k in range(L - 1):
        z = dense(z, 1 + 2 * L + k, f)
    return dense(z, 0, lambda t: t).permute(0, 2, 1)


def second_stage_loss(g, w, llr_in, logit_hx, logit_hz, synd_x, synd_z, num_iter=16, loss_from=8, factor=1.0):
    """The scalar training loss of feedback_gnn.py:434-442 and the new channel LLRs it was computed from."""
    new_llr = feedback_gnn(g, w, llr_in, logit_hx, logit_hz, synd_x, synd_z)
    xs, zs, _ = bp4_logit_trace(g, new_llr, synd_x, synd_z, num_iter, factor)
    gt_x = 1.0 - synd_z.to(DT)
    gt_z = 1.0 - synd_x.to(DT)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    loss = torch.zeros((), dtype=DT)
    for i in range(loss_from, num_iter):
        loss = loss + bce(xs[i + 1], gt_x) + bce(zs[i + 1], gt_z)
    return loss, new_llr

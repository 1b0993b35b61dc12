#!/usr/bin/env python3
"""Fit tanh(x) ~ x * P(x^2) / Q(x^2) on |x| <= XMAX (beyond it tanh rounds to +-1 in float32) for fg_tanh
(feedback_gnn_amd/csrc/fgnn_math.h).  Loeb's linearised iteration for the weighted rational minimax problem on a dense mpmath grid;
Q is normalised to Q(0) = 1 and P(0) = 1 is imposed (tanh x ~ x for small x keeps full relative accuracy there).

    python tools/fit_tanh.py [degP degQ xmax]
"""
import sys

import mpmath as mp
import numpy as np

mp.mp.prec = 120


def fit(dp, dq, xmax, N=6000, iters=40):
    k = np.arange(N)
    # Chebyshev-like nodes in x, denser near 0 and xmax
    xs = xmax * 0.5 * (1 - np.cos(np.pi * (k + 0.5) / N))
    xs = xs[xs > 1e-4]
    z = xs ** 2
    f = np.array([float(mp.tanh(mp.mpf(x)) / mp.mpf(x)) for x in xs])  # target for P/Q
    # unknowns: p1..p_dp (P = 1 + sum p_i z^i), q1..q_dq (Q = 1 + sum q_i z^i);  P - f Q = 0
    qprev = np.ones_like(xs)
    w = np.ones_like(xs)
    best = None
    for it in range(iters):
        A = np.concatenate([np.stack([z ** i for i in range(1, dp + 1)], 1), -f[:, None] * np.stack([z ** i for i in range(1, dq + 1)], 1)], 1)
        rhs = f - 1.0
        sc = np.sqrt(w) / (qprev * f)  # relative error of P/Q
        # column scaling for conditioning
        cs = np.abs(A).max(0)
        sol, *_ = np.linalg.lstsq((A / cs) * sc[:, None], rhs * sc, rcond=None)
        sol = sol / cs
        p = np.concatenate([[1.0], sol[:dp]])
        q = np.concatenate([[1.0], sol[dp:]])
        P = sum(p[i] * z ** i for i in range(dp + 1))
        Q = sum(q[i] * z ** i for i in range(dq + 1))
        err = np.abs(P / Q / f - 1.0)
        if best is None or err.max() < best[0]:
            best = (err.max(), p.copy(), q.copy())
        qprev = np.abs(Q)
        w = w * (err + 1e-300)
        w /= w.sum()
    return best


if __name__ == "__main__":
    cfgs = [(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]))] if len(sys.argv) > 3 else [(6, 3, 9.0), (5, 3, 9.0), (5, 4, 9.0), (4, 4, 9.0), (6, 4, 9.0)]
    for dp, dq, xmax in cfgs:
        e, p, q = fit(dp, dq, xmax)
        print(f"P deg {dp}, Q deg {dq}, |x| <= {xmax}: max rel err {e:.3e} = {e / 2 ** -24:.3f} ulp(2^-24)")
        print("  P:", ", ".join(f"{v:.9e}f" for v in p))
        print("  Q:", ", ".join(f"{v:.9e}f" for v in q))

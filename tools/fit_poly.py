#!/usr/bin/env python3
"""Fit the polynomial coefficients used by feedback_gnn_amd/csrc/fgnn_math.h.

The BP4 kernels and the CPU oracle share ONE set of f32 exp/log routines built only
from IEEE-754 fma/add/mul and integer ops, so that GPU and CPU results agree bit for bit.
This script produces the near-minimax coefficients (Lawson iteratively-reweighted least
squares on a dense mpmath grid) and prints them as C float literals.

  exp : e^r  ~ 1 + r + r^2*(c2 + c3 r + ... + cD r^(D-2)),   r in [-ln2/2, ln2/2]
  log : log1p(f) ~ f - f^2/2 + f^3*(p0 + p1 f + ... + pD f^D), f in [sqrt(.5)-1, sqrt(2)-1]

Run: python tools/fit_poly.py
"""
import numpy as np
import mpmath as mp

mp.mp.prec = 200


def lawson(xs, target, basis, weight, iters=60):
    """min max |weight*(basis@c - target)| via Lawson's algorithm."""
    w = np.ones_like(xs)
    A = basis
    c = None
    for _ in range(iters):
        sw = np.sqrt(w) * weight
        c, *_ = np.linalg.lstsq(A * sw[:, None], target * sw, rcond=None)
        err = np.abs(weight * (A @ c - target))
        w = w * (err + 1e-300)
        w = w / w.sum()
    err = np.abs(weight * (A @ c - target))
    return c, err.max()


def fit_exp(deg):
    lo, hi = -float(mp.log(2)) / 2, float(mp.log(2)) / 2
    N = 4001
    k = np.arange(N)
    xs = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (k + 0.5) / N)
    # target for q(r) = (e^r - 1 - r)/r^2
    tgt = np.array([float((mp.e ** mp.mpf(x) - 1 - mp.mpf(x)) / mp.mpf(x) ** 2) if x != 0 else 0.5 for x in xs])
    basis = np.stack([xs ** i for i in range(deg - 1)], axis=1)
    weight = xs ** 2 / np.exp(xs)  # relative error of e^r
    c, e = lawson(xs, tgt, basis, weight)
    return c, e


def fit_log(deg):
    lo, hi = float(mp.sqrt(0.5)) - 1, float(mp.sqrt(2)) - 1
    N = 4001
    k = np.arange(N)
    xs = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (k + 0.5) / N)
    tgt = np.array([float((mp.log1p(mp.mpf(x)) - mp.mpf(x) + mp.mpf(x) ** 2 / 2) / mp.mpf(x) ** 3) for x in xs])
    basis = np.stack([xs ** i for i in range(deg + 1)], axis=1)
    l1p = np.array([float(mp.log1p(mp.mpf(x))) for x in xs])
    weight = np.abs(xs ** 3 / l1p)  # relative error of log1p(f)
    c, e = lawson(xs, tgt, basis, weight)
    return c, e


def fit_tanh(deg):
    """tanh(x) ~ x + x^3 * T(x^2) on |x| <= 0.55 (small-argument branch; avoids the
    cancellation of 1 - 2/(e^{2x}+1))."""
    hi = 0.55
    N = 4001
    k = np.arange(N)
    xs = 0.5 * hi + 0.5 * hi * np.cos(np.pi * (k + 0.5) / N)
    xs = xs[xs > 1e-6]
    tgt = np.array([float((mp.tanh(mp.mpf(x)) - mp.mpf(x)) / mp.mpf(x) ** 3) for x in xs])
    basis = np.stack([(xs ** 2) ** i for i in range(deg + 1)], axis=1)
    th = np.array([float(mp.tanh(mp.mpf(x))) for x in xs])
    weight = np.abs(xs ** 3 / th)
    c, e = lawson(xs, tgt, basis, weight)
    return c, e


def cfloat(x):
    f = np.float32(x)
    return f"{f:.9e}f  /* {float(f).hex()} */"


if __name__ == "__main__":
    for deg in (5, 6, 7):
        c, e = fit_exp(deg)
        print(f"exp degree {deg}: max rel err {e:.3e} ({e / 2**-24:.3f} x 2^-24)")
        for i, ci in enumerate(c):
            print(f"   c{i + 2} = {cfloat(ci)}")
    for deg in (6, 7, 8):
        c, e = fit_log(deg)
        print(f"log P degree {deg}: max rel err {e:.3e} ({e / 2**-24:.3f} x 2^-24)")
        for i, ci in enumerate(c):
            print(f"   p{i} = {cfloat(ci)}")
    for deg in (3, 4, 5):
        c, e = fit_tanh(deg)
        print(f"tanh T degree {deg}: max rel err {e:.3e} ({e / 2**-24:.3f} x 2^-24)")
        for i, ci in enumerate(c):
            print(f"   t{i} = {cfloat(ci)}")

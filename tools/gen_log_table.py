#!/usr/bin/env python3
"""Generate the 32-entry table and the polynomial of fg_log / fg_log1p (feedback_gnn_amd/csrc/fgnn_math.h).

    log(x) = e*ln2 + LC[j] + log1p(r),   r = m*RC[j] - 1,   x = 2^e * m,  m in [0.7109375, 1.421875)

The grid: m is rounded to the nearest point of the float grid with 5 mantissa bits; points below 1 are 1/64 apart (46/64 .. 63/64,
j = 0..17), points from 1 on are 1/32 apart (32/32 .. 45/32, j = 18..31), so |r| <= 1/64.  RC[j] is a float near 1/c_j chosen so
that LC[j] = -log(RC[j]) is a float to within 2^-10 ulp ("accurate tables": no low word is needed, and c = 1 has RC = 1, LC = 0
exactly, which keeps full relative accuracy around x = 1).  The 32 entries of each array sit in 32 consecutive LDS banks: a wave's
lookup is conflict-free whatever the indices.

    python tools/gen_log_table.py      prints the C initialisers and the polynomial
"""
import struct

import mpmath as mp
import numpy as np

mp.mp.prec = 200


def f32(x):
    return struct.unpack("f", struct.pack("f", float(x)))[0]


def bits(x):
    return struct.unpack("I", struct.pack("f", x))[0]


def from_bits(b):
    return struct.unpack("f", struct.pack("I", b))[0]


def centres():
    return [mp.mpf(46 + j) / 64 if j < 18 else 1 + mp.mpf(j - 18) / 32 for j in range(32)]


def table(search=4096):
    rc, lc, miss = [], [], []
    for j, c in enumerate(centres()):
        if j == 18:
            rc.append(1.0); lc.append(0.0); miss.append(0.0)
            continue
        b0 = bits(f32(1 / c))
        best = None
        for d in range(-search, search + 1):
            cand = from_bits(b0 + d)
            exact = -mp.log(mp.mpf(cand))
            fl = f32(exact)
            ulp = mp.mpf(2) ** (mp.floor(mp.log(abs(mp.mpf(fl)), 2)) - 23)
            err = abs((exact - mp.mpf(fl)) / ulp)
            if best is None or err < best[0]:
                best = (err, cand, fl)
        miss.append(float(best[0])); rc.append(best[1]); lc.append(best[2])
    return rc, lc, miss


def fit_poly(rc, deg=3, N=3001):
    """log1p(r) ~ r + r^2 (c0 + c1 r + ... ), relative error in log1p(r), over the r-range the table produces."""
    cs = centres()
    rmax = 0.0
    for j, c in enumerate(cs):
        half = mp.mpf(1) / 128 if j < 18 else mp.mpf(1) / 64
        for m in (c - half, c + half):
            rmax = max(rmax, abs(float(m * mp.mpf(rc[j]) - 1)))
    rmax *= 1.0001
    k = np.arange(N)
    xs = rmax * np.cos(np.pi * (k + 0.5) / N)
    xs = xs[np.abs(xs) > 1e-6]
    tgt = np.array([float((mp.log1p(mp.mpf(x)) - mp.mpf(x)) / mp.mpf(x) ** 2) for x in xs])
    A = np.stack([xs ** i for i in range(deg)], axis=1)
    wgt = xs ** 2 / np.abs(np.array([float(mp.log1p(mp.mpf(x))) for x in xs]))
    w = np.ones_like(xs)
    for _ in range(80):
        sw = np.sqrt(w) * wgt
        c, *_ = np.linalg.lstsq(A * sw[:, None], tgt * sw, rcond=None)
        err = np.abs(wgt * (A @ c - tgt))
        w = w * (err + 1e-300); w /= w.sum()
    return [f32(v) for v in c], float(err.max()), rmax


if __name__ == "__main__":
    rc, lc, miss = table()
    print("/* RC */", ", ".join(f"{v:.9e}f" for v in rc))
    print("/* LC */", ", ".join(f"{v:.9e}f" for v in lc))
    print("worst LC representation error: %.5f ulp" % max(miss))
    for deg in (2, 3, 4):
        c, e, rmax = fit_poly(rc, deg)
        print(f"deg {deg}: rmax {rmax:.6f} coeffs", ", ".join(f"{v:.9e}f" for v in c), " max rel err %.3e (%.4f ulp)" % (e, e / 2 ** -24))

"""GF(2) linear algebra for CSS code construction (host side, runs once at start-up).

Mirrors the interface of the reference helpers in /root/reference sionna/fec/utils.py:1022-1228
(`row_echelon`, `rank`, `kernel`, `row_basis`, `compute_code_distance`, `inverse`) and `int2bin`
(:714).  The elimination order — scan columns left to right, pivot = first row at or below the
current pivot row that holds a 1, clear every other 1 in that column below (or everywhere, if
``reduced``) — decides which basis of the kernel comes out, and therefore the row order of
``hx_perp`` / ``hz_perp`` and the logical operators the rest of the pipeline sees.  The
implementation here keeps that order (checked against the reference's own functions by
tests/golden/make_golden_codes.py) but works on whole row blocks at a time instead of Python
loops over rows.
"""
import numpy as np


def row_echelon(mat, reduced=False):
    """Gaussian elimination over GF(2) without column swaps.

    Returns ``[echelon_form, rank, transform, pivot_cols]`` with
    ``transform @ mat % 2 == echelon_form``.  Rank-deficient and over-complete inputs are fine.
    """
    a = np.array(mat, dtype=bool, copy=True)
    rows, cols = a.shape
    # carry the row operations on an identity glued to the right of the matrix
    aug = np.concatenate([a, np.eye(rows, dtype=bool)], axis=1)
    pivots = []
    r = 0
    for c in range(cols):
        if r >= rows:
            break
        below = np.flatnonzero(aug[r:, c])
        if below.size == 0:
            continue
        first = r + below[0]
        if first != r:
            aug[[r, first]] = aug[[first, r]]
        if reduced:
            hit = np.flatnonzero(aug[:, c])
            hit = hit[hit != r]
        else:
            hit = r + 1 + np.flatnonzero(aug[r + 1:, c])
        if hit.size:
            aug[hit] ^= aug[r]
        pivots.append(c)
        r += 1
    return [aug[:, :cols].astype(int), r, aug[:, cols:].astype(int), pivots]


def rank(mat):
    """Rank of a binary matrix."""
    return row_echelon(mat)[1]


def kernel(mat):
    """Basis of {x : mat @ x = 0 (mod 2)} as rows, plus the rank and the pivot list of mat.T.

    The rows of the transform below the rank of ``mat.T`` annihilate ``mat.T``'s row space.
    """
    t = np.asarray(mat).T
    _, rk, transform, pivots = row_echelon(t)
    return transform[rk:t.shape[0]], rk, pivots


def row_basis(mat):
    """A maximal set of linearly independent rows of ``mat`` (in their original order)."""
    mat = np.asarray(mat)
    return mat[row_echelon(mat.T)[3]]


def compute_code_distance(mat, is_pcm=True, is_basis=False):
    """Minimum row weight of a basis of the code — the quantity the reference calls "distance"
    (an upper bound on it, not the true minimum distance; see codes_q.py:47 in the reference)."""
    gen = np.asarray(mat)
    if is_pcm:
        gen = kernel(gen)[0]
    if len(gen) == 0:
        return np.inf
    cw = gen if is_basis else row_basis(gen)
    return np.min(np.sum(cw, axis=1))


def inverse(mat):
    """Inverse of a square full-rank matrix, or left inverse of a full-column-rank one."""
    mat = np.asarray(mat)
    m, n = mat.shape
    ech, rk, transform, _ = row_echelon(mat, reduced=True)
    if m == n and rk == m:
        return transform
    if m > rk and n == rk:
        return ech.T @ transform % 2
    raise ValueError("This matrix is not invertible. Please provide either a full-rank square "
                     "matrix or a rectangular matrix with full column rank.")


def int2bin(num, len_):
    """``num`` as a list of ``len_`` bits, most significant first (int2bin(5, 4) == [0,1,0,1])."""
    assert num >= 0 and len_ >= 0
    return [(num >> (len_ - 1 - i)) & 1 for i in range(len_)]


def int_mod_2(x):
    """x mod 2 for integer arrays/tensors (sionna/fec/utils.py:1565-1582)."""
    return x & 1

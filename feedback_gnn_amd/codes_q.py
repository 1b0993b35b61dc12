"""CSS quantum code objects and constructions (host side, NumPy only).

Public names and results equal those of /root/reference sionna/fec/ldpc/codes_q.py, so the entry scripts
(`n882.py:34`, `n1270.py:37`) and notebooks keep working:

    css_code, create_circulant_matrix, create_generalized_bicycle_codes, hypergraph_product, hamming_code, rep_code,
    create_surface_codes, create_rotated_surface_codes, create_checkerboard_toric_codes, create_QC_GHP_codes,
    create_cyclic_permuting_matrix, create_bivariate_QC_codes, readAlist, alistToNumpy

The matrices (hx, hz, the kernel bases hx_perp / hz_perp in the reference's row order, the logical operators) are pinned to
the reference's own output by tests/golden/codes.npz, which tests/golden/make_golden_codes.py produces by running the
reference's NumPy-only functions.  Device graph tables are derived from these matrices by `feedback_gnn_amd.graph`.
"""
import numpy as np

from . import gf2


def _binary_matrix(rows, n_rows, n_cols):
    """0/1 int matrix from a list of column-index lists (one list per row)."""
    out = np.zeros((n_rows, n_cols), dtype=int)
    for r, cols in enumerate(rows):
        out[r, cols] = 1
    return out


class css_code:
    """CSS code defined by an X-type and a Z-type parity-check matrix.

    After construction the object carries what the decoders and the evaluation models read
    (reference codes_q.py:8-49): ``hx, hz`` · ``hx_perp, hz_perp`` (bases of ker hx / ker hz) · ``hx_basis, hz_basis`` with
    ``pivot_hx, pivot_hz`` (independent row subsets) · ``rank_hx, rank_hz`` · ``lx, lz`` (logical operators) ·
    ``N, K`` · ``L, Q`` (largest column / row weight) · ``D`` (smallest row weight of the two kernel bases — an upper bound
    the reference also stores, not the true distance) · ``name``.
    """

    def __init__(self, hx=np.array([[]]), hz=np.array([[]]), code_distance=np.nan, name=None, name_prefix="", check_css=False):
        if hx.shape[1] != hz.shape[1]:
            raise AssertionError("hx and hz should have equal number of columns!")
        if hx.shape[1] == 0:
            raise AssertionError("number of variable nodes should not be zero!")
        if check_css and np.any(hx @ hz.T % 2):
            raise AssertionError("CSS constraint not satisfied")
        self.hx, self.hz = hx, hz
        self.N = hx.shape[1]
        # a single elimination of h^T yields the kernel, the rank and a set of independent rows of h
        for tag, h in (("hx", hx), ("hz", hz)):
            ker, rk, pivots = gf2.kernel(h)
            setattr(self, tag + "_perp", ker)
            setattr(self, "rank_" + tag, rk)
            setattr(self, "pivot_" + tag, pivots)
            setattr(self, tag + "_basis", h[pivots])
        self.K = self.N - self.rank_hx - self.rank_hz
        self.compute_ldpc_params()
        self.lx = self.lz = np.array([[]])
        self.compute_logicals()
        self.D = code_distance
        if code_distance is np.nan:
            self.D = np.min([gf2.compute_code_distance(k, is_pcm=False, is_basis=True) for k in (self.hx_perp, self.hz_perp)])
        self.name = name if name is not None else f"{name_prefix}_n{self.N}_k{self.K}"

    def compute_ldpc_params(self):
        col_w = [int(h.sum(axis=0).max()) for h in (self.hx, self.hz)]
        row_w = [int(h.sum(axis=1).max()) for h in (self.hx, self.hz)]
        self.L = np.max(col_w).astype(int)
        self.Q = np.max(row_w).astype(int)

    def compute_logicals(self):
        """lx: vectors of ker(hz) outside the row space of hx; lz: vectors of ker(hx) outside the row space of hz.
        Stacking [row-space basis; kernel basis] and eliminating its transpose marks the kernel rows that add rank."""

        def complement(row_space, ker):
            stacked = np.vstack([row_space, ker])
            adds_rank = set(gf2.row_echelon(stacked.T)[3])
            first = row_space.shape[0]
            return stacked[[i for i in range(first, stacked.shape[0]) if i in adds_rank]]

        self.lx = complement(self.hx_basis, self.hz_perp)
        self.lz = complement(self.hz_basis, self.hx_perp)
        return self.lx, self.lz

    def canonical_logicals(self):
        """Re-pair the X logicals so that lx @ lz.T is the identity."""
        self.lx = gf2.inverse(self.lx @ self.lz.T % 2) @ self.lx % 2


# ---------------------------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------------------------
def create_circulant_matrix(l, pows):
    """l x l circulant with ones at (i + c mod l, i) for every shift c in ``pows``."""
    i = np.arange(l)
    h = np.zeros((l, l), dtype=int)
    for shift in pows:
        h[(i + shift) % l, i] = 1
    return h


def create_cyclic_permuting_matrix(n, shifts):
    """n x n table of circulant shifts: shifts[i] sits on the i-th wrapped sub-diagonal, every other entry is -1."""
    table = np.full((n, n), -1, dtype=int)
    rows = np.arange(n)
    for i, s in enumerate(shifts):
        table[rows, (rows - i) % n] = s
    return table


def rep_code(d):
    """(d-1) x d parity-check matrix of the length-d repetition code."""
    return _binary_matrix([[i, i + 1] for i in range(d - 1)], d - 1, d)


def hamming_code(rank):
    """r x (2^r - 1) Hamming parity-check matrix whose column i is the binary expansion of i+1."""
    r = int(rank)
    return np.array([gf2.int2bin(value, r) for value in range(1, 2 ** r)], dtype=int).T


# ---------------------------------------------------------------------------------------------------------------
# code families
# ---------------------------------------------------------------------------------------------------------------
def create_generalized_bicycle_codes(l, a, b, name=None):
    """GB code from two circulants: hx = [A | B], hz = [B^T | A^T]."""
    A, B = create_circulant_matrix(l, a), create_circulant_matrix(l, b)
    return css_code(np.hstack((A, B)), np.hstack((B.T, A.T)), name=name, name_prefix="GB")


def hypergraph_product(h1, h2, name=None):
    """HGP code: hx = [h1 (x) I | I (x) h2^T], hz = [I (x) h2 | h1^T (x) I]."""
    h1, h2 = np.asarray(h1, dtype=int), np.asarray(h2, dtype=int)
    (m1, n1), (m2, n2) = h1.shape, h2.shape
    eye = lambda k: np.eye(k, dtype=int)  # noqa: E731
    hx = np.hstack([np.kron(h1, eye(n2)), np.kron(eye(m1), h2.T)])
    hz = np.hstack([np.kron(eye(n1), h2), np.kron(h1.T, eye(m2))])
    return css_code(hx, hz, name=name, name_prefix="HP")


def create_surface_codes(n):
    """[[n^2+(n-1)^2, 1, n]] surface code = HGP of two repetition codes."""
    h = rep_code(n)
    return hypergraph_product(h, h, f"Surface_n{n**2 + (n-1)**2}_k{1}_d{n}")


def _plaquette(n, i, j):
    """Qubits of the 2x2 plaquette whose top-left corner is (i, j) on an n x n grid with wrap-around."""
    i2, j2 = (i + 1) % n, (j + 1) % n
    return [i * n + j, i2 * n + j2, i2 * n + j, i * n + j2]


def set_pcm_row(n, pcm, row_idx, i, j):
    """Set, in row ``row_idx`` of ``pcm``, the four qubits of the plaquette with top-left corner (i, j) (codes_q.py:147-150 of the
    reference: the helper its surface / toric constructions call; kept for scripts that build their own matrices with it)."""
    for q in _plaquette(n, i, j):
        pcm[row_idx][q] = 1


def create_rotated_surface_codes(n, name=None):
    """[[n^2, 1, n]] rotated surface code, n odd: bulk plaquettes alternate Z (even i+j) / X, weight-2 X checks on the
    top (even columns) and bottom (odd columns) edge, weight-2 Z checks on the right (even rows) and left (odd rows) edge."""
    if n % 2 != 1:
        raise AssertionError("n should be odd")
    x_rows, z_rows = [], []
    for i in range(n - 1):
        for j in range(n - 1):
            (x_rows if (i + j) % 2 else z_rows).append(_plaquette(n, i, j))
    for j in range(n - 1):
        start = j if j % 2 == 0 else (n - 1) * n + j
        x_rows.append([start, start + 1])
    for i in range(n - 1):
        col = n - 1 if i % 2 == 0 else 0
        z_rows.append([i * n + col, (i + 1) * n + col])
    m = (n * n - 1) // 2
    return css_code(_binary_matrix(x_rows, m, n * n), _binary_matrix(z_rows, m, n * n), name=name, name_prefix="Rotated_Surface")


def create_checkerboard_toric_codes(n, name=None):
    """[[n^2, 2]] toric code on an n x n checkerboard, n even: every plaquette is a check, Z on even i+j."""
    if n % 2 != 0:
        raise AssertionError("n should be even")
    x_rows, z_rows = [], []
    for i in range(n):
        for j in range(n):
            (x_rows if (i + j) % 2 else z_rows).append(_plaquette(n, i, j))
    return css_code(_binary_matrix(x_rows, n * n // 2, n * n), _binary_matrix(z_rows, n * n // 2, n * n), name=name,
                    name_prefix="Toric")


def create_QC_GHP_codes(l, a, b, name=None):
    """Quasi-cyclic generalized hypergraph product code (the [[882,24]] and [[1270,28]] benchmark codes).

    ``a``: m x n table of circulant shifts (-1 = zero block), ``b``: shifts of one circulant C.
    A = [P^{a_ij}], hx = [A | I_m (x) C], hz = [I_n (x) C^T | A^T].
    """
    a = np.asarray(a)
    m, n = a.shape
    A = np.zeros((m * l, n * l), dtype=int)
    for (i, j), shift in np.ndenumerate(a):
        if shift >= 0:
            A[i * l:(i + 1) * l, j * l:(j + 1) * l] = create_circulant_matrix(l, [shift])
    C = create_circulant_matrix(l, b)
    hx = np.hstack((A, np.kron(np.eye(m, dtype=int), C)))
    hz = np.hstack((np.kron(np.eye(n, dtype=int), C.T), A.T))
    return css_code(hx, hz, name=name, name_prefix="GHP")


def create_bivariate_QC_codes(l, m, A_x_pows, A_y_pows, B_x_pows, B_y_pows, name=None):
    """Bivariate bicycle ("IBM") codes: x = S_l (x) I_m, y = I_l (x) S_m, A and B sums of monomials in x and y."""
    x = np.kron(create_circulant_matrix(l, [-1]), np.eye(m, dtype=int))
    y = np.kron(np.eye(l, dtype=int), create_circulant_matrix(m, [-1]))

    def polynomial(x_pows, y_pows):
        monomials = [np.linalg.matrix_power(x, e) for e in x_pows] + [np.linalg.matrix_power(y, e) for e in y_pows]
        total = monomials[0]
        for mono in monomials[1:]:
            total = total + mono
        return total

    A, B = polynomial(A_x_pows, A_y_pows), polynomial(B_x_pows, B_y_pows)
    return css_code(np.hstack((A, B)), np.hstack((B.T, A.T)), name=name, name_prefix="IBM")


# ---------------------------------------------------------------------------------------------------------------
# A-list files (over-complete check matrices)
# ---------------------------------------------------------------------------------------------------------------
def alistToNumpy(lines):
    """Parsed A-list (a list of integer lists) -> dense float 0/1 matrix of shape [rows, cols].  The optional two degree
    lines after the header are skipped when present; entries equal to 0 are padding."""
    n_cols, n_rows = lines[0]
    has_degree_lines = len(lines[2]) == n_cols and len(lines[3]) == n_rows
    body = lines[4 if has_degree_lines else 2:]
    mat = np.zeros((n_rows, n_cols), dtype=float)
    for col in range(n_cols):
        for one_based_row in body[col]:
            if one_based_row:
                mat[one_based_row - 1, col] = 1
    return mat


def readAlist(directory):
    """Read a parity-check matrix from an A-list text file; returns an int 0/1 array."""
    with open(directory, "r") as handle:
        parsed = [[int(tok) for tok in line.rstrip().split(" ")] for line in handle]
    return alistToNumpy(parsed).astype(int)

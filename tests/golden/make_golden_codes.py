#!/usr/bin/env python3
"""Generate tests/golden/codes.npz by EXECUTING the reference's NumPy-only code.

Runs only in the build container (needs /root/reference).  TensorFlow is not installed there, and
`sionna/fec/utils.py` imports it at module level, so the functions that are pure NumPy are pulled
out by AST and exec'd in a scratch namespace; `sionna/fec/ldpc/codes_q.py` is exec'd whole with its
one `sionna` import line replaced by that namespace.  Nothing of the reference's text is written to
the fixture: the .npz holds only matrices (bit-packed) and scalars.

    python tests/golden/make_golden_codes.py
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_namespace():
    utils_src = open(f"{REF}/sionna/fec/utils.py").read()
    want = {"row_echelon", "rank", "kernel", "row_basis", "compute_code_distance", "inverse", "int2bin"}
    tree = ast.parse(utils_src)
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(ast.Module([node], []), "ref_utils", "exec"), ns)
    codes_src = open(f"{REF}/sionna/fec/ldpc/codes_q.py").read()
    codes_src = "\n".join(l for l in codes_src.splitlines() if not l.startswith("from sionna."))
    exec(compile(codes_src, "ref_codes_q", "exec"), ns)
    return ns


def pack(a):
    a = np.asarray(a).astype(np.uint8)
    return np.packbits(a, axis=1), np.array(a.shape, dtype=np.int64)


def main():
    R = load_reference_namespace()
    cases = {
        "steane": lambda: R["css_code"](R["hamming_code"](3), R["hamming_code"](3), name="Steane_n7_k1_d3"),
        "rsurf3": lambda: R["create_rotated_surface_codes"](3),
        "rsurf5": lambda: R["create_rotated_surface_codes"](5),
        "surf3": lambda: R["create_surface_codes"](3),
        "toric4": lambda: R["create_checkerboard_toric_codes"](4),
        "gb48": lambda: R["create_generalized_bicycle_codes"](24, [0, 2, 8, 15], [0, 2, 12, 17], name="GB_n48_k6_d8"),
        "gb126": lambda: R["create_generalized_bicycle_codes"](63, [0, 1, 14, 16, 22], [0, 3, 13, 20, 42]),
        "gb254": lambda: R["create_generalized_bicycle_codes"](127, [0, 15, 20, 28, 66], [0, 58, 59, 100, 121]),
        "hp_c7": lambda: R["hypergraph_product"](R["create_circulant_matrix"](7, [0, 1, 3]),
                                                R["create_circulant_matrix"](7, [0, 1, 3])),
        "ibm72": lambda: R["create_bivariate_QC_codes"](6, 6, [3], [1, 2], [1, 2], [3]),
        "ghp882": lambda: R["create_QC_GHP_codes"](63, R["create_cyclic_permuting_matrix"](7, [27, 54, 0]), [0, 1, 6]),
        "ghp1270": lambda: R["create_QC_GHP_codes"](
            127, np.array([[0, -1, 51, 52, -1], [-1, 0, -1, 111, 20], [0, -1, 98, -1, 122], [0, 80, -1, 119, -1],
                           [-1, 0, 5, -1, 106]]), [0, 1, 7], name="GHP_n1270_k28"),
    }
    out = {}
    for key, make in cases.items():
        c = make()
        for attr in ("hx", "hz", "hx_perp", "hz_perp", "lx", "lz"):
            bits, shape = pack(getattr(c, attr))
            out[f"{key}/{attr}"] = bits
            out[f"{key}/{attr}_shape"] = shape
        out[f"{key}/scalars"] = np.array([c.N, c.K, int(c.D), int(c.L), int(c.Q), c.rank_hx, c.rank_hz], dtype=np.int64)
        out[f"{key}/name"] = np.array(c.name)
        print(f"{key:8s} {c.name:28s} N={c.N} K={c.K} D={c.D} L={c.L} Q={c.Q} rank=({c.rank_hx},{c.rank_hz}) "
              f"perp=({c.hx_perp.shape[0]},{c.hz_perp.shape[0]})")
    # a reduced row echelon + inverse case for gf2.inverse / row_echelon(reduced=True)
    rng = np.random.RandomState(7)
    m = rng.randint(0, 2, size=(12, 9))
    ech, rk, tr, piv = R["row_echelon"](m, reduced=True)
    out["rre/in"], out["rre/ech"], out["rre/tr"] = m, ech, tr
    out["rre/rank_piv"] = np.array([rk] + list(piv), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "codes.npz"), **out)
    print("wrote", os.path.join(HERE, "codes.npz"), os.path.getsize(os.path.join(HERE, "codes.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())

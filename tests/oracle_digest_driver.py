"""Digest of what the CPU oracle computes — run by tests/test_oracle_compilers.py once per host compiler (FGNN_ORACLE_LIB_PATH selects the
build): sha256 over (a) the exhaustive-probe checksums of every shared-math / RNG routine on three 2^22-input slices of each of its
domain ranges and (b) the complete outputs of every decode path on seeded inputs.  Prints one JSON object {name: hexdigest}."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from feedback_gnn_amd.weights_io import read_weight_list  # noqa: E402
from test_gpu_math_bits import DOMAINS  # noqa: E402

SEED = 0x5EED


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


out = {}
for name, ranges in DOMAINS.items():
    parts = []
    for lo, hi in ranges:
        span = min(1 << 22, hi - lo + 1)
        mid = lo + (hi - lo) // 2
        for a in sorted({lo, min(mid, hi - span + 1), hi - span + 1}):
            parts.append(O.math_checksums(name, a, a + span - 1, 22))
    out["math:" + name] = sha(*parts)
for name, B in (("ghp882", 24), ("gb48", 40), ("rsurf5", 40)):
    g = H.oracle_literal_forms(name)
    ex, ez = g.pauli_noise(SEED, 0.07, 5, B)
    sx, sz = g.syndrome(ex, ez)
    L0 = H.llr_const(0.05)
    out[f"noise:{name}"] = sha(ex, ez, sx, sz)
    for cn in ("boxplus-phi", "minsum", "boxplus"):
        for lse in (0, 1):
            g.set_vn_shared_lse(lse)
            o = g.bp4_decode(sx, sz, 12, cn, 0.8, llr_const=L0)
            out[f"bp4:{name}:{cn}:lse{lse}"] = sha(o["llr"], o["x_hat"], o["z_hat"], o["x_logit"], o["z_logit"])
    if name == "ghp882":
        w = read_weight_list(H.WEIGHTS_882)
        for order in (0, 1):
            g.set_gnn_order(order)
            s = g.sandwich_decode(sx, sz, [16, 8, 4], [w, w], L0, return_llr=True)
            out[f"sandwich:{name}:order{order}"] = sha(s["x_hat"], s["z_hat"], s["llr"], s["rounds"])
print(json.dumps(out))

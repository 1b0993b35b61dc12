"""Drive every entry point of the CPU oracle on small inputs — run by tests/test_oracle_sanitizers.py in a subprocess whose oracle
library was compiled with -fsanitize=address,undefined (LD_PRELOAD of the ASan runtime, FGNN_ORACLE_LIB_PATH pointing at that build).
Any out-of-bounds access, use of an uninitialised stack slot that ASan can see, signed overflow or misaligned access aborts the
process; on success the last line printed is "oracle sanitizers: clean"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import numpy_ref as NR  # noqa: E402
from oracle import oracle as O  # noqa: E402
from feedback_gnn_amd.weights_io import read_weight_list  # noqa: E402

SEED = 0x5EED
rng = np.random.RandomState(0)


def rnd(shapes):
    return [rng.uniform(-0.4, 0.4, size=s).astype(np.float32) for s in shapes]


assert "asan" in os.environ.get("FGNN_ORACLE_LIB_PATH", "")
O.math_apply("tanh", np.linspace(-12, 12, 1001).astype(np.float32))
O.math_apply("phi", np.linspace(0, 20, 1001).astype(np.float32))
O.philox([1, 2, 3, 4], [5, 6])
for name in ("steane", "rsurf3", "rsurf5", "toric4", "gb48", "gb48_oc", "ghp882"):
    c, g = H.code(name), H.oracle_library_forms(name)
    B = 3 if name == "ghp882" else 7
    ex, ez = g.pauli_noise(SEED, 0.06, 11, B)
    g.pauli_noise_wt(SEED, min(4, g.n), 0, B)
    sx, sz = g.syndrome(ex, ez)
    L0 = H.llr_const(0.05)
    for cn in ("boxplus-phi", "minsum", "boxplus"):
        for lse in (0, 1):
            g.set_vn_shared_lse(lse)
            o = g.bp4_decode(sx, sz, 5, cn, 0.8, llr_const=L0, return_msgs=True)
            g.bp4_decode(sx, sz, 2, cn, 1.0, llr_ch=o["llr"], msg_init=(o["msg_x"], o["msg_z"]))
    g.set_vn_shared_lse(0)  # the library default
    g.residual(ex, ez, o["x_hat"], o["z_hat"])
    e = g.bsc_noise(SEED, 0.05, 0, B)
    synd = ((e.astype(np.int64) @ np.asarray(c.hx, dtype=np.int64).T) % 2).astype(np.uint8)
    for cn in ("boxplus-phi", "minsum", "boxplus"):
        g.bp2_decode(synd, 4, cn, 0.9, llr_const=-1.3)
    if name not in ("gb48_oc",):
        w = read_weight_list(H.WEIGHTS_882)
        for order in (0, 1):
            g.set_gnn_order(order)
            g.feedback_gnn(w, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
            g.sandwich_decode(sx, sz, [4, 3, 2], [w, w], L0, return_llr=True)
            g.gnn_bp4(rnd(NR.gnn_bp4_general_shapes(c, (20, 40, 2, 1, 1, 1, 0, 0, 0))), sx, sz, 2)
        g.set_gnn_order(0)  # the library default
        g.feedback_gnn_general((8, 16, 3, 2, 2, 0), rnd([(16, 3)] + [(4, 16), (16, 16), (16, 8)] * 2 + [(19, 16), (16, 16)]),
                               o["llr"], o["z_logit"], o["x_logit"], sx, sz)
        for cfg in ((12, 24, 3, 0, 3, 1, 1, 3, 2), (32, 96, 4, 2, 1, 0, 1, 16, 16), (5, 7, 1, 3, 0, 1, 0, 0, 0)):
            g.gnn_bp4_general(cfg, rnd(NR.gnn_bp4_general_shapes(c, cfg)), sx, sz, 2)
    if hasattr(c, "pivot_hx") and name in ("gb48", "ghp882"):
        idx = np.arange(B, dtype=np.int32)
        g.osd0(0, c.pivot_hx, sx, marg=o["llr"], index=idx)
        g.osd0(1, c.pivot_hz, sz, marg=o["llr"], index=idx)
    print(name, "ok", flush=True)
print("oracle sanitizers: clean")

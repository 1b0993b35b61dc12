"""BP4-64 with the channel LLRs as a constant (no LDS copy, 7 workgroups per CU) vs as a per-qubit tensor holding the same constant
(LDS copy, 5 workgroups per CU): what the second decoder of a sandwich pays per iteration.  python tools/ab_bp4_lch.py"""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
g = TannerGraph(code('ghp882'))
B = 65536
L0 = llr_const(0.05)
tag = os.path.basename(os.environ.get("FGNN_LIB_PATH", "current"))
ex, ez = g.pauli_noise(0x5EED, 0.01, 0, B); sx, sz = g.syndrome(ex, ez)
lch = torch.full((B, 3, g.n), L0, dtype=torch.float32, device=g.device)
g.set_saturation_shortcut(False)
for its in (64, 16):
    for name, kw in (("const", dict(llr_const=L0)), ("tensor", dict(llr_ch=lch))):
        g.bp4_decode(sx, sz, its, "boxplus-phi", 1.0, **kw)
        g.profile_enable(8)
        for _ in range(4): g.bp4_decode(sx, sz, its, "boxplus-phi", 1.0, **kw)
        torch.cuda.synchronize()
        ms = [r[0] for r in g.profile_read()]
        print(f"[{tag}] BP4-{its} llr {name}: {sum(ms)/len(ms):.3f} ms  ({sum(ms)/len(ms)/its:.4f} ms/iteration)", flush=True)

"""Kernel-only time (HIP events on the launch stream) of BP4-64: python tools/ab_bp4.py [code]; FGNN_LIB_PATH selects the build."""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
name = sys.argv[1] if len(sys.argv) > 1 else 'ghp882'
g = TannerGraph(code(name))
B = 65536 if name == 'ghp882' else 32768
L0 = llr_const(0.05)
tag = os.path.basename(os.environ.get("FGNN_LIB_PATH", "current"))
for p in (0.01, 0.10):
    ex, ez = g.pauli_noise(0x5EED, p, 0, B); sx, sz = g.syndrome(ex, ez)
    for on in (False, True):
        g.set_saturation_shortcut(on)
        g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0)
        g.profile_enable(8)
        for _ in range(5): g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0)
        torch.cuda.synchronize()
        ms = [r[0] if isinstance(r, (tuple, list)) else r for r in g.profile_read()]
        print(f"[{tag}] {name} p={p} shortcut={on}: {ms}", flush=True)

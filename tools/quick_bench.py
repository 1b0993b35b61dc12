"""Ad-hoc timing sweep (not the contract bench): python tools/quick_bench.py"""
import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const, WEIGHTS_882
from feedback_gnn_amd.graph import TannerGraph, GnnWeights
from feedback_gnn_amd.weights_io import read_weight_list
g = TannerGraph(code('ghp882'))
print(g.info())
B = 65536
ex, ez = g.pauli_noise(0x5EED, 0.01, 0, B)
sx, sz = g.syndrome(ex, ez)
L0 = llr_const(0.05)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps
for tpc, cpb in [(448, 1), (128, 1), (192, 1), (256, 1), (384, 1), (512, 1), (128, 2), (1024, 1), (256, 2)]:
    try:
        g.set_launch(tpc, cpb)
        dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0))
        print(f"BP64 phi regular tpc={tpc} cpb={cpb}: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
    except Exception as e: print(tpc, cpb, e)
g.set_launch(256, 1); g.force_generic(True)
dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0)); print(f"generic tpc=256: {dt*1e3:.1f} ms")
g.force_generic(False)
for cn in ("minsum", "boxplus"):
    dt = timeit(lambda: g.bp4_decode(sx, sz, 64, cn, 1.0, llr_const=L0))
    print(f"BP64 {cn} tpc=256: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
w = GnnWeights(read_weight_list(WEIGHTS_882), g.device)
o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0)
for tpc in (256, 448, 512):
    g.set_launch(tpc, 1)
    dt = timeit(lambda: g.feedback_gnn(w, o['llr'], o['z_logit'], o['x_logit'], sx, sz))
    print(f"GNN tpc={tpc}: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
    dt = timeit(lambda: g.bp4_decode(sx, sz, 16, "boxplus-phi", 1.0, llr_ch=o['llr']))
    print(f"BP16 with llr_ch tpc={tpc}: {dt*1e3:.1f} ms")
g.set_launch(0, 0)
for p in (0.01, 0.05, 0.10):
    ex, ez = g.pauli_noise(0x5EED, p, 0, B); sx, sz = g.syndrome(ex, ez)
    for on in (False, True):
        g.set_saturation_shortcut(on)
        dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0))
        print(f"BP64 p={p} shortcut={on}: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")

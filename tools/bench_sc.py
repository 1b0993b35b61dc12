"""BP4-64 and the sandwich with the exact optimisations on/off at several p (python tools/bench_sc.py [code])."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const, WEIGHTS_882, WEIGHTS_1270
from feedback_gnn_amd.graph import TannerGraph, GnnWeights
from feedback_gnn_amd.weights_io import read_weight_list
name = sys.argv[1] if len(sys.argv) > 1 else 'ghp882'
g = TannerGraph(code(name))
B = 65536 if name == 'ghp882' else 32768
L0 = llr_const(0.05)
w = GnnWeights(read_weight_list(WEIGHTS_882 if name == 'ghp882' else WEIGHTS_1270), g.device)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps
for p in (0.01, 0.05, 0.10):
    ex, ez = g.pauli_noise(0x5EED, p, 0, B); sx, sz = g.syndrome(ex, ez)
    for sc, fpe in ((False, False), (True, False), (True, True)):
        g.set_saturation_shortcut(sc); g.set_fixed_point_exit(fpe)
        dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0))
        ds = timeit(lambda: g.sandwich_decode(sx, sz, [64, 16], [w], L0, compact=True))
        print(f"{name} p={p} shortcut={sc} fixed_point_exit={fpe}: BP64 {dt*1e3:6.2f} ms {B/dt/1e6:6.2f} M cw/s | "
              f"compacted sandwich (64,G,16) {ds*1e3:6.2f} ms {B/ds/1e6:6.2f} M cw/s", flush=True)

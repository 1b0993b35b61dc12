"""BP4-64 with/without the saturation shortcut at several p (python tools/bench_sc.py)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
name = sys.argv[1] if len(sys.argv) > 1 else 'ghp882'
g = TannerGraph(code(name))
B = 65536 if name == 'ghp882' else 32768
L0 = llr_const(0.05)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps
for p in (0.01, 0.05, 0.10):
    ex, ez = g.pauli_noise(0x5EED, p, 0, B); sx, sz = g.syndrome(ex, ez)
    for on in (False, True):
        g.set_saturation_shortcut(on)
        dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0))
        print(f"{name} BP64 p={p} shortcut={on}: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s", flush=True)
g.set_saturation_shortcut(False)
for cn in ("minsum", "boxplus"):
    dt = timeit(lambda: g.bp4_decode(sx, sz, 64, cn, 1.0, llr_const=L0))
    print(f"BP64 {cn}: {dt*1e3:.1f} ms")

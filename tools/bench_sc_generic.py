"""BP4-64 on codes that take the runtime-degree kernel, exact optimisations on/off (python tools/bench_sc_generic.py)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
for name, B, p0 in (("gb254", 131072, 0.1), ("gb126", 262144, 0.1), ("hp_c7", 262144, 0.1)):
    g = TannerGraph(code(name))
    L0 = llr_const(p0)
    def timeit(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t = time.time()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.time() - t) / reps
    for launch in (None, (256, 1)):
        if launch: g.set_launch(*launch)
        for p in (0.01, 0.05):
            ex, ez = g.pauli_noise(0x5EED, p, 0, B); sx, sz = g.syndrome(ex, ez)
            res = []
            for sc, fpe in ((False, False), (True, False), (True, True)):
                g.set_saturation_shortcut(sc); g.set_fixed_point_exit(fpe)
                dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 0.8, llr_const=L0))
                res.append(f"{dt*1e3:7.2f} ms ({B/dt/1e6:5.2f} M/s)")
            print(f"{name} launch={launch or g.info()['threads_per_codeword']} p={p}: off {res[0]} | shortcut {res[1]} | +exit {res[2]}", flush=True)

import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
g = TannerGraph(code('ghp1270')); B = 32768
ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B); sx, sz = g.syndrome(ex, ez)
g.set_saturation_shortcut(False)
def timeit(fn, reps=2):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.time() - t) / reps
for tpc in (128, 192, 256, 320, 384, 512, 640, 768, 1024):
    g.set_launch(tpc, 1)
    dt = timeit(lambda: g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05)))
    print(f"ghp1270 BP64 tpc={tpc}: {dt*1e3:.1f} ms {B/dt/1e3:.0f} k cw/s")

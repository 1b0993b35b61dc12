import sys, time, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import code, llr_const, WEIGHTS_882
from feedback_gnn_amd.graph import TannerGraph, GnnWeights
from feedback_gnn_amd.weights_io import read_weight_list
g = TannerGraph(code('ghp882'))
print(g.info())
B=65536
ex,ez = g.pauli_noise(0x5EED, 0.01, 0, B)
sx,sz = g.syndrome(ex,ez)
L0=llr_const(0.05)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t=time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time()-t)/reps
for tpc,cpb in [(448,1),(896,1),(320,1),(256,1),(448,2),(512,1),(640,1)]:
    try:
        g.set_launch(tpc,cpb)
        dt=timeit(lambda: g.bp4_decode(sx,sz,64,"boxplus-phi",1.0,llr_const=L0))
        print(f"BP64 phi tpc={tpc} cpb={cpb}: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
    except Exception as e: print(tpc,cpb,e)
g.set_launch(0,0)
for cn in ("minsum","boxplus"):
    dt=timeit(lambda: g.bp4_decode(sx,sz,64,cn,1.0,llr_const=L0))
    print(f"BP64 {cn}: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
w=GnnWeights(read_weight_list(WEIGHTS_882), g.device)
o=g.bp4_decode(sx,sz,64,"boxplus-phi",1.0,llr_const=L0)
dt=timeit(lambda: g.feedback_gnn(w,o['llr'],o['z_logit'],o['x_logit'],sx,sz))
print(f"GNN: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
ws=g.sandwich_workspace(B)
dt=timeit(lambda: g.sandwich_decode(sx,sz,[64,16],[w],L0,workspace=ws))
print(f"sandwich 64,G,16: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
dt=timeit(lambda: g.sandwich_decode(sx,sz,[64,16],[w],L0,workspace=ws,compact=True))
print(f"sandwich compact: {dt*1e3:.1f} ms  {B/dt/1e3:.1f} k cw/s")
dt=timeit(lambda: g.residual(ex,ez,o['x_hat'],o['z_hat']))
print(f"residual: {dt*1e3:.1f} ms")
dt=timeit(lambda: g.pauli_noise(0x5EED,0.01,0,B)); print(f"noise: {dt*1e3:.2f} ms")
dt=timeit(lambda: g.syndrome(ex,ez)); print(f"syndrome: {dt*1e3:.2f} ms")

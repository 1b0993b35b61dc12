"""Kernel time of the feedback GNN (and GNN_BP4) with torch events: python tools/ab_gnn.py; FGNN_LIB_PATH selects the build."""
import os, sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const, WEIGHTS_882, WEIGHTS_1270
from feedback_gnn_amd.graph import TannerGraph, GnnWeights, GnnBp4Weights, GNNBP4_SHAPES
from feedback_gnn_amd.weights_io import read_weight_list
tag = os.path.basename(os.environ.get("FGNN_LIB_PATH", "current"))
def ev_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, wf, B in (("ghp882", WEIGHTS_882, 65536), ("ghp1270", WEIGHTS_1270, 32768)):
    g = TannerGraph(code(name))
    ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B); sx, sz = g.syndrome(ex, ez)
    o = g.bp4_decode(sx, sz, 8, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    w = GnnWeights(read_weight_list(wf), g.device)
    out = g.feedback_gnn(w, o['llr'], o['z_logit'], o['x_logit'], sx, sz)
    ms = ev_time(lambda: g.feedback_gnn(w, o['llr'], o['z_logit'], o['x_logit'], sx, sz))
    print(f"[{tag}] {name} GNN B={B}: {ms:.2f} ms  checksum {float(out.double().sum()):.6f}", flush=True)
if "--c5" in sys.argv:
    g = TannerGraph(code('ghp1270')); B = 4096
    rng = np.random.RandomState(0)
    w = GnnBp4Weights([rng.uniform(-0.3, 0.3, size=s).astype(np.float32) for s in GNNBP4_SHAPES], g.device)
    ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B); sx, sz = g.syndrome(ex, ez)
    r = g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False)
    ms = ev_time(lambda: g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False), reps=3)
    print(f"[{tag}] ghp1270 GNN_BP4 10 it B={B}: {ms:.2f} ms ({B/ms:.1f} k cw/s)  checksum {float(r['llr'].double().sum()):.6f}", flush=True)

"""A/B probe: time of one GNN_BP4 launch at the BASELINE configs[4] shard shape (16 384 codewords x 10 iterations).
   FGNN_LIB_PATH=feedback_gnn_amd/lib/ab/libfgnn_hip_<tag>.so python tools/ab_gnnbp4.py [B]"""
import os, sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code
from feedback_gnn_amd.graph import TannerGraph, GnnBp4Weights, GNNBP4_SHAPES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = TannerGraph(code('ghp1270'))
rng = np.random.RandomState(0)
w = GnnBp4Weights([rng.uniform(-0.3, 0.3, size=s).astype(np.float32) for s in GNNBP4_SHAPES], g.device)
ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B); sx, sz = g.syndrome(ex, ez)
o = g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(2): o = g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 2
chk = int(o["x_hat"].sum()) * 31 + int(o["z_hat"].sum())
print(f"[{os.environ.get('FGNN_LIB_PATH', 'default')}] GNN_BP4 {B} x 10 it: {ms:.1f} ms = {B / ms:.1f} k cw/s, {0.875e8 * 10 * B / ms / 1e9:.1f} TFLOP/s, checksum {chk}")

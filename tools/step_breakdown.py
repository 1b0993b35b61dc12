"""Where a fixed-dataflow sandwich step goes at a given batch size: HIP-event time of every BP4 / feedback-GNN launch (fgnn_profile_*)
next to the wall time of the step:   python tools/step_breakdown.py [B ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F  # noqa: E402
from helpers import WEIGHTS_882, code  # noqa: E402

Bs = [int(x) for x in sys.argv[1:]] or [5000, 65536]
c = code("ghp882")
dec1 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True)
g = dec1.graph
dec2 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=g)
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True, graph=g)
F.load_weights(G, WEIGHTS_882)
model = F.Sandwich_BP_GNN_Evaluation_Model(c, [dec1, dec2], [G], num_layers=2, p0=0.05, seed=0x5EED)
g.set_saturation_shortcut(False)
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
for B in Bs:
    for _ in range(3):
        model.mc_step(B, 0.01, counts)
    K = max(4, int(0.5 / (60e-3 * B / 65536 + 60e-6)))
    g.profile_enable(3 * K)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        model.mc_step(B, 0.01, counts)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    L = g.profile_read()
    g.profile_enable(0)
    bp64 = np.mean([ms for ms, it, b in L if it == 64]); gnn = np.mean([ms for ms, it, b in L if it == -1]); bp16 = np.mean([ms for ms, it, b in L if it == 16])
    print(f"B = {B}: step {dt * 1e3:.3f} ms (host issue time {t_host / K * 1e3:.3f} ms per step); BP4-64 {bp64:.3f} + GNN {gnn:.3f} + BP4-16 {bp16:.3f} = {bp64 + gnn + bp16:.3f} ms; "
          f"rest (noise, syndromes, flags, merge, residual, counters, gaps) {dt * 1e3 - bp64 - gnn - bp16:.3f} ms; per codeword: BP4-64 {bp64 / B * 1e3:.4f} us, GNN {gnn / B * 1e3:.4f}, BP4-16 {bp16 / B * 1e3:.4f}", flush=True)

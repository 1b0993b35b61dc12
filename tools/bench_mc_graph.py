"""Monte-Carlo loop at small batches: eager `mc_step` calls against one hipGraph replay per K steps (`mc_graph`), same samples.
    python tools/bench_mc_graph.py [B=256] [K=50] [iters=32] [cn_type=boxplus] [factor=0.625] [p=0.05]"""
import sys
import time

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F  # noqa: E402
from helpers import code  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
iters = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "32").split(",")]
cn = sys.argv[4] if len(sys.argv) > 4 else "boxplus"
factor = float(sys.argv[5]) if len(sys.argv) > 5 else 0.625
p = float(sys.argv[6]) if len(sys.argv) > 6 else 0.05
c = code("ghp882")


def model():
    d0 = F.QLDPCBPDecoder(code=c, num_iter=iters[0], normalization_factor=factor, cn_type=cn, stage_one=True)
    decs = [d0] + [F.QLDPCBPDecoder(code=c, num_iter=it, normalization_factor=factor, cn_type=cn, stage_one=True, graph=d0.graph) for it in iters[1:]]
    G = []
    if len(iters) > 1:
        g = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True, graph=d0.graph)
        F.load_weights(g, "feedback_GNN_n882_k24_wt_4_60_iter_64_16_mixed.npz")
        G = [g] * (len(iters) - 1)
    m = F.Sandwich_BP_GNN_Evaluation_Model(c, decs, G, num_layers=len(iters), seed=0x5EED)
    m.graph.set_saturation_shortcut(False)  # the fixed dataflow bench.py times
    return m


e, g = model(), model()
ce = torch.zeros(3, dtype=torch.int64, device="cuda")
cg = torch.zeros(3, dtype=torch.int64, device="cuda")
replay = g.mc_graph(B, p, K, cg)
for rep in range(3):
    for _ in range(K):
        e.mc_step(B, p, ce)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4 * K):
        e.mc_step(B, p, ce)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t) / (4 * K)
    replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t) / (4 * K)
    print(f"B={B} iters={iters} {cn} {factor}: eager mc_step {te * 1e3:.4f} ms/step ({B / te / 1e3:.0f} k cw/s), hipGraph of {K} steps "
          f"{tg * 1e3:.4f} ms/step ({B / tg / 1e3:.0f} k cw/s), ratio {te / tg:.3f}; counters equal: {ce.tolist() == cg.tolist()}", flush=True)

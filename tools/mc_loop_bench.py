"""Monte-Carlo loop throughput at the reference's batch sizes (n882.py uses 5 000, the notebooks 10 000) next to the bench
batch, product default (exact shortcuts on), feedback rounds compacted:   python tools/mc_loop_bench.py [steps]
Shows how much of a small-batch step is launch / synchronisation overhead rather than kernel time."""
import json, sys, time
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F
from helpers import code

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
c = code('ghp882')
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True)
F.load_weights(G, "feedback_GNN_n882_k24_wt_4_60_iter_64_16_mixed.npz")
d1 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=G.graph)
d2 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=G.graph)
out = {}
for nG in (1, 3):
    for kw in ({"compact": True}, {"compact": True, "graph_capture": True}):
        try:
            m = F.Sandwich_BP_GNN_Evaluation_Model(c, [d1] + [d2] * nG, [G] * nG, num_layers=nG + 1, **kw)
        except TypeError:
            continue
        for p in (0.02, 0.05, 0.10):
            for B in (5000, 10000, 65536):
                cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
                for _ in range(3):
                    m.mc_step(B, p, cnt)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(steps):
                    m.mc_step(B, p, cnt)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / steps
                key = f"nG={nG} {'graph' if kw.get('graph_capture') else 'eager'} p={p} B={B}"
                out[key] = {"ms_per_step": round(dt * 1e3, 3), "M_cw_per_s": round(B / dt / 1e6, 3)}
                print(key, out[key], flush=True)
print(json.dumps(out))

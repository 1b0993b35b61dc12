"""Product default vs fixed dataflow on the same Philox samples, at scale: the compacted sandwich with every exact optimisation on
(saturation shortcut, closed-form first iteration, fixed-point exit, epilogue shortcuts, fused flag test, small-launch geometry)
must return exactly the decisions of the plain sandwich that evaluates every transcendental of every iteration on every sample.
    python tools/exactness_at_scale.py [samples_per_point=4194304]  ->  one JSON line"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import feedback_gnn_amd as F
from helpers import code, WEIGHTS_882, WEIGHTS_1270

total = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
B = 65536
out = {"samples_per_point": total, "points": []}
for name, wf, nG in (("ghp882", WEIGHTS_882, 3), ("ghp1270", WEIGHTS_1270, 1)):
    c = code(name)
    G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True)
    F.load_weights(G, wf)
    g = G.graph
    d1 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=g)
    d2 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=g)
    fast = F.Sandwich_BP_GNN_Evaluation_Model(c, [d1] + [d2] * nG, [G] * nG, num_layers=nG + 1, compact=True)
    slow = F.Sandwich_BP_GNN_Evaluation_Model(c, [d1] + [d2] * nG, [G] * nG, num_layers=nG + 1, compact=False)
    for p in (0.02, 0.06, 0.10, 0.14):
        mism = flagged = 0
        t_fast = t_slow = 0.0
        for first in range(0, total, B):
            g.set_saturation_shortcut(True)
            torch.cuda.synchronize(); t = time.perf_counter()
            a = fast.decode(B, p, first_sample=first)
            torch.cuda.synchronize(); t_fast += time.perf_counter() - t
            g.set_saturation_shortcut(False)
            t = time.perf_counter()
            b = slow.decode(B, p, first_sample=first)
            torch.cuda.synchronize(); t_slow += time.perf_counter() - t
            same = (a["x_hat"] == b["x_hat"]).all(dim=1) & (a["z_hat"] == b["z_hat"]).all(dim=1)
            mism += int((~same).sum())
            _, _, fl = g.residual(a["noise_x"], a["noise_z"], a["x_hat"], a["z_hat"], want_arrays=False)
            flagged += int((fl & 1).sum())
        g.set_saturation_shortcut(True)
        row = {"code": name, "feedback_rounds": nG, "p": p, "samples": total, "samples_with_any_different_decision": mism,
               "still_flagged": flagged, "product_default_s": round(t_fast, 2), "fixed_dataflow_s": round(t_slow, 2)}
        out["points"].append(row)
        print(row, flush=True)
print(json.dumps(out))

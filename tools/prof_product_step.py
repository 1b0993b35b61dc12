"""A few product-default Monte-Carlo steps (exact shortcuts on, feedback rounds compacted) for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/prof_product_step.py [p] [nG] [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import feedback_gnn_amd as F
from helpers import code
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
nG = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
c = code('ghp882')
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True)
F.load_weights(G, "feedback_GNN_n882_k24_wt_4_60_iter_64_16_mixed.npz")
d1 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=G.graph)
d2 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=G.graph)
m = F.Sandwich_BP_GNN_Evaluation_Model(c, [d1] + [d2] * nG, [G] * nG, num_layers=nG + 1, compact=True)
cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
for _ in range(6):
    m.mc_step(B, p, cnt)
torch.cuda.synchronize()
print(cnt.tolist())

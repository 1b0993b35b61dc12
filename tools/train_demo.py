"""Second-stage training demo on the GPU (Feedback_GNN.ipynb cell 8 with harvested failures instead of the absent dataset).
usage: python tools/train_demo.py [code=ghp882] [steps=60] [lr=2e-4] [p=0.09]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import code as get_code
from feedback_gnn_amd import QLDPCBPDecoder, Feedback_GNN, Sandwich_BP_GNN_Evaluation_Model, First_Stage_BP_Model, Second_Stage_GNN_BP_Model
from feedback_gnn_amd.training import Adam, harvest_failures, train_second_stage

name = sys.argv[1] if len(sys.argv) > 1 else "ghp882"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.09
c = get_code(name)
dec1 = QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True)
dec2 = QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_two=True, graph=dec1.graph)
G = Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh",
                 use_bias=True, graph=dec1.graph)
ev = Sandwich_BP_GNN_Evaluation_Model(c, [dec1], [], num_layers=1)
t0 = time.time()
X, Z = harvest_failures(ev, 8192, p, 100 * steps)
print(f"harvested {X.shape[0]} failures in {time.time()-t0:.1f}s, mean weight {((X|Z).sum(1)).mean():.1f}")
m1, m2 = First_Stage_BP_Model(c, dec1), Second_Stage_GNN_BP_Model(c, G, dec2, num_iter=16)
torch.cuda.synchronize(); t0 = time.time()
hist = train_second_stage(m1, m2, X, Z, batch_size=100, learning_rate=lr, log_every=10)
torch.cuda.synchronize(); dt = time.time() - t0
h = np.array(hist)
k = max(1, len(h) // 6)
print(f"{len(h)} steps in {dt:.1f}s ({dt/len(h)*1e3:.1f} ms/step)")
print("loss  first/last sixth:", h[:k, 0].mean(), h[-k:, 0].mean())
print("flagged first/last sixth:", h[:k, 2].mean(), h[-k:, 2].mean())

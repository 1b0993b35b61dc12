"""Train a feedback GNN from the Keras initialisation on MI355X and compare it with BP alone and with the shipped weights.

Follows /root/reference examples/Generate_dataset.ipynb cells 4-5 (fixed-weight errors wt_from..wt_to that BP-64 fails on,
the "easy" set) and examples/Feedback_GNN.ipynb cell 8 (one epoch, batch 100, Adam 2e-4, gradients clipped to +-10).
usage: python tools/train_from_scratch.py [code=ghp882] [per_wt=6000] [batch=65536] [max_rounds=8] [eval_samples=200000]
Writes gpurun_out/train_<code>.json and gpurun_out/trained_<code>.npz.
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import code as get_code, WEIGHTS_882, WEIGHTS_1270
from feedback_gnn_amd import (QLDPCBPDecoder, Feedback_GNN, Sandwich_BP_GNN_Evaluation_Model, First_Stage_BP_Model,
                              Second_Stage_GNN_BP_Model, load_weights)
from feedback_gnn_amd.training import train_second_stage
from feedback_gnn_amd.weights_io import write_weight_list

name = sys.argv[1] if len(sys.argv) > 1 else "ghp882"
per_wt = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
max_rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
eval_samples = int(sys.argv[5]) if len(sys.argv) > 5 else 200000
wt_from, wt_to = (4, 60) if name == "ghp882" else (10, 80)
shipped = WEIGHTS_882 if name == "ghp882" else WEIGHTS_1270
c = get_code(name)
mk = lambda it, **kw: QLDPCBPDecoder(code=c, num_iter=it, normalization_factor=1.0, cn_type="boxplus-phi", **kw)  # noqa: E731
dec1 = mk(64, stage_one=True)
g = dec1.graph
dec2 = mk(16, stage_two=True, graph=g)
dec2e = mk(16, stage_one=True, graph=g)
newG = lambda: Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean",  # noqa: E731
                            activation="tanh", use_bias=True, graph=g)
G = newG()

# ---- dataset: BP-64 failures on fixed-weight errors ----
t0 = time.time()
harvest = Sandwich_BP_GNN_Evaluation_Model(c, [dec1], [], num_layers=1, wt=True)
xs, zs, per = [], [], {}
drawn = 0
for wt in range(wt_from, wt_to + 1):
    have = 0
    for _ in range(max_rounds):
        fx, fz = harvest.failures(batch, wt)
        drawn += batch
        take = min(int(fx.shape[0]), per_wt - have)
        if take > 0:
            xs.append(fx[:take].cpu()); zs.append(fz[:take].cpu())
            have += take
        if have >= per_wt:
            break
    per[wt] = have
X, Z = torch.cat(xs).numpy(), torch.cat(zs).numpy()
t_data = time.time() - t0
print(f"dataset: {X.shape[0]} BP-64 failures out of {drawn} fixed-weight samples in {t_data:.1f}s; per weight (first/last): "
      f"{[per[w] for w in list(per)[:6]]} ... {[per[w] for w in list(per)[-3:]]}", flush=True)

# ---- training: one epoch ----
m1, m2 = First_Stage_BP_Model(c, dec1), Second_Stage_GNN_BP_Model(c, G, dec2, num_iter=16)
torch.cuda.synchronize(); t0 = time.time()
hist = np.array(train_second_stage(m1, m2, X, Z, batch_size=100, learning_rate=2e-4, clip_value_grad=10.0, log_every=500))
torch.cuda.synchronize(); t_train = time.time() - t0
k = max(1, len(hist) // 10)
print(f"training: {len(hist)} steps in {t_train:.1f}s ({t_train/len(hist)*1e3:.1f} ms/step); loss {hist[:k,0].mean():.3f} -> "
      f"{hist[-k:,0].mean():.3f}; flagged rate of the batch after GNN+BP16 {hist[:k,2].mean():.3f} -> {hist[-k:,2].mean():.3f}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
write_weight_list(G.get_weights(), f"gpurun_out/trained_{name}.npz")

# ---- evaluation: depolarizing noise, BP64 then (G, BP16) x 3 (Feedback_GNN.ipynb cell 10) ----
Gs = newG(); load_weights(Gs, shipped)
res = {}
for p in (0.10, 0.08):
    for tag, fb in (("bp64", None), ("bp64+(G_trained_here,bp16)x3", G), ("bp64+(G_shipped,bp16)x3", Gs)):
        decs, fbs, L = ([dec1], [], 1) if fb is None else ([dec1] + [dec2e] * 3, [fb] * 3, 4)
        ev = Sandwich_BP_GNN_Evaluation_Model(c, decs, fbs, num_layers=L, seed=777)
        counts = torch.zeros(3, dtype=torch.int64, device=g.device)
        for _ in range(max(1, eval_samples // 8192)):
            ev.mc_step(8192, p, counts)
        fl, bl, tot = [int(v) for v in counts.cpu()]
        res[f"p={p:.2f} {tag}"] = dict(flagged=fl / tot, bler=bl / tot, samples=tot)
        print(f"p={p:.2f} {tag:32s} flagged {fl/tot:.5f}  logical {bl/tot:.5f}  ({tot} samples)", flush=True)
json.dump(dict(code=name, dataset=int(X.shape[0]), drawn=drawn, t_data_s=t_data, steps=len(hist), t_train_s=t_train,
               ms_per_step=t_train / len(hist) * 1e3, loss_first=float(hist[:k, 0].mean()), loss_last=float(hist[-k:, 0].mean()),
               eval=res), open(f"gpurun_out/train_{name}.json", "w"), indent=1)

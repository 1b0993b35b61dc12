"""The reference's complete training recipe for the [[882,24]] feedback GNN, run on one MI355X from the Keras initialisation:
examples/Generate_dataset.ipynb (easy set = BP failures on fixed-weight errors; coarse GNN; hard set = failures of
BP64 -> GNN -> BP64; mixed set with the hard samples repeated 50x) and examples/Feedback_GNN.ipynb cell 8 (one epoch,
batch 100, Adam 2e-4, clip 10).  Sample counts are scaled by `scale` (1.0 = the notebook's counts).
usage: python tools/train_full_recipe.py [scale=0.2] [eval_samples=300000] [hard_scale=scale] [easy_scale=scale] [quirk=0] [seeds=1] [epochs=1] [first_seed=0] [cosine=0]
`seeds` > 1 repeats the final (mixed-set) training from `seeds` different initialisations / shuffles of the SAME mined data and evaluates each:
the run-to-run spread of the recipe itself.
`hard_scale` scales the hard-sample mining alone (the authors collected their ~11 k hard samples over repeated runs of that cell),
`easy_scale` the easy set of weights 4-40 (theirs holds 439 916 samples = 1.76 passes of cell 5).  `quirk=1` reproduces what
Generate_dataset.ipynb cell 16 actually assembles for [[882,24]]: the z part of the weight-41..60 hard samples is loaded from the
*x* file (`dz_41_60_hard = np.load(".._x_hard.npy")`), i.e. those samples enter training with z = x.
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import code as get_code, WEIGHTS_882
from feedback_gnn_amd import (QLDPCBPDecoder, Feedback_GNN, Sandwich_BP_GNN_Evaluation_Model, First_Stage_BP_Model,
                              Second_Stage_GNN_BP_Model, load_weights)
from feedback_gnn_amd.training import harvest_failures, train_second_stage
from feedback_gnn_amd.weights_io import write_weight_list

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
eval_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
hard_scale = float(sys.argv[3]) if len(sys.argv) > 3 else scale
easy_scale = float(sys.argv[4]) if len(sys.argv) > 4 else scale
quirk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n_seeds = int(sys.argv[6]) if len(sys.argv) > 6 else 1
epochs = int(sys.argv[7]) if len(sys.argv) > 7 else 1
seed0 = int(sys.argv[8]) if len(sys.argv) > 8 else 0  # first seed of the final training (seeds seed0 .. seed0 + seeds - 1)
cosine = int(sys.argv[9]) if len(sys.argv) > 9 else 0  # 1: cosine decay of the learning rate over the final training (the reference's own hint
                                                       # for multi-epoch runs, Feedback_GNN.ipynb cells 2 / 8: CosineDecay(2e-4, decay_steps))
CODE = os.environ.get("FGNN_TRAIN_CODE", "ghp882")  # ghp1270: the [[1270,28]] recipe of Generate_dataset.ipynb cells 4-13 (weights 10-60 / 61-80)
c = get_code(CODE)
if CODE == "ghp882":
    W_EASY1, W_EASY2, W_HARD, N_EASY2, W_SPLIT, WSHIP = range(4, 41), range(41, 61), range(4, 61), 300000, 41, WEIGHTS_882
else:  # easy: 867 278 samples of weight 10-60 + 856 of weight 61-80; hard: weights 10-80 (cell 9)
    from helpers import WEIGHTS_1270
    W_EASY1, W_EASY2, W_HARD, N_EASY2, W_SPLIT, WSHIP = range(10, 61), range(61, 81), range(10, 81), 856, 61, WEIGHTS_1270
mk = lambda it, **kw: QLDPCBPDecoder(code=c, num_iter=it, normalization_factor=1.0, cn_type="boxplus-phi", **kw)  # noqa: E731
dec64 = mk(64, stage_one=True)
g = dec64.graph
dec16 = mk(16, stage_one=True, graph=g)
dec16_2 = mk(16, stage_two=True, graph=g)
newG = lambda seed=0: Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean",  # noqa: E731
                                   activation="tanh", use_bias=True, graph=g, seed=seed)
log = {}
T0 = time.time()


def collect(model, weights, batch, iters, cap=None):
    xs, zs, drawn = [], [], 0
    for wt in weights:
        x, z = harvest_failures(model, batch, wt, cap or 10 ** 9, max_batches=iters, on_device=True)
        drawn += batch * iters
        xs.append(x); zs.append(z)
    return torch.cat(xs), torch.cat(zs), drawn


def train(G, dec_first, X, Z, tag, seed=0, epochs=1):
    m1, m2 = First_Stage_BP_Model(c, dec_first), Second_Stage_GNN_BP_Model(c, G, dec16_2, num_iter=16)
    torch.cuda.synchronize(); t0 = time.time()
    steps = epochs * ((int(X.shape[0]) + 99) // 100)
    lr = (lambda t: 2e-4 * 0.5 * (1.0 + np.cos(np.pi * min(t, steps) / steps))) if (cosine and tag.startswith("mixed")) else 2e-4
    h = np.array(train_second_stage(m1, m2, X, Z, batch_size=100, learning_rate=lr, clip_value_grad=10.0, log_every=4000, seed=seed,
                                    epochs=epochs))
    torch.cuda.synchronize(); dt = time.time() - t0
    k = max(1, len(h) // 10)
    log[tag] = dict(samples=int(X.shape[0]), steps=len(h), seconds=dt, ms_per_step=dt / len(h) * 1e3,
                    loss_first=float(h[:k, 0].mean()), loss_last=float(h[-k:, 0].mean()))
    print(f"[{tag}] {len(h)} steps in {dt:.1f}s ({dt/len(h)*1e3:.1f} ms/step), loss {h[:k,0].mean():.3f} -> {h[-k:,0].mean():.4f}", flush=True)


# 1. easy sets (Generate_dataset.ipynb cells 4-5): BP-64 failures, 50 x 50000 samples per weight
it_easy = max(1, int(round(50 * scale)))
t0 = time.time()
bp_only = Sandwich_BP_GNN_Evaluation_Model(c, [dec64], [], num_layers=1, wt=True)
Xe1, Ze1, d1 = collect(bp_only, W_EASY1, 50000, max(1, int(round(50 * easy_scale))))
Xe2, Ze2, d2 = collect(bp_only, W_EASY2, 50000, it_easy if CODE == 'ghp882' else 1)
n2 = min(int(Xe2.shape[0]), int(N_EASY2 * scale))
sel = torch.from_numpy(np.random.RandomState(0).permutation(int(Xe2.shape[0]))[:n2]).to(g.device)  # seeded: a device randperm draws a new seed per process
Xe2, Ze2 = Xe2[sel], Ze2[sel]
log["easy"] = dict(drawn=d1 + d2, wt_4_40=int(Xe1.shape[0]), wt_41_60_used=n2, seconds=time.time() - t0)
print(f"[easy] {Xe1.shape[0]} (wt 4-40) + {n2} (wt 41-60) failures from {d1+d2} samples in {time.time()-t0:.1f}s", flush=True)

# 2. coarse GNN on the wt 4-40 easy set, first stage of 16 iterations ("..._wt_4_40_iter_16_16")
Gc = newG()
train(Gc, dec16, Xe1, Ze1, "coarse")

# 3. hard sets (cells 8, 15): failures of BP64 -> coarse GNN -> BP64, 200 x 5000 samples per weight
it_hard = max(1, int(round(200 * hard_scale)))
t0 = time.time()
two_stage = Sandwich_BP_GNN_Evaluation_Model(c, [dec64, dec64], [Gc], num_layers=2, wt=True)
Xh, Zh, dh = collect(two_stage, W_HARD, 5000, it_hard)
if CODE != "ghp882":  # cell 13: all hard samples of weight 10-60, 3000 random ones of weight 61-80
    wt_h = (Xh | Zh).sum(1)
    hi_idx = torch.nonzero(wt_h >= W_SPLIT).flatten()
    keep_hi = hi_idx[torch.from_numpy(np.random.RandomState(1).permutation(int(hi_idx.numel()))[:3000]).to(g.device)]
    keep = torch.cat([torch.nonzero(wt_h < W_SPLIT).flatten(), keep_hi])
    Xh, Zh = Xh[keep], Zh[keep]
if quirk:  # cell 16 of Generate_dataset.ipynb: z of the weight 41..60 hard samples := their x
    wt_h = (Xh | Zh).sum(1)
    hi = wt_h >= W_SPLIT
    Zh = torch.where(hi[:, None], Xh, Zh)
log["hard"] = dict(drawn=dh, found=int(Xh.shape[0]), seconds=time.time() - t0, quirk=quirk)
print(f"[hard] {Xh.shape[0]} two-stage failures from {dh} samples in {time.time()-t0:.1f}s", flush=True)

# 4. mixed set (cell 16): easy + hard x 50, then one epoch from a fresh GNN with the 64/16 pipeline
rep = 50
X = torch.cat([Xe1, Xe2] + [Xh] * rep); Z = torch.cat([Ze1, Ze2] + [Zh] * rep)
G = newG(seed0)
train(G, dec64, X, Z, "mixed", seed=seed0, epochs=epochs)
os.makedirs("gpurun_out", exist_ok=True)
write_weight_list(G.get_weights(), f"gpurun_out/trained_full_{CODE}.npz")
extra = []
for sd in range(seed0 + 1, seed0 + n_seeds):
    Gi = newG(sd)
    train(Gi, dec64, X, Z, f"mixed_seed{sd}", seed=sd, epochs=epochs)
    write_weight_list(Gi.get_weights(), f"gpurun_out/trained_full_{CODE}_seed{sd}.npz")
    extra.append((f"trained_here_seed{sd}", Gi))

# 5. evaluation as Feedback_GNN.ipynb cell 10
Gs = newG(); load_weights(Gs, WSHIP)
res = {}
for p in ((0.10, 0.08) if CODE == "ghp882" else (0.12, 0.10)):
    for tag, fb in [("bp64", None), ("coarse", Gc), ("trained_here", G), ("shipped", Gs)] + extra:
        decs, fbs, L = ([dec64], [], 1) if fb is None else ([dec64] + [dec16] * 3, [fb] * 3, 4)
        ev = Sandwich_BP_GNN_Evaluation_Model(c, decs, fbs, num_layers=L, seed=777)
        counts = torch.zeros(3, dtype=torch.int64, device=g.device)
        for _ in range(max(1, eval_samples // 16384)):
            ev.mc_step(16384, p, counts)
        fl, bl, tot = [int(v) for v in counts.cpu()]
        res[f"p={p:.2f} {tag}"] = dict(flagged=fl, block_errors=bl, samples=tot, bler=bl / tot)
        print(f"p={p:.2f} {tag:14s} flagged {fl:6d}  logical {bl:6d} / {tot}  BLER {bl/tot:.5f}", flush=True)
log["eval"] = res
log["scale"] = scale
log["hard_scale"] = hard_scale
log["easy_scale"] = easy_scale
log["seeds"] = n_seeds
log["epochs"] = epochs
log["cosine"] = cosine
log["first_seed"] = seed0
log["total_seconds"] = time.time() - T0
log["code"] = CODE
json.dump(log, open(f"gpurun_out/train_full_{CODE}.json", "w"), indent=1)
print(f"total {time.time()-T0:.1f}s")

"""Fixed-dataflow sandwich step (BASELINE configs[2]: [[882,24]], BP4-64 + feedback GNN + BP4-16, p = 0.01, noise + syndromes + decode +
residual + counters on device) at the reference's own batch sizes — n882.py:66 runs batch_size = 5 000, QLDPC.ipynb 10 000 — next to the
65 536 the headline uses:   python tools/batch_size_table.py [out.json]
Prints codewords/s per batch size and the fraction of the 65 536 rate; also `mc_steps` (k batches decoded as one launch, counters per
batch boundary: what sim_ber uses to keep the reference's batch size at the full-chip rate) and the bare `mc_step` loop of a model built
with `streams=2` (consecutive batches alternate between two HIP streams)."""
import json
import sys
import time

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F  # noqa: E402
from helpers import WEIGHTS_882, code  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else None
c = code("ghp882")
dec1 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True)
g = dec1.graph
dec2 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=g)
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True, graph=g)
F.load_weights(G, WEIGHTS_882)
model = F.Sandwich_BP_GNN_Evaluation_Model(c, [dec1, dec2], [G], num_layers=2, p0=0.05, seed=0x5EED)
g.set_saturation_shortcut(False)  # fixed dataflow, like bench.py's headline
counts = torch.zeros(3, dtype=torch.int64, device="cuda")


def rate(B, min_s=0.6):
    for _ in range(3):
        model.mc_step(B, 0.01, counts)
    torch.cuda.synchronize()
    reps = max(4, int(min_s / (60e-3 * B / 65536 + 60e-6)))
    t = time.perf_counter()
    for _ in range(reps):
        model.mc_step(B, 0.01, counts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return B / dt, dt * 1e3, reps


rows = []
for B in (256, 1024, 2048, 4096, 5000, 10000, 16384, 32768, 65536):
    r, ms, reps = rate(B)
    rows.append({"batch": B, "cw_per_s": r, "ms_per_step": ms, "steps_timed": reps})
full = rows[-1]["cw_per_s"]
for row in rows:
    row["frac_of_65536_rate"] = row["cw_per_s"] / full
    print(f"B = {row['batch']:6d}: {row['ms_per_step']:8.3f} ms per step, {row['cw_per_s'] / 1e3:8.1f} k codewords/s = {row['frac_of_65536_rate']:.3f} of the 65 536 rate "
          f"({row['steps_timed']} steps timed)", flush=True)
# the reference's batch size at the full-chip rate: k batches as one launch, counters at every batch boundary (sim_ber's stopping rule)
ring = torch.zeros((13, 3), dtype=torch.int64, device="cuda")
for _ in range(2):
    model.mc_steps(5000, 0.01, 13, counts, ring)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(8):
    model.mc_steps(5000, 0.01, 13, counts, ring)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 8
extra = {"mc_steps_13x5000": {"cw_per_s": 65000 / dt, "ms": dt * 1e3, "frac_of_65536_rate": 65000 / dt / full}}
print(f"mc_steps(5000, k = 13) [65 000 codewords as one launch, counters per 5 000]: {dt * 1e3:.2f} ms, {65000 / dt / 1e3:.1f} k codewords/s = {65000 / dt / full:.3f}")
# Sandwich_BP_GNN_Evaluation_Model(streams=2): consecutive, independent batches issued alternately on two HIP streams (own workspaces,
# shared atomic counters), so that one batch's kernels fill the SIMDs the other's prologues, epilogues and kernel tails leave idle
m2s = F.Sandwich_BP_GNN_Evaluation_Model(c, [dec1, dec2], [G], num_layers=2, p0=0.05, seed=0x5EED, streams=2)
two = []
for B in (1024, 2048, 4096, 5000, 10000, 16384, 65536):
    for _ in range(6):
        m2s.mc_step(B, 0.01, counts)
    m2s.join()
    torch.cuda.synchronize()
    reps = 2 * max(4, int(0.3 / (60e-3 * B / 65536 + 60e-6)))
    t = time.perf_counter()
    for _ in range(reps):
        m2s.mc_step(B, 0.01, counts)
    m2s.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    two.append({"batch": B, "cw_per_s": B / dt, "ms_per_step": dt * 1e3, "frac_of_65536_rate": B / dt / full})
    print(f"two streams, B = {B:6d}: {dt * 1e3:8.3f} ms per step, {B / dt / 1e3:8.1f} k codewords/s = {B / dt / full:.3f} of the 65 536 rate", flush=True)
extra["two_streams"] = two
if OUT:
    json.dump({"rows": rows, **extra}, open(OUT, "w"), indent=1)

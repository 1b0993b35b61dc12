"""Does the reference-shaped harness (PlotBER.simulate -> sim_ber, /root/reference sionna/utils/misc.py:636-738) run at the rate of the
bare device loop?  Same model (ghp882, BP4-64 + nG x (GNN, BP4-16), compacted, product default), same p, same number of batches:
  (a) model.mc_step in a Python loop with one read-back at the end  (what bench.py's extras time)
  (b) PlotBER.simulate with device counters (the default)
  (c) PlotBER.simulate on the per-batch array path (device_counters=False: what round 1 did)
python tools/harness_rate.py [p] [nG] [batches] [batch_size=65536]"""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F
from helpers import code, WEIGHTS_882
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
nG = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
B = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
c = code("ghp882")
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True)
F.load_weights(G, WEIGHTS_882)
d1 = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=G.graph)
d2 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=G.graph)
def model():
    return F.Sandwich_BP_GNN_Evaluation_Model(c, [d1] + [d2] * nG, [G] * nG, num_layers=nG + 1, compact=True)
m = model(); cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
for _ in range(3): m.mc_step(B, p, cnt)
torch.cuda.synchronize(); cnt.zero_(); m = model()
t = time.perf_counter()
for _ in range(K): m.mc_step(B, p, cnt)
ref = cnt.tolist(); ta = time.perf_counter() - t
print(f"(a) bare mc_step loop      : {K * B / ta / 1e6:7.2f} M cw/s   counters {ref}")
for name, dc in (("(b) simulate, device counters", True), ("(c) simulate, per-batch arrays", False)):
    pb = F.PlotBER(); m = model()
    t = time.perf_counter()
    pb.simulate(m, ebno_dbs=[p], batch_size=B, num_target_block_errors=10**9, max_mc_iter=K, early_stop=False, verbose=False,
                add_bler=True, device_counters=dc)
    tb = time.perf_counter() - t
    st = F.sim_ber.last
    got = [int(st["flag_errors"][0]), int(st["block_errors"][0]), int(st["num_blocks"][0])]
    print(f"{name}: {K * B / tb / 1e6:7.2f} M cw/s   counters {got}  identical: {got == ref}   rate vs (a): {ta / tb:.3f}")

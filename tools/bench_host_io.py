"""PCIe-inclusive rate of the sandwich decoder when the syndromes arrive in HOST memory and the decisions are wanted back there
(a decoder fed by an experiment rather than by the on-device Monte-Carlo channel):  python tools/bench_host_io.py [p] [chunks]

Pinned host buffers, three torch streams (H2D, decode, D2H) chained by events, two chunks in flight; syndromes as uint8 [B, m] as
the C ABI takes them, decisions returned bit-packed (fgnn_pack_decisions, 2n bits per codeword) or as uint8 [B, n] x 2."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F
from helpers import code, WEIGHTS_882
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 65536
c = code("ghp882")
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True)
F.load_weights(G, WEIGHTS_882)
g = G.graph
W = [G.device_weights]
L0 = float(np.log(np.float32(3 * (1 - 0.05) / 0.05)))
# K chunks of measured syndromes in pinned host memory (generated on the device once, copied out: not part of the timed region)
host_sx, host_sz = [], []
for k in range(K):
    ex, ez = g.pauli_noise(0x5EED, p, k * B, B); sx, sz = g.syndrome(ex, ez)
    host_sx.append(sx.cpu().pin_memory()); host_sz.append(sz.cpu().pin_memory())
nb = (2 * g.n + 7) // 8
for label, shortcut, compact, packed in (("fixed dataflow, packed decisions", False, False, True),
                                         ("product default + compaction, packed decisions", True, True, True),
                                         ("product default + compaction, uint8 decisions", True, True, False)):
    g.set_saturation_shortcut(shortcut)
    out_host = [torch.empty((B, nb) if packed else (2, B, g.n), dtype=torch.uint8).pin_memory() for _ in range(K)]
    s_in, s_run, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ws = [g.sandwich_workspace(B) for _ in range(2)]
    def run():
        done = []
        bufs = [None, None]
        for k in range(K):
            slot = k & 1
            with torch.cuda.stream(s_in):
                if bufs[slot] is not None:
                    s_in.wait_event(bufs[slot][2])  # the slot's previous decode has consumed its inputs
                dx = host_sx[k].to("cuda", non_blocking=True); dz = host_sz[k].to("cuda", non_blocking=True)
                e_in = torch.cuda.Event(); e_in.record(s_in)
            with torch.cuda.stream(s_run):
                s_run.wait_event(e_in)
                o = g.sandwich_decode(dx, dz, [64, 16], W, L0, compact=compact, workspace=ws[slot])
                res = F.pack_decisions(o["x_hat"], o["z_hat"]) if packed else torch.stack([o["x_hat"], o["z_hat"]])
                e_run = torch.cuda.Event(); e_run.record(s_run)
            with torch.cuda.stream(s_out):
                s_out.wait_event(e_run)
                out_host[k].copy_(res, non_blocking=True)
                e_out = torch.cuda.Event(); e_out.record(s_out)
            bufs[slot] = (dx, dz, e_run, res, e_out)
            done.append((dx, dz, o, res))  # keep the tensors alive until the end of the timed region
        torch.cuda.synchronize()
    run()
    t = time.perf_counter(); run(); dt = time.perf_counter() - t
    # device-resident rate of the same decode for comparison
    dx, dz = host_sx[0].cuda(), host_sz[0].cuda()
    g.sandwich_decode(dx, dz, [64, 16], W, L0, compact=compact, workspace=ws[0]); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): g.sandwich_decode(dx, dz, [64, 16], W, L0, compact=compact, workspace=ws[0])
    torch.cuda.synchronize(); dd = (time.perf_counter() - t) / 3
    byt = (g.m_x + g.m_z) + (nb if packed else 2 * g.n)
    print(f"{label}: host->host {K * B / dt / 1e6:6.2f} M cw/s ({K * B * byt / dt / 1e9:5.2f} GB/s over PCIe, {byt} B per codeword); "
          f"device-resident {B / dd / 1e6:6.2f} M cw/s")
g.set_saturation_shortcut(True)

"""Feedback GNN (20/40/2/mean/tanh/bias) at a BASELINE shard shape, both associations, on the MFMA-tile kernel and on the streaming
kernel: time per launch (HIP events) and bit-equality of what they return.    python tools/ab_gnn_literal_stream.py [code] [B]
(FGNN_LIB_PATH=<other build> for A/B builds of the same ABI)"""
import sys

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F  # noqa: E402
from helpers import WEIGHTS_882, WEIGHTS_1270, code, llr_const  # noqa: E402

NAME = sys.argv[1] if len(sys.argv) > 1 else "ghp882"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
c = code(NAME)
dec = F.QLDPCBPDecoder(code=c, num_iter=64, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True)
g = dec.graph
G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True, graph=g)
F.load_weights(G, WEIGHTS_882 if NAME == "ghp882" else WEIGHTS_1270)
ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B)
sx, sz = g.syndrome(ex, ez)
o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
llr, lx, lz = o["llr"], o["x_logit"], o["z_logit"]


def run(stream, fact, reps=5):
    g.set_gnn_factored(fact)
    g.set_gnn_stream(stream)
    out = g.feedback_gnn(G.device_weights, llr, lx, lz, sx, sz)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.feedback_gnn(G.device_weights, llr, lx, lz, sx, sz)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


full = (llr, lx, lz, sx, sz)
for Bs in (B, 8192, 4096, 2048, 1024, 256):  # the launch sizes either side of the library's switch (4 096 codewords)
    if Bs > B:
        continue
    llr, lx, lz, sx, sz = [t[:Bs].contiguous() for t in full]
    for fact in (True, False):
        t0, o0 = run(False, fact, reps=5 if Bs == B else 50)
        t1, o1 = run("always", fact, reps=5 if Bs == B else 50)
        print(f"{NAME} feedback GNN, B={Bs}, {'factored' if fact else 'literal'} association: MFMA tiles {t0:.3f} ms, streaming kernel {t1:.3f} ms, "
              f"{t0 / t1:.3f}x; outputs bit-equal: {torch.equal(o0, o1)}", flush=True)

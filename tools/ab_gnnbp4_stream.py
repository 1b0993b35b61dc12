"""GNN_BP4 at the configs[4] shard shape (16 384 codewords x 10 iterations of [[1270,28]]) on the MFMA-tile kernel and on the streaming
packed-FMA kernel, both associations: time per launch (HIP events) and bit-equality of everything the two kernels return.
    python tools/ab_gnnbp4_stream.py [B]"""
import sys

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from bench import gnnbp4_flops_per_codeword, gnnbp4_seeded_weights  # noqa: E402
from helpers import code  # noqa: E402
from feedback_gnn_amd.graph import GnnBp4Weights, TannerGraph  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = TannerGraph(code('ghp1270'))
w = GnnBp4Weights(gnnbp4_seeded_weights(0), g.device)
ex, ez = g.pauli_noise(0x5EED, 0.01, 0, B)
sx, sz = g.syndrome(ex, ez)
ws = torch.empty(B * (g.n + g.m_x + g.m_z) * 20 * 4, dtype=torch.uint8, device='cuda')
flops = gnnbp4_flops_per_codeword(g.n, g.m_x + g.m_z, g.E_x + g.E_z, 10) * B


def run(stream, fact, reps=3):
    g.set_gnn_factored(fact)
    g.set_gnn_stream(stream)
    o = g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False, workspace=ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False, workspace=ws)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, o


for fact in (True, False):
    t0, o0 = run(False, fact)
    t1, o1 = run("always", fact)
    same = all(torch.equal(o0[k], o1[k]) for k in ("llr", "x_hat", "z_hat"))
    print(f"[[1270,28]] GNN_BP4 10 it, B={B}, {'factored' if fact else 'literal'} association: MFMA tiles {t0:.1f} ms ({flops / t0 / 1e9:.1f} TFLOP/s of the "
          f"reference's algorithm), streaming packed-FMA kernel {t1:.1f} ms ({flops / t1 / 1e9:.1f} TFLOP/s), {t0 / t1:.3f}x; outputs bit-equal: {same}", flush=True)

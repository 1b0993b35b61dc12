"""Launch the GNN_BP4 streaming kernel (and the MFMA kernel) once each at a given batch for rocprofv3 passes: python3 tools/prof_gnnbp4_stream.py [B] [stream|mfma|both]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code
from feedback_gnn_amd.graph import TannerGraph, GnnBp4Weights
from bench import gnnbp4_seeded_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
which = sys.argv[2] if len(sys.argv) > 2 else "both"
g = TannerGraph(code('ghp1270'))
w = GnnBp4Weights(gnnbp4_seeded_weights(0), g.device)
ex, ez = g.pauli_noise(0x5EED, 0.01, 0, B); sx, sz = g.syndrome(ex, ez)
for mode in ((False, "always") if which == "both" else (("always",) if which == "stream" else (False,))):
    g.set_gnn_stream(mode)
    for _ in range(2):
        g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False)
torch.cuda.synchronize(); print("done")

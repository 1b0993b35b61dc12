"""Launch the GNN_BP4 kernel at the C5 shard shape (for rocprofv3 --kernel-trace / --pmc passes): python3 tools/prof_gnnbp4.py [B]"""
import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code
from feedback_gnn_amd.graph import TannerGraph, GnnBp4Weights
from bench import gnnbp4_seeded_weights  # the weights bench.py --config c5 times
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384  # BASELINE configs[4]: 131 072 / 8 GPUs
g = TannerGraph(code('ghp1270'))
import os
if os.environ.get("FGNN_BENCH_GNN_ORDER"):  # "literal" (the library default) / "factored": the same switch as bench.py
    g.set_gnn_factored(os.environ["FGNN_BENCH_GNN_ORDER"] == "factored")
w = GnnBp4Weights(gnnbp4_seeded_weights(0), g.device)
ex, ez = g.pauli_noise(0x5EED, 0.01, 0, B); sx, sz = g.syndrome(ex, ez)
for _ in range(2):
    g.gnn_bp4_decode(w, sx, sz, 10, return_logits=False)
torch.cuda.synchronize(); print("done")

"""Launch each hot kernel a few times at the benchmark shape (for rocprofv3 --kernel-trace / --pmc passes).
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... -d out -- python3 tools/prof_kernels.py [code] [B] [fixed|product] [iters, e.g. 64,16] [p] [cn_type] [factor]
One iteration count (e.g. 32) = the BP4 launch alone (BASELINE configs[0] / [1]).  `fixed` (default) switches the exact shortcuts off, like bench.py's headline: every exp/log of every iteration is evaluated."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const, WEIGHTS_882, WEIGHTS_1270
from feedback_gnn_amd.graph import TannerGraph, GnnWeights
from feedback_gnn_amd.weights_io import read_weight_list
name = sys.argv[1] if len(sys.argv) > 1 else 'ghp882'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ITERS = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [64, 16]
g = TannerGraph(code(name))
g.set_saturation_shortcut(len(sys.argv) > 3 and sys.argv[3] == 'product')
import os
if os.environ.get("FGNN_BENCH_BP4_LSE"):  # "literal" (the library default) / "shared": the same switches as bench.py
    g.set_bp4_shared_lse(os.environ["FGNN_BENCH_BP4_LSE"] == "shared")
if os.environ.get("FGNN_BENCH_GNN_ORDER"):  # "literal" (the library default) / "factored"
    g.set_gnn_factored(os.environ["FGNN_BENCH_GNN_ORDER"] == "factored")
P = float(sys.argv[5]) if len(sys.argv) > 5 else 0.01
CN = sys.argv[6] if len(sys.argv) > 6 else "boxplus-phi"
FACTOR = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0
ex, ez = g.pauli_noise(0x5EED, P, 0, B)
sx, sz = g.syndrome(ex, ez)
L0 = llr_const(0.05)
w = GnnWeights(read_weight_list(WEIGHTS_882 if name == 'ghp882' else WEIGHTS_1270), g.device)
for _ in range(2):
    o = g.bp4_decode(sx, sz, ITERS[0], CN, FACTOR, llr_const=L0)
    if len(ITERS) == 1:
        continue
    nl = g.feedback_gnn(w, o['llr'], o['z_logit'], o['x_logit'], sx, sz)
    o2 = g.bp4_decode(sx, sz, ITERS[1], CN, FACTOR, llr_ch=nl)
    g.residual(ex, ez, o2['x_hat'], o2['z_hat'])
torch.cuda.synchronize()
print("done")

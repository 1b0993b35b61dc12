"""Per-sample agreement of the library's default operation sequence with the reference's formulas term by term
(FGNN_OPT_BP4_SHARED_LSE = FGNN_OPT_GNN_FACTORED = 0), through the whole (64, G, 16) / (64, G, 64) sandwich of the benchmark:

    python tools/forms_agreement.py [samples_per_point] [out.json] [p,p,...]

For every physical error rate: number of samples whose final decisions differ, max |dLLR| of the last decoder's marginals, number of
samples with |dLLR| > 1e-4, samples left flagged by either form.  Same Philox samples for both forms."""
import json
import sys

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import WEIGHTS_882, WEIGHTS_1270, code, llr_const  # noqa: E402
from feedback_gnn_amd.graph import GnnWeights, TannerGraph  # noqa: E402
from feedback_gnn_amd.weights_io import read_weight_list  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 65536
OUT = sys.argv[2] if len(sys.argv) > 2 else None
PS = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0.005, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.08, 0.10)
L0 = llr_const(0.05)
rows = []


def merge(tot, r):
    for k, v in r.items():
        if isinstance(v, dict):
            merge(tot[k], v)
        else:
            tot[k] = max(tot[k], v) if k.startswith("max_") else tot[k] + v


for name, wfile, iters in (("ghp882", WEIGHTS_882, [64, 16]), ("ghp1270", WEIGHTS_1270, [64, 64])):
    g = TannerGraph(code(name))
    g.set_saturation_shortcut(False)  # the benchmark's fixed dataflow (the shortcut is bit-identical anyway)
    w = GnnWeights(read_weight_list(wfile), g.device)
    for p in PS:
        tot = None
        for first in range(0, N, 65536):
            nb = min(65536, N - first)
            ex, ez = g.pauli_noise(0x5EED, p, first, nb)
            sx, sz = g.syndrome(ex, ez)
            r = g.forms_agreement(sx, sz, iters, [w], L0)
            if tot is None:
                tot = r
            else:
                merge(tot, r)
        tot.update(code=name, p=p, iters=iters)
        rows.append(tot)
        print(json.dumps(tot), flush=True)
        torch.cuda.synchronize()
if OUT:
    json.dump(rows, open(OUT, "w"), indent=1)

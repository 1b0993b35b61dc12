"""BP4-64 on [[882,24]] at 65 536 codewords for the three check-node rules of QLDPCBPDecoder (decoding_q.py:18,95-107), fixed dataflow and
product default:  python tools/bench_cn_types.py"""
import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code
from feedback_gnn_amd.graph import TannerGraph
g = TannerGraph(code('ghp882')); B = 65536
L0 = float(np.log(np.float32(3 * (1 - 0.05) / 0.05)))
ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B); sx, sz = g.syndrome(ex, ez)
for cn, f in (("boxplus-phi", 1.0), ("boxplus", 0.625), ("minsum", 0.8)):
    for on in (False, True):
        g.set_saturation_shortcut(on)
        g.bp4_decode(sx, sz, 64, cn, f, llr_const=L0); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): g.bp4_decode(sx, sz, 64, cn, f, llr_const=L0)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        print(f"BP4-64 {cn:12s} factor {f}: {ms:7.2f} ms per {B} = {B / ms / 1e3:6.2f} M cw/s  ({'product default' if on else 'fixed dataflow'})")

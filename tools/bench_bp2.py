"""Binary syndrome BP (LDPCBPDecoder(is_syndrome=True), /root/reference sionna/fec/ldpc/decoding.py) on the hx graph of [[882,24]]:
kernel time at the reference's QLDPC.ipynb cell 7 setting (64 iterations, boxplus-phi) for 65 536 syndromes.   python tools/bench_bp2.py"""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code
from feedback_gnn_amd.graph import TannerGraph
import numpy as np
g = TannerGraph(code('ghp882'))
B = 65536
if os.environ.get('BP2_LAUNCH'):  # A/B of the launch geometry: BP2_LAUNCH=tpc,cpb
    g.set_launch(*map(int, os.environ['BP2_LAUNCH'].split(',')))
for p in (0.01, 0.05):
    e = g.bsc_noise(0x5EED, p, 0, B)
    sx, _ = g.syndrome(torch.zeros_like(e), e)
    L = float(np.log((1 - p) / p))
    for cn, f in (("boxplus-phi", 1.0), ("minsum", 0.8), ("boxplus", 1.0)):
        g.bp2_decode(sx, 64, cn, f, llr_const=L, B=B)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): g.bp2_decode(sx, 64, cn, f, llr_const=L, B=B)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        print(f"BP2-64 [[882,24]] hx graph, p={p}, {cn}: {ms:.2f} ms per {B} syndromes = {B / ms / 1e3:.2f} M/s (reference boxplus-phi row, RTX 4090: 81.3 k/s)")

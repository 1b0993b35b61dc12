"""BASELINE.json configs[0]: [[882,24]] quaternary BP 32 iterations, batch 256, p = 0.05 — the reference's CPU-runnable case.
Times the CPU oracle (OpenMP, all host cores) and the GPU on that exact shape for both check-node rules the reference could
mean (class default 'boxplus' with factor 0.625, and the notebooks' 'boxplus-phi'), end to end:
noise -> syndromes -> BP4-32 -> residual check.   python tools/bench_c1.py"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
from oracle.oracle import OracleGraph, num_threads
B, P, IT = 256, 0.05, 32
c = code('ghp882'); og = OracleGraph(c, forms="library-default"); gg = TannerGraph(c)
out = {"config": "[[882,24]] BP4 32 iterations, batch 256, p=0.05 (BASELINE.json configs[0])", "cpu_threads": num_threads()}
for cn, fac in (("boxplus", 0.625), ("boxplus-phi", 0.625)):
    # GPU first: the oracle's OpenMP threads keep spinning for a while after a parallel region and would slow the launch loop
    def gpu():
        ex, ez = gg.pauli_noise(0x5EED, P, 0, B); sx, sz = gg.syndrome(ex, ez)
        g = gg.bp4_decode(sx, sz, IT, cn, fac, llr_const=llr_const(P), want_logits=False); return gg.residual(ex, ez, g['x_hat'], g['z_hat'], want_arrays=False)[2], g
    time.sleep(0.5); gpu(); torch.cuda.synchronize(); t = time.perf_counter(); reps = 50
    for _ in range(reps): gfl, g = gpu()
    torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t) / reps
    def cpu():
        ex, ez = og.pauli_noise(0x5EED, P, 0, B); sx, sz = og.syndrome(ex, ez)
        o = og.bp4_decode(sx, sz, IT, cn, fac, llr_const=llr_const(P)); return og.residual(ex, ez, o['x_hat'], o['z_hat'])[2], o
    cpu(); t = time.perf_counter(); reps = 5
    for _ in range(reps): fl, o = cpu()
    t_cpu = (time.perf_counter() - t) / reps
    same = bool(np.array_equal(fl, gfl.cpu().numpy()) and np.array_equal(o['llr'], g['llr'].cpu().numpy()))
    out[cn] = {"cpu_ms_per_batch": t_cpu * 1e3, "cpu_cw_per_s": B / t_cpu, "gpu_ms_per_batch": t_gpu * 1e3, "gpu_cw_per_s": B / t_gpu,
               "bit_identical": same, "flagged": int((fl & 1).sum())}
print(json.dumps(out))

"""How many CPUs does this box really grant, and how do the two CPU baselines of bench.py scale with threads?  A measurement helper for
bench.py's cpu_baseline legs (it times oracle/torch_cpu_baseline.py — test infrastructure, nothing of the product):
    python tools/cpu_probe.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
import helpers as H
from oracle import torch_cpu_baseline as T
from feedback_gnn_amd.weights_io import read_weight_list
og=H.oracle_library_forms('ghp882'); w=read_weight_list(H.WEIGHTS_882); L0=H.llr_const(0.05)
tg=T.Graph(H.code('ghp882'))
torch.set_flush_denormal(True)
for B in (256, 1024):
    ex,ez=og.pauli_noise(0x5EED,0.01,0,B); sx,sz=og.syndrome(ex,ez)
    for nt in (8,16,32,64,128):
        torch.set_num_threads(nt)
        T.sandwich_decode(tg,w,sx[:32],sz[:32],[2,2],L0)
        t=time.perf_counter(); T.sandwich_decode(tg,w,sx,sz,[16,4],L0); dt=time.perf_counter()-t
        print(f"torch B={B} threads={nt}: 20 iterations {dt:.2f}s -> full (80 it) ~{B/(dt*4):.1f} cw/s", flush=True)

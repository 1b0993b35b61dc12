"""Kernel-only time (HIP events on the launch stream) of the fixed-dataflow BP4-64 (constant channel LLR) and BP4-16 (per-qubit LLRs) in
both forms of the qubit update, with a CRC of every output so that builds can be compared bit for bit:
    python tools/ab_bp4_forms.py [code]          FGNN_LIB_PATH selects the build (tools/ab_variants.sh walks feedback_gnn_amd/lib/ab/)"""
import os, sys, zlib, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code, llr_const
from feedback_gnn_amd.graph import TannerGraph
name = sys.argv[1] if len(sys.argv) > 1 else 'ghp882'
g = TannerGraph(code(name))
B = 65536 if name == 'ghp882' else 32768
L0 = llr_const(0.05)
tag = os.path.basename(os.environ.get("FGNN_LIB_PATH", "current"))


def crc(o):
    c = 0
    for k in ("llr", "x_hat", "z_hat", "x_logit", "z_logit"):
        c = zlib.crc32(o[k].cpu().numpy().tobytes(), c)
    return f"{c:08x}"


ex, ez = g.pauli_noise(0x5EED, 0.05, 0, B); sx, sz = g.syndrome(ex, ez)
g.set_saturation_shortcut(False)
for shared in (False, True):
    g.set_bp4_shared_lse(shared)
    o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0)
    llr = o["llr"].clone()
    o2 = g.bp4_decode(sx, sz, 16, "boxplus-phi", 1.0, llr_ch=llr)
    c1, c2 = crc(o), crc(o2)
    res = []
    for it, kw in ((64, dict(llr_const=L0)), (16, dict(llr_ch=llr))):
        g.profile_enable(8)
        for _ in range(6): g.bp4_decode(sx, sz, it, "boxplus-phi", 1.0, **kw)
        torch.cuda.synchronize()
        ms = sorted(r[0] if isinstance(r, (tuple, list)) else r for r in g.profile_read())
        g.profile_enable(0)
        res.append(ms[len(ms) // 2])
    print(f"[{tag}] {name} B={B} {'shared ' if shared else 'literal'}: BP4-64 {res[0]:.3f} ms (crc {c1})  BP4-16+llr_ch {res[1]:.3f} ms (crc {c2})", flush=True)

"""Fidelity and speed of the opt-in hardware-transcendental BP4 (FGNN_OPT_HW_TRANSCENDENTALS) against the exact kernel on the same
samples:  python tools/hwt_fidelity.py [p ...]"""
import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import code
from feedback_gnn_amd.graph import TannerGraph
c = code('ghp882'); g = TannerGraph(c); B = 65536
L0 = float(np.log(np.float32(3 * (1 - 0.05) / 0.05)))
g.set_saturation_shortcut(False)
hxp = torch.from_numpy(np.asarray(c.hx_perp)).to(g.device).float(); hzp = torch.from_numpy(np.asarray(c.hz_perp)).to(g.device).float()
for p in [float(x) for x in sys.argv[1:]] or [0.01, 0.06, 0.10]:
    ex, ez = g.pauli_noise(0x5EED, p, 0, B); sx, sz = g.syndrome(ex, ez)
    def run():
        g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0); torch.cuda.synchronize()
        t = time.perf_counter(); o = g.bp4_decode(sx, sz, 64, "boxplus-phi", 1.0, llr_const=L0); torch.cuda.synchronize()
        return o, time.perf_counter() - t
    a, ta = run()
    g.set_hw_transcendentals(True); h, th = run(); g.set_hw_transcendentals(False)
    ones = torch.ones(B, dtype=torch.uint8, device=g.device)
    ca = g.flag_update(a["x_hat"], a["z_hat"], sx, sz, ones.clone()) == 0
    ch = g.flag_update(h["x_hat"], h["z_hat"], sx, sz, ones.clone()) == 0
    both = ca & ch
    same = (a["x_hat"] == h["x_hat"]).all(1) & (a["z_hat"] == h["z_hat"]).all(1)
    dx, dz = (a["x_hat"] ^ h["x_hat"]).float(), (a["z_hat"] ^ h["z_hat"]).float()
    equiv = ~(((dx @ hxp.t()) % 2).bool().any(1) | ((dz @ hzp.t()) % 2).bool().any(1))
    dl = (a["llr"] - h["llr"]).abs().flatten(1).max(1).values
    print(f"p={p}: exact {ta*1e3:.1f} ms, hw {th*1e3:.1f} ms ({ta/th:.2f}x); converged exact {int(ca.sum())} hw {int(ch.sum())} both {int(both.sum())}; "
          f"on both-converged: identical decisions {float(same[both].float().mean()):.5f}, same class {float(equiv[both].float().mean()):.5f}, "
          f"|dLLR|<=1e-4 {float((dl[both] <= 1e-4).float().mean()):.5f}, median |dLLR| {float(dl[both].median()):.3g}, max {float(dl[both].max()):.3g}; "
          f"max llr hw {[f'{v:.9g}' for v in h['llr'].amax(dim=(0, 2)).cpu().numpy()]} finite {bool(torch.isfinite(h['llr']).all())}")

"""Timing of BP4(min-sum,120 it)+OSD-0 at the shape of examples/OSD.ipynb cell 6 (50 000 samples, p=0.09)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feedback_gnn_amd as F
from helpers import code
c = code('ghp882')
dec = F.QLDPCBPDecoder(code=c, num_iter=120, normalization_factor=0.8, cn_type="minsum", stage_one=True)
m = F.BP4_OSD_Model(c, dec, F.OSD0_Decoder(c.N))
m(50000, 0.09); torch.cuda.synchronize()
t = time.time(); z, ls = m(50000, 0.09); torch.cuda.synchronize(); dt = time.time() - t
print(f"BP4-minsum-120 + OSD-0, 50000 samples p=0.09: {dt*1e3:.1f} ms ({50000/dt/1e3:.0f} k cw/s), OSD on {m.last_num_osd} failures, "
      f"logical errors {int(ls.any(1).sum())}  [reference: 9.7 s on an RTX 4090, OSD.ipynb cell 6]")
g = m.graph
ex, ez = g.pauli_noise(1, 0.09, 0, 50000); sx, sz = g.syndrome(ex, ez)
o = g.bp4_decode(sx, sz, 120, "minsum", 0.8, llr_const=3.3, want_logits=False)
fl = g.residual(ex, ez, o['x_hat'], o['z_hat'], want_arrays=False)[2]
idx, n = g.compact(fl, 1)
torch.cuda.synchronize(); t = time.time()
g.osd0(0, sx, o['z_hat'], marg=o['llr'], index=idx, nact=n); g.osd0(1, sz, o['x_hat'], marg=o['llr'], index=idx, nact=n)
torch.cuda.synchronize(); print(f"OSD-0 alone on {n} samples (both sides): {(time.time()-t)*1e3:.2f} ms  [reference 3.78 s for 649 samples]")

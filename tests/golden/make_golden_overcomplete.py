#!/usr/bin/env python3
"""Generate tests/golden/overcomplete.npz: the two over-complete generalized-bicycle check matrices the reference ships as
A-list DATA files (sionna/fec/ldpc/codes_q/GB_46_2_H_800.alist, GB_48_6_H_2000.alist; used by examples/QLDPC.ipynb cell 5),
read with the reference's own `readAlist` and turned into codes with the reference's own `css_code` (both NumPy-only,
executed as in make_golden_codes.py).  The .npz holds bit-packed matrices and scalars only.

    python tests/golden/make_golden_overcomplete.py        # build container only (needs /root/reference)
"""
import os

import numpy as np

from make_golden_codes import HERE, REF, load_reference_namespace, pack


def main():
    R = load_reference_namespace()
    out = {}
    for key, fname, half in (("gb46_oc", "GB_46_2_H_800.alist", 400), ("gb48_oc", "GB_48_6_H_2000.alist", 1000)):
        pcm = R["readAlist"](f"{REF}/sionna/fec/ldpc/codes_q/{fname}")
        c = R["css_code"](hx=pcm[:half], hz=pcm[half:], name=None, name_prefix="GB")  # QLDPC.ipynb cell 5
        for attr in ("hx", "hz", "hx_perp", "hz_perp", "lx", "lz"):
            bits, shape = pack(getattr(c, attr))
            out[f"{key}/{attr}"] = bits
            out[f"{key}/{attr}_shape"] = shape
        out[f"{key}/scalars"] = np.array([c.N, c.K, int(c.D), int(c.L), int(c.Q), c.rank_hx, c.rank_hz], dtype=np.int64)
        out[f"{key}/name"] = np.array(c.name)
        col = [int(pcm[:half].sum(0).min()), int(pcm[:half].sum(0).max())]
        row = [int(pcm.sum(1).min()), int(pcm.sum(1).max())]
        print(f"{key}: {c.name} pcm {pcm.shape} N={c.N} K={c.K} rank=({c.rank_hx},{c.rank_hz}) row weights {row} hx column weights {col}")
    path = os.path.join(HERE, "overcomplete.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

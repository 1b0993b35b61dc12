"""Exhaustive checks of the shared float32 math (feedback_gnn_amd/csrc/fgnn_math.h) that do not depend on BP dynamics.

The oracle is compiled from the product's own math header, so "HIP == oracle" on decoder outputs alone could not see (a) a wrong
constant in the header — both sides would share it — or (b) a divergence between the hipcc/gfx950 build and the gcc/x86 build on an
input BP never visits.  Two tests close that:

  (a) `test_shared_math_against_float64_libm_exhaustively` (CPU): every elementary routine against double-precision libm over every
      float32 of the domain it is used on, with the worst error in ulps bounded — a wrong coefficient, table entry or threshold
      shows up as thousands of ulps;
  (b) `test_device_math_bits_equal_the_gcc_build_exhaustively` (GPU): for all 14 routines (and three probes of fgnn_rng.h: Philox, the uint32 -> [0,1) map, the channel thresholds) the bits the device build returns equal
      the bits the gcc build returns for EVERY input of the domain (up to 2^32 - 2^24 floats each), compared through per-window
      checksums (tests/math_bits_exhaustive.hip on the device, og_math_checksums in the oracle).

What the routines restate: decoding_q.py:265-273 (softplus, reduce_logsumexp), :365-373 (_phi), :313-363 (tanh / atanh of the
'boxplus' rule), feedback_gnn.py:139-141 (mean over three edges), gnn.py:63-69 (tanh / sigmoid activations), gnn.py:333-338."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O


def _b(x):
    return int(np.float32(x).view(np.uint32))


POS, NEG = (0x00000000, 0x7f7fffff), (0x80000000, 0xff7fffff)  # every finite float32, by sign (bit patterns ascend with magnitude)
# bit-pattern ranges of the inputs each routine can be handed on this path (NaN / inf never reach them: DESIGN.md §3)
DOMAINS = {
    "exp": [(0, _b(88.0)), (0x80000000, _b(-87.0))],               # callers clamp to [-87, 13.94], [-20, 0], [8.5e-8, 16.64], [-87, 0]
    "log": [(0x00800000, 0x7f7fffff)],                              # normal positive floats
    "log1p": [(0, _b(2.0 ** 24))],                                  # u = exp(.) <= e^16.64 < 2^24
    "softplus": [POS, NEG], "phi": [POS, NEG], "phi_gnn": [POS, NEG], "tanh": [POS, NEG], "sigmoid": [POS, NEG],
    "atanh": [(0, 0x3f7ffffe), (0x80000000, 0xbf7ffffe)],           # |x| <= 1 - 2^-23 (decoding_q.py:356-358 clips to that)
    "lse2_corr": [POS, NEG],                                        # fg_lse2_corr(x, 0): the part of reduce_logsumexp that depends on a - b
    "lse2_1": [POS, NEG],                                           # fg_lse2(x, 1): with the max term
    "div3": [POS, NEG], "rcp_unit": [POS, NEG],
    "div_atanh": [(0, 0x3f7ffffe)],
    # fgnn_rng.h, shared the same way: a Philox4x32-10 block keyed and countered by the input word (all 2^32 words), the uint32 -> [0,1)
    # map, and the float32 depolarizing thresholds of every p in [0, 1] (pauli.py:100-108 with px = pz = 2p/3, py = p/3)
    "philox": [(0, 0xffffffff)], "u32_to_unit": [(0, 0xffffffff)], "pauli_thr": [(0, 0x3f800000)],
}
RNG_PROBES = ("philox", "u32_to_unit", "pauli_thr")
PHILOX_KAT = ["6627e8d5 e169c58d bc57ac4c 9b00dbd8", "408f276d 41c83b0e a20bc7c6 6d5451fd", "d16cfe09 94fdcceb 5001e420 24126ea1"]  # Random123 kat_vectors
CHUNK_LOG2 = 22


def test_checksum_probe_is_the_elementwise_function():
    """og_math_checksums (what both exhaustive tests stand on) = sums over og_math_apply's outputs, window by window."""
    lo, hi = 0x3f7ff000, 0x40801234  # straddles three 2^22 windows
    x = np.arange(lo, hi + 1, dtype=np.uint32)
    for name in ("tanh", "phi", "log1p", "lse2_1", "u32_to_unit"):
        if name == "u32_to_unit":  # an integer-valued probe: its own NumPy restatement (TensorFlow's Uint32ToFloat)
            y = (((x >> 9) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1.0)).view(np.uint32).astype(np.uint64)
        else:
            y = O.math_apply(name, x.view(np.float32)).view(np.uint32).astype(np.uint64)
        c = O.math_checksums(name, lo, hi, CHUNK_LOG2)
        win = (x >> CHUNK_LOG2) - (lo >> CHUNK_LOG2)
        assert c.shape == (int(win.max()) + 1, 2)
        for k in range(c.shape[0]):
            sel = win == k
            assert int(c[k, 0]) == int(y[sel].sum(dtype=np.uint64))
            assert int(c[k, 1]) == int((y[sel] * (x[sel] | 1).astype(np.uint64)).sum(dtype=np.uint64))
    assert sorted(O.MATH_FUNCTIONS) == sorted(DOMAINS)


# (routine, bit range, bound in float32 ulps of the exact value); measured maxima in the comments (profiles/r5_math_ulp_exhaustive.txt)
ULP_SWEEPS = [
    ("exp", (0, _b(88.0)), 0.92),                      # 0.9091 at 5.1997986
    ("exp", (0x80000000, _b(-87.0)), 0.92),            # 0.8822
    ("log", (0x00800000, 0x7f7fffff), 1.21),           # 1.1962 at 0.991808
    ("log1p", (0, _b(2.0 ** 24)), 1.30),               # 1.2852
    ("tanh", (0, _b(9.5)), 5.8),                       # 5.7508 at 6.35 (3.8 on |x| <= 2); odd symmetry: tests/test_math.py
    ("atanh", (0, 0x3f7ffffe), 2.45),                  # 2.3940; the sign is copied
    ("sigmoid", (0, _b(88.0)), 1.55),                  # 1.4983 (beyond 88 the exponential is clamped and the result is 1)
    ("sigmoid", (0x80000000, _b(-87.0)), 2.45),        # 2.4019 (e / (1 + e)); below -87 the exponential is clamped (1.6e-38 absolute)
    ("softplus", (0, _b(14.0)), 1.80),                 # 1.7388 (above the threshold 13.94 the result is the argument itself)
    ("softplus", (0x80000000, _b(-87.0)), 7.5),        # 7.3645 at -13.9427: tf2xla's own rule returns exp(t) below the threshold
                                                       # (y - log1p(y) = y^2 / 2 = 6.9 ulp there); below -87 exp is flushed to 0 (1.6e-38)
    # x / 3 and 1 / t are correctly rounded (0.33 / 0.50 ulp over all finite inputs, profiles/r5_math_ulp_exhaustive.txt); the GPU suite
    # holds the device sequences to IEEE division exhaustively (tests/div_exhaustive.hip), so one binade each suffices here
    ("div3", (_b(1.0), _b(2.0)), 0.5), ("rcp_unit", (_b(0.5), _b(1.0)), 0.5),
]


def test_shared_math_against_float64_libm_exhaustively():
    """Every float32 of each routine's domain against double-precision libm: a wrong constant cannot hide behind the shared header."""
    lines = []
    for name, (lo, hi), bound in ULP_SWEEPS:
        _, worst, at = O.math_checksums(name, lo, hi, CHUNK_LOG2, ulp=True)
        lines.append(f"{name:9s} bits 0x{lo:08x}..0x{hi:08x} {hi - lo + 1:11d} floats  max err {worst:.4f} ulp at x = "
                     f"{float(np.uint32(at).view(np.float32))!r}")
        assert 0 <= worst <= bound, lines[-1]
    out = os.environ.get("FGNN_MATH_ULP_REPORT")
    if out:
        open(out, "w").write("\n".join(lines) + "\n")


@pytest.mark.gpu
def test_device_math_bits_equal_the_gcc_build_exhaustively():
    """hipcc/gfx950 bits == gcc/x86 bits for every input of every routine's domain (5.7e10 evaluations per side), and the device's Philox4x32-10 on the Random123 known-answer vectors."""
    import __graft_entry__ as entry
    exe = entry.build_math_bits_exhaustive()
    out = os.path.join(os.path.dirname(exe), "math_bits.bin")
    jobs = [(name, lo, hi) for name, rs in DOMAINS.items() for lo, hi in rs]
    args = [f"{O.MATH_FUNCTIONS[n]}:0x{lo:08x}:0x{hi:08x}" for n, lo, hi in jobs]
    res = subprocess.run([exe, out, str(CHUNK_LOG2)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout
    for i, kat in enumerate(PHILOX_KAT):  # the DEVICE's Philox on the Random123 known-answer inputs
        assert f"philox_kat {i}: {kat}" in res.stdout, res.stdout
    dev = np.fromfile(out, dtype=np.uint64).reshape(-1, 2)
    pos, total, report = 0, 0, []
    for name, lo, hi in jobs:
        host = O.math_checksums(name, lo, hi, CHUNK_LOG2)
        d = dev[pos:pos + host.shape[0]]
        pos += host.shape[0]
        bad = np.nonzero((d != host).any(1))[0]
        first = (int(bad[0]) + (lo >> CHUNK_LOG2)) << CHUNK_LOG2 if bad.size else None
        report.append(f"{name:9s} bits 0x{lo:08x}..0x{hi:08x} {hi - lo + 1:11d} inputs, {host.shape[0]:4d} windows: "
                      + ("device == gcc" if not bad.size else f"{bad.size} windows differ, first window starts at bits 0x{first:08x}"))
        assert not bad.size, "\n".join(report)
        total += hi - lo + 1
    assert pos == dev.shape[0] and total > 5.6e10
    rep = os.environ.get("FGNN_MATH_BITS_REPORT")
    if rep:
        open(rep, "w").write("\n".join(report) + f"\ntotal {total} inputs per side, window = 2^{CHUNK_LOG2} bit patterns, "
                             "checksums: sum(bits), sum(bits * (input | 1)) mod 2^64\n")

#!/usr/bin/env python3
"""profiles/traffic.json from tools/pmc_summary.py summaries (the rocprofv3 --pmc passes of tools/refresh_traffic.sh): HBM bytes and
VALU / MFMA wave-instructions per launch of the dominant kernel of every BASELINE.json configuration at its per-GPU shard shape —
the 64-iteration BP4 kernel of the first decoder and the feedback-GNN kernel (c3: [[882,24]] x 65 536; c4: [[1270,28]] x 32 768) and
the GNN_BP4 kernel (c5: [[1270,28]] x 16 384, 10 iterations).  bench.py quotes these under roofline.traffic /
roofline.valu_wave_insts_per_launch with their source: they are offline measurements, not taken by the bench run itself.  Every entry
carries the sha256 of the kernel sources (csrc_sha256) and of the library binary (lib_sha256) it was measured on; bench.py refuses an
entry whose sources differ from the tree's.

    python tools/make_traffic_json.py <tag> sandwich:ghp882:65536:64=<summary> sandwich:ghp1270:32768:64=<summary> \
                                            gnnbp4:ghp1270:16384:10=<summary> > profiles/traffic.json
A `merge=<traffic.json>` argument keeps the entries of an existing file that this call does not re-measure (a refresh of one
configuration after only its kernel changed).
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feedback_gnn_amd import _lib  # host-only helpers: fingerprints of the kernel sources / of the built library

tag = sys.argv[1]
LIB_SHA = _lib.library_sha256()
FETCH_NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 "
              "reports half the bytes of wide reads; byte-granular reads are uncalibrated, so this is an upper bound); WRITE_SIZE is exact")


def read_rows(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+grid=\s*(\d+)\s+wg=\s*(\S+)\s+(\S+)\s+mean=(\S+)\s+n=(\d+)\s+avg_ms=(\S+)", line)
        if m:
            rows.setdefault((m.group(1).strip(), m.group(3)), {})[m.group(4)] = (float(m.group(5)), float(m.group(7)))
    return rows


def pick(rows, prefix, accept=lambda name: True):
    """The (kernel, workgroup/duration bucket) row with the longest launches among the kernels whose name starts with `prefix`."""
    cands = [(k, v) for k, v in rows.items() if k[0].startswith(prefix) and "FETCH_SIZE" in v and accept(k[0])]
    cands.sort(key=lambda kv: kv[1]["FETCH_SIZE"][1], reverse=True)
    if not cands:
        return None
    (name, wg), v = cands[0]
    # pmc_summary.py buckets launches by duration (wg=<threads>/b<log2 bucket>): a launch near a bucket boundary (the 0.15 ms launch of
    # c1) may land in the neighbouring bucket in another counter's pass — take a counter this row lacks from the same kernel and
    # workgroup size in the bucket whose launches took the closest time
    v = dict(v)
    threads = wg.split("/")[0]
    for (n2, wg2), v2 in rows.items():
        if n2 == name and wg2 != wg and wg2.split("/")[0] == threads:
            for ctr, (val, ms) in v2.items():
                if ctr not in v and abs(ms - v["FETCH_SIZE"][1]) <= 0.1 * v["FETCH_SIZE"][1]:
                    v[ctr] = (val, ms)
    return (name, wg), v


def entry(kind, name, v):
    fetch_kb, write_kb = v["FETCH_SIZE"][0], v["WRITE_SIZE"][0]
    return {"kernel": name, "taken_at": tag, "csrc_sha256": _lib.source_fingerprint(kind), "lib_sha256": LIB_SHA,
            "avg_ms_under_pmc": v["SQ_INSTS_VALU"][1] if "SQ_INSTS_VALU" in v else None,
            "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
            "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024), "note": FETCH_NOTE,
            "valu_wave_insts_per_launch": v.get("SQ_INSTS_VALU", (None,))[0],
            "valu_note": "SQ_INSTS_VALU of the same launch; a wave64 VALU instruction occupies a SIMD-32 for 2 cycles",
            "mfma_insts_per_launch": v.get("SQ_INSTS_MFMA", (None,))[0], "mfma_busy_cycles": v.get("SQ_VALU_MFMA_BUSY_CYCLES", (None,))[0],
            "wait_inst_any": v.get("SQ_WAIT_INST_ANY", (None,))[0], "wave_cycles": v.get("SQ_WAVE_CYCLES", (None,))[0],
            "lds_idx_active": v.get("SQ_LDS_IDX_ACTIVE", (None,))[0], "lds_bank_conflict": v.get("SQ_LDS_BANK_CONFLICT", (None,))[0],
            "grbm_gui_active": v.get("GRBM_GUI_ACTIVE", (None,))[0]}


out = {}
for spec in sys.argv[2:]:
    what, path = spec.split("=", 1)
    if what == "merge":
        out.update(json.load(open(path)))
        continue
    kind, code, B, iters, *cn = what.split(":")  # optional fifth field: the check-node rule when it is not 'boxplus-phi' (bench.py's key suffix)
    suffix = f"_{cn[0]}" if cn and cn[0] != "boxplus-phi" else ""
    rows = read_rows(path)
    if kind == "sandwich":
        # the FIRST decoder's launch (constant channel LLR: template argument NQ = 0) — the later decoders of a sandwich carry
        # their per-qubit channel LLRs in registers (NQ = 4 / 5) and are a different instantiation
        # (template tail: ..., NQ, TRACE, GMEM> — GMEM appeared in round 5; the round-4 summaries end in NQ, TRACE>)
        bp = pick(rows, "bp4_kernel", lambda nm: re.search(r",\s*0,\s*false(,\s*false)?>$", nm) is not None)
        if bp:
            (name, wg), v = bp
            out[f"bp4_{code}_it{iters}_B{B}{suffix}"] = entry("bp4", name, v)
        gn = pick(rows, "gnn_stream_kernel") or pick(rows, "gnn_mfma_kernel")  # the streaming VALU kernel is the default of the factored order
        if gn:
            (name, wg), v = gn
            out[f"gnn_{code}_B{B}"] = entry("gnn", name, v)
    elif kind == "gnnbp4":
        gb = pick(rows, "gnn_bp4")
        if gb:
            (name, wg), v = gb
            out[f"gnnbp4_{code}_it{iters}_B{B}"] = entry("gnnbp4", name, v)
    else:
        raise SystemExit(f"unknown kind {kind}")
print(json.dumps(out, indent=1))

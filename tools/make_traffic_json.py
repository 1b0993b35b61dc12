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
A spec is kind:code:B:iters[:cn_type[:forms]]; forms = "literal" (default: the library's default operation sequence, keys without a
suffix) or "reassociated" (the opt-in forms: keys get `_shared` (bp4) / `_factored` (gnn, gnnbp4), as bench.pmc_key names them).
A `merge=<traffic.json>` argument keeps the entries of an existing file that this call does not re-measure and that were measured on
the current kernel sources (a refresh of one configuration after only its kernel changed); stale entries are dropped.
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
        # keep what this call does not re-measure — but only entries measured on the CURRENT kernel sources: bench.py refuses any other
        # (pmc_entry), so a stale entry is dead weight that reads like evidence
        for k, e in json.load(open(path)).items():
            if e.get("csrc_sha256") == _lib.source_fingerprint(k.split("_")[0]):
                out[k] = e
        continue
    kind, code, B, iters, *rest = what.split(":")  # optional: the check-node rule when it is not 'boxplus-phi', then the forms the pass ran
    cn = rest[0] if rest and rest[0] else "boxplus-phi"
    reassoc = len(rest) > 1 and rest[1] == "reassociated"
    suffix = "" if cn == "boxplus-phi" else f"_{cn}"
    rows = read_rows(path)
    if kind == "sandwich":
        # the FIRST decoder's launch (constant channel LLR: template argument NQ = 0) — the later decoders of a sandwich carry
        # their per-qubit channel LLRs in registers (NQ = 4 / 5) and are a different instantiation
        # (template tail: ..., NQ, TRACE, GMEM, LSE> — LSE appeared in round 6: 0 = literal, 1 = shared log-sum-exp (compile-time forms of
        # the (3,3,6) phi kernels), 2 = chosen at run time; GMEM in round 5; the round-4 summaries end in NQ, TRACE>)
        want_lse = ("1", "2") if reassoc else ("0", "2")
        def is_first_decoder(nm):
            m = re.search(r",\s*0,\s*false,\s*false,\s*([012])>$", nm)
            return m is not None and m.group(1) in want_lse
        bp = pick(rows, "bp4_kernel", is_first_decoder)
        if bp:
            (name, wg), v = bp
            out[f"bp4_{code}_it{iters}_B{B}{suffix}" + ("_shared" if reassoc else "")] = entry("bp4", name, v)
        # the streaming VALU kernel runs launches of 4 096 codewords or more (either association: <DV, LITERAL, EMBPK>), the MFMA tiles the rest
        want_lit = "false" if reassoc else "true"
        gn = (pick(rows, "gnn_stream_kernel", lambda nm: re.search(r"<\d+,\s*" + want_lit + r",", nm) is not None)
              or pick(rows, "gnn_mfma_kernel", lambda nm: re.search(r",\s*" + ("true" if reassoc else "false") + r">$", nm) is not None))
        if gn:
            (name, wg), v = gn
            out[f"gnn_{code}_B{B}" + ("_factored" if reassoc else "")] = entry("gnn", name, v)
    elif kind == "gnnbp4":
        gb = pick(rows, "gnn_bp4")
        if gb:
            (name, wg), v = gb
            out[f"gnnbp4_{code}_it{iters}_B{B}" + ("_factored" if reassoc else "")] = entry("gnnbp4", name, v)
    else:
        raise SystemExit(f"unknown kind {kind}")
print(json.dumps(out, indent=1))

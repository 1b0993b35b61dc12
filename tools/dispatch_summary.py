"""Per-launch view of a rocprofv3 kernel trace: the BP4 symbol serves both sandwich stages, so --stats averages a 64- and a
16-iteration launch; this lists kernels grouped by (name, duration bucket).   python tools/dispatch_summary.py <..._kernel_trace.csv>"""
import collections, csv, math, re, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        m = re.search(r'::(\w+(?:<[^>]*>)?)', r['Kernel_Name'])
        name = m.group(1) if m else r['Kernel_Name'][:48]
        ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        agg[(name, int(round(math.log2(max(ms, 1e-4)) * 2)))].append(ms)
    print(f"# {path}")
    for (name, _), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name:34s} launches={len(v):3d} avg_ms={sum(v)/len(v):9.4f} min={min(v):9.4f} max={max(v):9.4f} total_ms={sum(v):9.3f}")

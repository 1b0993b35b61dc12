"""Summarise rocprofv3 --pmc counter_collection.csv files: one line per (kernel, counter), averaged over dispatches
of the same kernel+grid.   python tools/pmc_summary.py gpurun_out/pmc_*/runc/*_counter_collection.csv"""
import collections
import csv
import re
import sys

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in rows:
        m = re.search(r'::(\w+(?:<[^>]*>)?)', r['Kernel_Name'])
        name = m.group(1) if m else r['Kernel_Name'][:40]
        d_ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        # bp4 launches of different iteration counts share a name: bucket by duration (x2 steps)
        import math
        bucket = int(round(math.log2(max(d_ms, 1e-3)) * 2))
        key = (name, r['Grid_Size'], f"{r['Workgroup_Size']}/b{bucket}")
        agg[key + (r['Counter_Name'],)].append(float(r['Counter_Value']))
        dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    print(f"# {path}")
    for k in sorted(agg):
        v = agg[k]
        d = dur[k[:3]]
        print(f"{k[0]:28s} grid={k[1]:>9s} wg={k[2]:>4s} {k[3]:26s} mean={sum(v)/len(v):.6e} n={len(v)} avg_ms={sum(d)/len(d):.3f}")

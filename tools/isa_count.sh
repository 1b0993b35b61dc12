#!/bin/bash
# VALU / LDS instruction mix of the loop bodies of a kernel:  tools/isa_count.sh <file.hip> <mangled-name-substring>
# (compiles the device side to assembly, cuts out the kernel, prints the per-loop instruction histogram)
set -e
SRC=$1; PAT=$2; OUT=${3:-/tmp/isa}
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -S --cuda-device-only -o $OUT/k_all.s $SRC 2>/dev/null
python3 - "$OUT/k_all.s" "$PAT" <<'PY'
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and pat in l and l.rstrip().endswith(("E", ":")) or (l.startswith("_Z") and pat in l and ":" in l))
end = next(i for i in range(start, len(txt)) if ".end_amdhsa_kernel" in txt[i] or txt[i].startswith("\t.section") and i > start + 10)
body = txt[start:end]
# split at labels; report blocks with > 100 instructions
blocks, cur, name = [], [], "entry"
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur)); cur = []; name = l.split(":")[0]
    elif re.match(r"^\s+[a-z]", l):
        cur.append(l.split()[0])
blocks.append((name, cur))
tot = collections.Counter()
for name, ins in blocks:
    if len(ins) < 100:
        continue
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    print(f"{name}: {len(ins)} instructions, VALU {valu}, LDS {lds}, SALU {sum(v for k,v in c.items() if k.startswith('s_'))}")
    print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
PY
grep -A30 "$PAT" $OUT/k_all.s | grep -m3 -E "vgpr_count|sgpr_count|lds_size" || true

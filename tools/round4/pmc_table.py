#!/usr/bin/env python3
"""MEASUREMENT TOOL: per-kernel means of rocprofv3 --pmc passes (p_counter_collection.csv under the given directories), largest grid only."""
import collections, csv, glob, re, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*\)\s*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].strip()))
            per[k].append((int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
        for k, rs in per.items():
            gmax = max(g for g, _, _ in rs)
            for g, c, v in rs:
                if g >= 0.98 * gmax: rows[k][c].append(v)
for k in sorted(rows):
    if not k.startswith("k_"): continue
    print(k)
    for c in sorted(rows[k]):
        v = rows[k][c]; print(f"    {c:32s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")

#!/usr/bin/env python3
"""MEASUREMENT TOOL: per-kernel duration over the dispatches with the kernel's largest grid (the 8-chunk launches) from a rocprofv3
--kernel-trace directory.  usage: kt_largest.py DIR [prefix]"""
import collections, csv, glob, re, sys
per = collections.defaultdict(list)
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*\)\s*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].strip()))
        per[n].append((int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
pre = sys.argv[2] if len(sys.argv) > 2 else "k_"
tot = 0.0
for k, v in sorted(per.items()):
    if not k.startswith(pre):
        continue
    gmax = max(g for g, _ in v); t = sorted(x for g, x in v if g >= 0.98 * gmax)
    print(f"   {k:34s} grid {gmax:9d} x{len(t):4d}  avg {sum(t) / len(t):8.1f} us  median {t[len(t) // 2]:8.1f}  min {t[0]:8.1f}  max {t[-1]:8.1f}")
    if k.startswith("k_prep"): tot += sum(t) / len(t)
print(f"   sum of k_prep* averages {tot:.1f} us")

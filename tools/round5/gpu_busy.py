#!/usr/bin/env python3
"""MEASUREMENT TOOL: how busy the device is over a command's run, from a rocprofv3 --kernel-trace directory: the span from the first kernel's
start to the last one's end, the time with at least one kernel running (union of the intervals), the time with a k_inflate running, and
per kernel the number of dispatches and the sum of their durations.  usage: gpu_busy.py DIR [wall_seconds]"""
import collections, csv, glob, json, re, sys
iv = []; per = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*\)\s*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].strip()))
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        iv.append((a, b, n)); per[n][0] += 1; per[n][1] += (b - a) / 1e9
def union(xs):
    t = 0; end = None
    for a, b in sorted(xs):
        if end is None or a > end: t += b - a; end = b
        elif b > end: t += b - end; end = b
    return t / 1e9
span = (max(b for _, b, _ in iv) - min(a for a, _, _ in iv)) / 1e9
out = {"kernels": len(iv), "span_first_to_last_kernel_s": round(span, 4), "some_kernel_running_s": round(union([(a, b) for a, b, _ in iv]), 4),
       "k_inflate_running_s": round(union([(a, b) for a, b, n in iv if n.startswith("k_inflate")]), 4),
       "preparation_or_pileup_running_s": round(union([(a, b) for a, b, n in iv if n.startswith("k_prep") or n.startswith("k_pileup")]), 4),
       "per_kernel": {k: {"dispatches": v[0], "sum_s": round(v[1], 4)} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]}}
if len(sys.argv) > 2: out["wall_s"] = float(sys.argv[2])
print(json.dumps(out, indent=1))

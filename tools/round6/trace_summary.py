#!/usr/bin/env python3
# round 6: a kernel trace of the command (rocprofv3 --kernel-trace, csv) reduced to: the inflate phase, how long some kernel / k_inflate ran in it, the idle gaps, per-kernel sums
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r['Start_Timestamp']) for r in rows)
ev = sorted([(int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name'].split('(')[0], r['Queue_Id']) for r in rows])
def union(evs):
    tot = 0; cs = ce = None
    for s, e in sorted(evs):
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)
inf = [(s, e) for s, e, n, q in ev if n == 'k_inflate']
first = inf[0][0]; last = max(e for s, e in inf)
print("inflate phase %.1f..%.1f ms (%.1f ms), %d launches, k_inflate running %.1f ms (sum of durations %.1f ms)" % (first / 1e6, last / 1e6, (last - first) / 1e6, len(inf), union(inf) / 1e6, sum(e - s for s, e in inf) / 1e6))
allk = sorted([(s, e) for s, e, n, q in ev if first <= s <= last])
print("some kernel running in the phase: %.1f ms" % (union(allk) / 1e6))
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in ev: by[n][0] += 1; by[n][1] += e - s
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:14]: print("  %-40s %5d %8.2f ms, %.3f each" % (n[:40], c, t / 1e6, t / c / 1e6))
cur = allk[0][1]; gaps = []
for s, e in allk:
    if s > cur + 5e5: gaps.append((round(cur / 1e6, 1), round((s - cur) / 1e6, 2)))
    cur = max(cur, e)
print("idle gaps > 0.5 ms: %d, %.1f ms in sum" % (len(gaps), sum(g for _, g in gaps))); print(gaps[:60])
qs = collections.defaultdict(set)
for s, e, n, q in ev: qs[n].add(q)
print({k[:22]: sorted(v) for k, v in qs.items() if 'rocclr' not in k})

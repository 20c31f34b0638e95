#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/gpu_round.sh (under gpurun_out/) into the small summaries kept in profiles/.
One entry per kernel, keyed by its FULL name (template arguments included: round 2's summary matched "k_pileup_multi" by prefix and so
averaged the CpG-only and the dense-context instantiation), and one per kernel family of bench.py's step."""
import collections, csv, glob, json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def find(d, name):
    g = glob.glob(f"{src}/{tag}_{d}/**/{name}", recursive=True)
    return g[0] if g else None


def canon(kernel_name):
    n = re.sub(r"^void\s+", "", kernel_name.strip())
    n = re.sub(r"\(.*\)\s*$", "", n)               # the argument list
    return n.replace(", ", ",").replace(" ", "")


kt = find("prof_kt", "kt_kernel_stats.csv")
if kt:
    shutil.copy(kt, f"{dst}/{tag}_rocprofv3_kernel_stats.csv")
for f in ("bench.json", "bench_under_rocprof.json", "pytest_gpu.log", "smoke.log"):
    if os.path.exists(f"{src}/{tag}_{f}"):
        shutil.copy(f"{src}/{tag}_{f}", f"{dst}/{tag}_{f}")

# A kernel is launched over ONE chunk (while the intervals are loaded; the single-chunk legs) and over EIGHT (bench.py's step).  What the
# bench line quotes is the eight-chunk launch, so the figures below are taken over the dispatches with the kernel's LARGEST grid only (the
# 16 resident intervals give two different groups of eight, whose grids differ by a few workgroups: "largest" = within 2 % of the maximum).
rows = collections.defaultdict(list)                                              # kernel -> (grid, counter, value)
for d in ("sq", "fetch", "write_lds", "cache"):
    f = find("prof_pmc_" + d, "p_counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            rows[canon(r["Kernel_Name"])].append((int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))          # kernel -> counter -> values per (largest-grid) dispatch
for k, rs in rows.items():
    gmax = max(g for g, _, _ in rs)
    for g, c, v in rs:
        if g >= 0.98 * gmax:
            agg[k][c].append(v)

# durations of the same dispatches from the kernel trace of the --stats run
dur = {}
ktf = find("prof_kt", "kt_kernel_trace.csv")
if ktf:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(ktf)):
        per[canon(r["Kernel_Name"])].append((int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    with open(f"{dst}/{tag}_rocprofv3_kernel_stats_largest_grid.csv", "w") as o:
        o.write("kernel,grid_size,dispatches,avg_us,median_us,min_us,max_us\n")
        for k, v in sorted(per.items()):
            if not k.startswith("k_"):
                continue
            gmax = max(g for g, _ in v); t = [x for g, x in v if g >= 0.98 * gmax]
            dur[k] = {"grid": gmax, "dispatches": len(t), "avg_us": sum(t) / len(t), "median_us": sorted(t)[len(t) // 2], "min_us": min(t), "max_us": max(t)}
            o.write(f"\"{k}\",{gmax},{len(t)},{sum(t) / len(t):.2f},{sorted(t)[len(t) // 2]:.2f},{min(t):.2f},{max(t):.2f}\n")

kernels = {}
for name, cs in agg.items():
    if not name.startswith("k_"):
        continue
    e = {"name": name, "chunks_per_launch": chunks, "counters": {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for k, v in cs.items()}}
    e["dispatches"] = max(len(v) for v in cs.values())
    if name in dur:
        e["duration_us"] = dur[name]
    if "FETCH_SIZE" in cs:
        fr = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024
        wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024 if "WRITE_SIZE" in cs else 0.0
        e["hbm_bytes_per_launch"] = {"fetch_raw": fr, "write_raw": wr, "fetch_x2_plus_write": 2 * fr + wr}
    kernels[name] = e
fam = {"pileup": ["k_pileup_multi<false,false>"], "preparation": ["k_prep_zero", "k_prep_scan", "k_prep_segs"], "pileup_dense": ["k_pileup_multi<false,true>"]}
families = {}
for fn, ks in fam.items():
    have = [kernels[k] for k in ks if k in kernels and "hbm_bytes_per_launch" in kernels[k]]
    if len(have) == len(ks):
        families[fn] = {"kernels": ks, "dispatches": [k["dispatches"] for k in have], "chunks_per_launch": chunks,
                        "duration_us_sum_of_kernel_averages": sum(dur[k]["avg_us"] for k in ks if k in dur) if all(k in dur for k in ks) else None,
                        "hbm_bytes_per_launch": {x: sum(k["hbm_bytes_per_launch"][x] for k in have) for x in ("fetch_raw", "write_raw", "fetch_x2_plus_write")}}
out = {"command": "python bench.py --no-cpu-baseline --steps 2 --warmup 1 --passes 4 (16 resident 1 Mb intervals in rotation, ~0.9 GB of records; one rocprofv3 --pmc pass per counter group); only the dispatches with each kernel's largest grid (the 8-chunk launches) are averaged",
       "unit_note": "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; fetch_x2_plus_write applies MI355X_MICROARCH.md's gfx950 correction (128-byte requests tallied as 64) to the fetch side only",
       "kernels": kernels, "families": families}
json.dump(out, open(f"{dst}/{tag}_rocprofv3_pmc_summary.json", "w"), indent=1)
if kt:
    print(open(f"{dst}/{tag}_rocprofv3_kernel_stats.csv").read())
print(json.dumps({k: v for k, v in families.items()}, indent=1)[:3000])

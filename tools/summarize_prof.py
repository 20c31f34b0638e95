#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/gpu_round.sh (under gpurun_out/) into the small summaries kept in profiles/."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/{tag}_prof_kt/kt_kernel_stats.csv", f"{dst}/{tag}_rocprofv3_kernel_stats.csv")
for f in ("bench.json", "bench_under_rocprof.json", "pytest_gpu.log", "smoke.log"):
    if os.path.exists(f"{src}/{tag}_{f}"):
        shutil.copy(f"{src}/{tag}_{f}", f"{dst}/{tag}_{f}")
pmc = {}
for d in ("sq", "fetch", "write_lds", "cache"):
    f = f"{src}/{tag}_prof_pmc_{d}/p_counter_collection.csv"
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_pileup" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
out = {"kernel": "k_pileup<false>", "command": "python bench.py --steps 20 --warmup 5 --no-cpu-baseline (one rocprofv3 --pmc pass per counter group)", "counters": pmc}
if "FETCH_SIZE" in pmc:
    fs, ws = pmc["FETCH_SIZE"]["mean_per_dispatch"], pmc.get("WRITE_SIZE", {"mean_per_dispatch": 0})["mean_per_dispatch"]
    out["hbm_traffic_bytes_per_launch"] = {
        "fetch_raw": fs * 1024, "write_raw": ws * 1024,
        "fetch_x2_gfx950_correction": fs * 2048,
        "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x (MI355X_MICROARCH.md, HBM section); "
                "this kernel's reads are byte-granular so the factor is uncalibrated: the truth lies between fetch_raw and fetch_x2. After the first "
                "iteration the whole 50 MB working set sits in the 256 MiB Infinity Cache, whose hits these counters include."}
json.dump(out, open(f"{dst}/{tag}_rocprofv3_pmc_summary.json", "w"), indent=1)
print(open(f"{dst}/{tag}_rocprofv3_kernel_stats.csv").read())
print(json.dumps(out, indent=1)[:1500])

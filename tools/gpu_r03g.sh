#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03g_prof_kt -o kt -- python $R/tools/prep_bench.py 16 > $O/r03g_prep_bench.json 2> /dev/null
cat $O/r03g_prep_bench.json; find $O/r03g_prof_kt -name "*kernel_stats.csv" | head -1 | xargs cat
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("/root/repo/gpurun_out/r03g_prof_kt/**/kt_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# durations by kernel and grid size
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].split("(")[0]
    if "prep" in n or "pileup" in n:
        agg[(n,r.get("Grid_Size_X") or r.get("Grid_Size"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(agg.items()):
    print(k, len(v), "avg_us %.1f" % (sum(v)/len(v)/1e3), "min_us %.1f" % (min(v)/1e3))
PY

#!/bin/bash
# round 3, GPU call: where the preparation's time goes now -- per-kernel trace and counters of tools/prep_bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/tools/prep_bench.py 16"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pj_kt -o kt -- $B > /dev/null 2> $O/r03j_prof.err
f=$(find /tmp/pj_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/r03j_prep_kernel_stats.csv && head -12 "$f" | cut -c1-220
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU -d /tmp/pj_sq -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pj_f -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d /tmp/pj_w -o p -- $B > /dev/null 2>&1
python - <<'PY' | tee $O/r03j_prep_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pj_sq", "pj_f", "pj_w"):
    for f in glob.glob(f"/tmp/{d}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            if n.startswith("k_prep") or n.startswith("void k_prep") or "k_pileup" in n: agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in agg.items():
    print(n)
    for k, v in sorted(cs.items()): print(f"   {k:32s} mean {sum(v)/len(v):16.1f} max {max(v):16.1f} n {len(v)}")
PY

#!/bin/bash
# round 3, GPU call: what the name-table compare-and-swaps cost k_prep_scan (a build that stores instead, -DPREP_EXP_NOATOMIC: timing only, wrong
# results).  Kept as the record of how profiles/r03aa_prep_noatomic.txt was made: the switch was removed from csrc/mdk_prep.hip afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PREP_BENCH_FAST=1
for v in "" noatomic; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  rm -rf /tmp/pa_kt /tmp/pa_w /tmp/pa_f
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pa.json 2> /dev/null
  f=$(find /tmp/pa_kt -name '*kernel_stats.csv' | head -1)
  echo "== variant [${v:-default}] $(python -c "import json; d=json.load(open('/tmp/pa.json')); print('prep us/chunk', round(d['prep_ms_per_chunk_8_per_launch']*1000,1))")"
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pa_w -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pa_f -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pa_w", "pa_f"):
    for f in glob.glob(f"/tmp/{d}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            if n.startswith("k_prep_s"): agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in agg.items():
    for k, v in sorted(cs.items()): print(f"   {n:14s} {k:12s} max {max(v)/1e3:10.1f} MB")
PY
done 2>&1 | tee $O/r03aa_prep_noatomic.txt

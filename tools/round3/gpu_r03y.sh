#!/bin/bash
# round 3, GPU call: k_prep_scan writes its PrepReads through an LDS stage as whole lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py tests/test_gpu_mbias.py tests/test_gpu_perread.py -m gpu -x -q > $O/r03y_pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/r03y_pytest.log
cd /tmp; export TMPDIR=/tmp PREP_BENCH_FAST=1
for v in 1 2; do
  rm -rf /tmp/py_kt
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/py_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/py.json 2> /dev/null
  f=$(find /tmp/py_kt -name '*kernel_stats.csv' | head -1)
  echo "== run $v $(python -c "import json; d=json.load(open('/tmp/py.json')); print('prep us/chunk', round(d['prep_ms_per_chunk_8_per_launch']*1000,1))")"
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
done 2>&1 | tee $O/r03y_prep_stage.txt
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/py_w -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/py_f -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1
python - <<'PY' | tee -a $O/r03y_prep_stage.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("py_w", "py_f"):
    for f in glob.glob(f"/tmp/{d}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            if n.startswith("k_prep"): agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in agg.items():
    for k, v in sorted(cs.items()): print(f"{n:14s} {k:12s} max {max(v)/1e3:10.1f} MB  n {len(v)}")
PY

#!/bin/bash
# round 5, call c: the hardened tree + the new preparation (scan without inter-workgroup dependency, links instead of table walks):
# whole -m gpu suite, then the preparation's per-kernel times with and without the LDS window
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
( cd /tmp; export TMPDIR=/tmp
for v in "" stage0; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  PREP_BENCH_FAST= timeout 200 python $R/tools/prep_bench.py 16 > /tmp/pl_full.json 2> /tmp/pl_full.err; echo "== variant [${v:-default}] $(cat /tmp/pl_full.json)"; tail -2 /tmp/pl_full.err
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  f=$(find /tmp/pl_kt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
done ) 2>&1 | tee $O/prep_variants.txt

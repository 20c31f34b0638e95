#!/bin/bash
# round 3, GPU call: what the preparation kernels spend their time on -- builds with parts switched off (-DPREP_EXP=n, timing only).
# Kept as the record of how profiles/r03l_prep_experiments.txt was made: the PREP_EXP switches were removed from csrc/mdk_prep.hip afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PREP_BENCH_FAST=1
for v in "" e1 e2 e4 e7 e8 e32 e40; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  rm -rf /tmp/pl_kt
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  f=$(find /tmp/pl_kt -name '*kernel_stats.csv' | head -1)
  echo "== variant [${v:-default}] $(python -c "import json; d=json.load(open('/tmp/pl.json')); print('prep us/chunk', round(d['prep_ms_per_chunk_8_per_launch']*1000,1))")"
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
done 2>&1 | tee $O/r03l_prep_exp.txt

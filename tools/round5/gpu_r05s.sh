#!/bin/bash
# round 5, call s: k_prep_link with all its loads in flight at once, and what it is made of (timing only: L1 without the earlier workgroups' reports, L2 without table and decisions, L6 without those and the loads, L7 without all)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05s; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" L1 L2 L6 L7; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  [ -z "$v" ] && ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -3 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done 2>&1 | tee $O/prep_variants.txt
unset MDK_BUILD_DIR


#!/bin/bash
# round 5, call r: k_prep_link settles the reads no other read of their name can meet (mdk_isolated) and asks the table only for the rest;
# link0 = every name enters the table at the end of k_prep_scan (the arrangement before, and the one kept).  The builds are at no commit.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05r; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" link0; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -3 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done 2>&1 | tee $O/prep_variants.txt
unset MDK_BUILD_DIR
cd $R; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_tail.txt

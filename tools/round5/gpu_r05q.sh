#!/bin/bash
# round 5, call q: the name table asked by a kernel of its own (k_prep_link; link0 = asked at the end of k_prep_scan as before), with every
# read asking (local0) and without the LDS window (stage0); what k_prep_segs is made of now (x1 no tile atomics, x2 no write pass, x4 no
# wait, x7 none of the three: timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05q; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" link0 local0 stage0 x1 x2 x4 x7; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  case "$v" in x*) ;; *) ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -1 ) ;; esac
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done 2>&1 | tee $O/prep_variants.txt
unset MDK_BUILD_DIR
cd $R; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x 2>&1 | tail -2 | tee $O/pytest_tail.txt

#!/bin/bash
# round 5, call k: (1) the scan reading its records from the L2 after a coalesced warm-up pass instead of staging them in LDS (pf2);
# (2) the dense-context kernel with the reads' bytes going through LDS in 16-byte loads (qst): parity tests + kernel time, default vs qst
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
( cd /tmp; export TMPDIR=/tmp
for v in "" pf2; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  [ -z "$v" ] || ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py -m gpu -q -x 2>&1 | tail -1 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done ) 2>&1 | tee $O/prep_variants.txt
export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_qst
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_bed.py -m gpu -q -x > $O/pytest_qst.log 2>&1; echo "pytest qst rc=$?"; tail -2 $O/pytest_qst.log
unset MDK_BUILD_DIR
timeout 600 python tools/kbench.py --resident 16 --variants "default:;qst:MDK_BUILD_DIR=$R/methyldackel_amd/_exp_qst" --cmds 'dense:--CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146' 2>&1 | tee $O/kbench_dense.txt

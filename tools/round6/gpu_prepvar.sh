#!/bin/bash
# round 6: build variants of the preparation kernels (methyldackel_amd/_exp_<name>, built here by `make B=... HIPFLAGS=...`) against the default:
# test_gpu_prep.py, then the per-kernel times of the 8-chunk launches.  usage: gpu_prepvar.sh TAG variant...
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$1; shift; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" "$@"; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  [ -z "$SKIP_TESTS" ] && ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py -m gpu -q -x 2>&1 | tail -1 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done 2>&1 | tee $O/${T}_prep_variants.txt

#!/bin/bash
# round 5, call h: scan with the table's compare-and-swaps at workgroup scope (timing + does test_gpu_prep still pass); mbias with the deferred
# histogram launch and the device inflate (tests); the command end to end at 128 Mb and 512 Mb, teardown in place / detached / a queue of samples
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_mbias.py tests/test_gpu_prep.py tests/test_gpu_stress.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
( cd /tmp; export TMPDIR=/tmp
for v in "" wgcas; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  [ -z "$v" ] || ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py -m gpu -q -x 2>&1 | tail -1 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done ) 2>&1 | tee $O/prep_variants.txt
unset MDK_BUILD_DIR
timeout 900 python tools/round5/e2e.py $O 4 2>&1 | tee $O/e2e.log

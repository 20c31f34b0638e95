#!/bin/bash
# round 5, call j: k_prep_segs at 8 waves/SIMD; where the time outside the command's own clock goes (teardown in place) under switches
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== $(cat /tmp/pl.json)"; python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep ) 2>&1 | tee $O/prep_variants.txt
timeout 900 python tools/round5/exit_probe.py $O 2>&1 | tee $O/exit_probe.log

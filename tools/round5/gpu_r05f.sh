#!/bin/bash
# round 5, call f: the scan pairs a workgroup's reads in LDS before the chunk's table is asked (default) vs every read asks (nolocal);
# timing-only builds of k_prep_segs with parts left out (x1 no tile atomics, x2 no write pass, x4 no ticket wait, x8 ticket from blockIdx, x31 all)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
( cd /tmp; export TMPDIR=/tmp
for v in "" nolocal x1 x2 x4 x8 x31; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done ) 2>&1 | tee $O/prep_variants.txt

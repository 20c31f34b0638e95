#!/bin/bash
# round 5, call g: k_prep_segs stages its segments in LDS (made while the earlier tickets' counts arrive, written out as whole lines), tile runs
# merged per workgroup; new edge-case test (stage overflow, far skips, mates in other workgroups)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_mbias.py tests/test_gpu_perread.py tests/test_gpu_bed.py tests/test_zoo.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
( cd /tmp; export TMPDIR=/tmp
for v in ""; do
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
  PREP_BENCH_FAST= timeout 200 python $R/tools/prep_bench.py 16 2>/dev/null
done ) 2>&1 | tee $O/prep_variants.txt

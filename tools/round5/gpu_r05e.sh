#!/bin/bash
# round 5, call e: preparation with the blocks' last admitted starts (no slow path for a block's first read), the first compare-and-swap
# issued before the PrepReads leave, 96 SGPRs in k_prep_segs.  Variants: 128-thread workgroups; no atomics at all (timing only).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
df -h /tmp /dev/shm . 2>/dev/null > $O/df.txt; free -g >> $O/df.txt; nproc >> $O/df.txt
timeout 900 python -m pytest tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_mbias.py tests/test_gpu_perread.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
( cd /tmp; export TMPDIR=/tmp
for v in "" pb128 nocas; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  [ "$v" = nocas ] || ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py -m gpu -q -x 2>&1 | tail -1 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done ) 2>&1 | tee $O/prep_variants.txt

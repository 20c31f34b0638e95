#!/bin/bash
# round 3, GPU call: one 8-byte name-table entry per name (2 MB table) and the chunks of a launch dealt to the XCDs (PREP_XCD), against
# the same table with the contiguous mapping (methyldackel_amd/_exp_noxcd)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py tests/test_gpu_mbias.py tests/test_gpu_perread.py tests/test_gpu_multi.py -m gpu -x -q > $O/r03w_pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/r03w_pytest.log
cd /tmp; export TMPDIR=/tmp PREP_BENCH_FAST=1
for v in "" noxcd "" noxcd; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  rm -rf /tmp/pw_kt
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pw.json 2> /dev/null
  f=$(find /tmp/pw_kt -name '*kernel_stats.csv' | head -1)
  echo "== variant [${v:-xcd}] $(python -c "import json; d=json.load(open('/tmp/pw.json')); print('prep us/chunk', round(d['prep_ms_per_chunk_8_per_launch']*1000,1))")"
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
done 2>&1 | tee $O/r03w_prep_xcd.txt

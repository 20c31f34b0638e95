#!/bin/bash
# round 3, GPU call: name table with the members in the entry, pair fast path; occupancy / block size variants of the preparation kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py tests/test_gpu_mbias.py tests/test_gpu_perread.py -m gpu -x -q > $O/r03k_pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/r03k_pytest.log
for v in "" a b c d; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  timeout 200 python tools/prep_bench.py 16 > $O/r03k_prep_bench_${v:-default}.json 2> $O/r03k_prep_bench.err; echo "variant [${v:-default}] rc=$?"; python -c "
import json,sys; d=json.load(open('$O/r03k_prep_bench_${v:-default}.json')); print({k: round(v*1000,1) for k,v in d.items() if k.startswith('prep_ms')})"
done
unset MDK_BUILD_DIR
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_kt -o kt -- python $R/tools/prep_bench.py 16 > /dev/null 2> $O/r03k_prof.err
f=$(find /tmp/pk_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/r03k_prep_kernel_stats.csv && head -6 "$f" | cut -c1-220

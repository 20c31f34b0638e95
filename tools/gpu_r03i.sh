#!/bin/bash
# round 3, GPU call: preparation kernels with wide loads (v3) -- prep tests, timings, per-kernel trace; then the ranks-mode tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py -m gpu -x -q > $O/r03i_pytest_prep.log 2>&1; echo "prep tests rc=$?"; tail -6 $O/r03i_pytest_prep.log
timeout 300 python tools/prep_bench.py 16 > $O/r03i_prep_bench.json 2> $O/r03i_prep_bench.err; echo "prep_bench rc=$?"; cat $O/r03i_prep_bench.json; tail -3 $O/r03i_prep_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2> $O/r03i_prof.err; f=$(find /tmp/prof_i -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/r03i_prep_kernel_stats.csv && head -12 "$f" | cut -c1-200 )
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_scaled_configs.py -m gpu -x -q > $O/r03i_pytest_ranks.log 2>&1; echo "ranks tests rc=$?"; tail -15 $O/r03i_pytest_ranks.log

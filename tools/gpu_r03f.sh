#!/bin/bash
# round 3, GPU call: fused multi-chunk preparation kernels + group launches in the command: whole GPU suite, timings
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python tools/prep_bench.py 16 > $O/r03f_prep_bench.json 2> $O/r03f_prep_bench.err; echo "prep_bench rc=$?"; cat $O/r03f_prep_bench.json; tail -3 $O/r03f_prep_bench.err
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03f_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/r03f_pytest_gpu.log

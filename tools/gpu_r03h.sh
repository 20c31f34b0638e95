#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_inflate.py -m gpu -x -q > $O/r03h_pytest.log 2>&1; echo "tests rc=$?"; tail -15 $O/r03h_pytest.log
timeout 1500 python bench.py --data-dir /tmp/mdk_bench_data > $O/r03h_bench.json 2> $O/r03h_bench.err; echo "bench rc=$?"; cat $O/r03h_bench.json; tail -5 $O/r03h_bench.err

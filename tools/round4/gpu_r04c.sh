#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_inflate.py tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r04c_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04c_pytest.log
timeout 900 python tools/round4/r04b.py r04c > gpurun_out/r04c_stdout.txt 2> gpurun_out/r04c_stderr.txt; echo rc=$?
grep "^==" gpurun_out/r04c_e2e.txt; tail -3 gpurun_out/r04c_stderr.txt

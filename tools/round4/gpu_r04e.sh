#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_parity.py tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_multi.py tests/test_gpu_mbias.py tests/test_gpu_perread.py tests/test_gpu_bed.py -m gpu -x -q > gpurun_out/r04e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04e_pytest.log
R04_VARIANTS=default,norelease,norectab,cap24,cap8,hostinflate timeout 900 python tools/round4/r04b.py r04e 128000000,32000000 > gpurun_out/r04e_stdout.txt 2> gpurun_out/r04e_stderr.txt; echo rc=$?
grep "^==" gpurun_out/r04e_e2e.txt; tail -3 gpurun_out/r04e_stderr.txt

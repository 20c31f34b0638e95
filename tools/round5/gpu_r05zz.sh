#!/bin/bash
# round 5, last call: the tests that run files larger than a device piece, on the tree with 96 MB device pieces by default
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05zz; mkdir -p $O; cd $R
timeout 125 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_scaled_configs.py tests/test_gpu_bench.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log

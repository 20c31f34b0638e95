#!/bin/bash
# round 4, last GPU call: the command with the runtime narrowed to its own device, on the last tree of the round
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_inflate.py -m gpu -q -x > $O/r04end2_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04end2_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04end2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r04end2_smoke.log

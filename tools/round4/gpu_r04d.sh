#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/eph tools/exit_probe_hip.hip
( for rep in 1 2 3; do for m in init vram reg reg_unreg reg_unreg_drop reg_unreg_sleep_drop reg_unreg_unmap hostmalloc plain plain_drop; do /tmp/eph $m 3; sleep 0.3; done; done ) > gpurun_out/r04d_exit_probe.txt 2>&1
cat gpurun_out/r04d_exit_probe.txt
timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04d_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04d_pytest.log
R04_VARIANTS=default,cap12,cap20 timeout 900 python tools/round4/r04b.py r04d 128000000 > gpurun_out/r04d_stdout.txt 2> gpurun_out/r04d_stderr.txt; echo rc=$?
grep "^==" gpurun_out/r04d_e2e.txt; tail -3 gpurun_out/r04d_stderr.txt

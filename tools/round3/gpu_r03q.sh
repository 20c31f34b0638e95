#!/bin/bash
# round 3, GPU call: k_prep_segs without scratch (pairs resolved in closed form, fields of the other read taken apart in registers)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py tests/test_gpu_mbias.py -m gpu -x -q > $O/r03q_pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/r03q_pytest.log
timeout 300 python tools/prep_bench.py 16 > $O/r03q_prep_bench.json 2> $O/r03q_prep_bench.err; echo "prep_bench rc=$?"; cat $O/r03q_prep_bench.json

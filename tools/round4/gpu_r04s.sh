#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_parity.py -m gpu -x -q > $O/r04s_pytest.log 2>&1; echo pytest rc=$?; tail -2 $O/r04s_pytest.log
timeout 900 python tools/round4/r04s.py > $O/r04s_stdout.txt 2> $O/r04s_stderr.txt; echo rc=$?
grep "^==" $O/r04s_e2e.txt | cut -c1-330; tail -3 $O/r04s_stderr.txt; grep -h "first chunk\|first group\|warm-up\|md_dev_open" $O/r04s_e2e.txt | head -8

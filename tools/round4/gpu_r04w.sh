#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/r04w_pytest.log 2>&1; echo pytest rc=$?; tail -2 $O/r04w_pytest.log
timeout 600 python tools/round4/r04w.py > $O/r04w_stdout.txt 2> $O/r04w_stderr.txt; echo rc=$?
grep "^==" $O/r04w_e2e.txt | cut -c1-260; tail -2 $O/r04w_stderr.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 400 python tools/round4/r04x.py > $O/r04x_stdout.txt 2> $O/r04x_stderr.txt; echo rc=$?
grep "^==" $O/r04x_e2e.txt | cut -c1-260; tail -2 $O/r04x_stderr.txt
MDK_GROUPS_IN_FLIGHT=5 timeout 400 python -m pytest tests -m gpu -x -q > $O/r04x_pytest_g5.log 2>&1; echo pytest rc=$?; tail -2 $O/r04x_pytest_g5.log

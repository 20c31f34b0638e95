#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/r04p_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/r04p_pytest.log
timeout 900 python tools/round4/r04p.py > $O/r04p_stdout.txt 2> $O/r04p_stderr.txt; echo rc=$?
grep "^==" $O/r04p_e2e.txt | cut -c1-330; tail -3 $O/r04p_stderr.txt; grep -h "reference text" $O/r04p_e2e.txt | head -3

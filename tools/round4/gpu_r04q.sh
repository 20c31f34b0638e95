#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python tools/round4/r04q.py > $O/r04q_stdout.txt 2> $O/r04q_stderr.txt; echo rc=$?
grep "^==" $O/r04q_e2e.txt | cut -c1-330; tail -3 $O/r04q_stderr.txt; grep -h "first chunk\|first group" $O/r04q_e2e.txt | head -6

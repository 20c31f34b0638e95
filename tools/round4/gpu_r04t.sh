#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for m in B A C B A; do timeout 200 python tools/round4/r04t.py $m >> $O/r04t_e2e.txt 2>> $O/r04t_stderr.txt; done
grep "^==" $O/r04t_e2e.txt; tail -3 $O/r04t_stderr.txt

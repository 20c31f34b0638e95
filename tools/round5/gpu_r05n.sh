#!/bin/bash
# round 5, call n: how the XL sample's run depends on the split of the inflate between host threads and device
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
timeout 900 python tools/round5/e2e_sweep.py $O 2>&1 | tee $O/e2e_sweep.log

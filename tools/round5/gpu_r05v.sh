#!/bin/bash
# round 5, call v: device slabs beyond one per team (the teams wait for a free one a quarter of their time with 4: r05u) -- 4 / 12 / 24, with 8 and 12 teams
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
export E2E_CFGS='[["extra4","64",{}],["extra12","64",{"MDK_DSLAB_EXTRA":"12"}],["extra24","64",{"MDK_DSLAB_EXTRA":"24"}],["teams12_extra16","64",{"MDK_GPU_INFLATE_TEAMS":"12","MDK_DSLAB_EXTRA":"16"}],["extra12_groups5","64",{"MDK_DSLAB_EXTRA":"12","MDK_GROUPS_IN_FLIGHT":"5"}]]'
timeout 600 python tools/round5/e2e_sweep.py $O 2>&1 | tee $O/e2e_sweep.log

#!/bin/bash
# round 5, call u: where the inflate teams' time goes at 512 Mb (new profile line), with 4 / 8 shared piece streams and 8 / 16 device teams
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
export E2E_CFGS='[["default","64",{}],["streams8","64",{"MDK_PIECE_STREAMS":"8"}],["teams16_streams8","64",{"MDK_GPU_INFLATE_TEAMS":"16","MDK_PIECE_STREAMS":"8"}],["host_only","64",{"MDK_GPU_INFLATE_TEAMS":"0"}]]'
timeout 600 python tools/round5/e2e_sweep.py $O 2>&1 | tee $O/e2e_sweep.log

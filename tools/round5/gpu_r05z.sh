#!/bin/bash
# round 5, call z: device pieces of 64 (default) / 96 / 112 / 128 MB at 512 Mb: a 64 MB piece is ~3,400 members, one wavefront each, on a
# device that holds ~6,100 of them, and its k_inflate runs mostly alone (r05x)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
export E2E_CFGS='[["piece64","64",{}],["piece96","64",{"MDK_GPU_PIECE_MB":"96"}],["piece112","64",{"MDK_GPU_PIECE_MB":"112"}],["piece128_teams6","64",{"MDK_GPU_PIECE_MB":"128","MDK_GPU_INFLATE_TEAMS":"6"}]]'
timeout 400 python tools/round5/e2e_sweep.py $O 2>&1 | tee $O/e2e_sweep.log

#!/bin/bash
# round 5, call y: the 512 Mb run with more hardware queues for the runtime's streams (GPU_MAX_HW_QUEUES; the trace of r05x shows 1.4 k_inflate
# at a time and no kernel at all for a third of the span) and more shared piece streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R
export E2E_CFGS='[["default","64",{}],["hwq8","64",{"GPU_MAX_HW_QUEUES":"8"}],["hwq16_streams8","64",{"GPU_MAX_HW_QUEUES":"16","MDK_PIECE_STREAMS":"8"}],["hwq8_streams8_extra12","64",{"GPU_MAX_HW_QUEUES":"8","MDK_PIECE_STREAMS":"8","MDK_DSLAB_EXTRA":"12"}]]'
timeout 400 python tools/round5/e2e_sweep.py $O 2>&1 | tee $O/e2e_sweep.log

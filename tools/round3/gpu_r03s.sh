#!/bin/bash
# round 3, GPU call: device inflate teams / piece size sweep on the 128 Mb sample (the command's own clock, MDK_HOST_PROFILE)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/mdk_big; mkdir -p $D; tools/_build/mdk_synth -o $D/s -L 128000000 -c 30 -s 1 > /dev/null 2>&1
M=$R/methyldackel_amd/_build/MethylDackel
for cfg in "" "MDK_GPU_INFLATE_TEAMS=4" "MDK_GPU_PIECE_MB=32" "MDK_GPU_PIECE_MB=32 MDK_GPU_INFLATE_TEAMS=4" "MDK_GPU_PIECE_MB=16 MDK_GPU_INFLATE_TEAMS=4" "MDK_INFLATE_TEAMS=2" "MDK_HOST_INFLATE=1"; do
  for rep in 1 2 3; do
    mkdir -p $D/o; ( cd $D/o; sleep 0.5; env $cfg MDK_HOST_PROFILE=1 timeout 90 $M extract $D/s.fa $D/s.bam -@ 64 -o out 2> err.txt; echo "[$cfg] rep $rep rc $? $(grep -o 'total [0-9.]*s' err.txt | head -1) $(grep -o 'pieces inflated by the host teams [0-9]*, on the device [0-9]*' err.txt) $(grep -o 'waiting for a free slot [0-9.]*s' err.txt) $(grep -o 'wait-for-chunk [0-9.]*s submit [0-9.]*s download [0-9.]*s' err.txt)" )
  done
done 2>&1 | tee $O/r03s_inflate_sweep.txt

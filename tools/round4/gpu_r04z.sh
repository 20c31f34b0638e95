#!/bin/bash
# round 4, GPU call z (the last half minute): the experimental 128-bit decode round (-DINF_WIDE=1, built into methyldackel_amd/_build_wide) on a device
# for the first time: piece_bench with verification against zlib, whole file as one launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/mdk_r04; mkdir -p $D; tools/_build/mdk_synth -o $D/s32 -L 32000000 -c 30 -s 11 > /dev/null 2>&1
LD_LIBRARY_PATH=$R/methyldackel_amd/_build_wide timeout 60 tools/_build/piece_bench $D/s32.bam 1024 1 1 > $O/r04z_piece_bench_whole_wide.json 2> $O/r04z_piece_bench.err; echo rc=$?; cut -c1-1100 $O/r04z_piece_bench_whole_wide.json; tail -2 $O/r04z_piece_bench.err

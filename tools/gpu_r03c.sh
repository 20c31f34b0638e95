#!/bin/bash
# round 3, third GPU call: inflate v3 (64-bit bit buffer, 16-bit length entries, single-exit loop, real scalar/vector mix)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/r03c; mkdir -p $D
$R/tools/_build/mdk_synth -o $D/s32 -L 32000000 -c 30 -s 1589478401 > $D/s32.json
PIECE_BENCH_VARIANTS="0 1 3 4 6" timeout 600 $R/tools/_build/piece_bench $D/s32.bam 600 1 1 > $O/r03c_w4k_all.json 2> $O/r03c_w4k_all.err; echo "w4k rc=$?"; cat $O/r03c_w4k_all.json; tail -3 $O/r03c_w4k_all.err
PIECE_BENCH_VARIANTS="4" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 128 2 0 > $O/r03c_w4k_128.json 2>&1; cat $O/r03c_w4k_128.json
export LD_LIBRARY_PATH=$R/methyldackel_amd/_exp_w2k
PIECE_BENCH_VARIANTS="0 1 3 4 6" timeout 600 $R/tools/_build/piece_bench $D/s32.bam 600 1 1 > $O/r03c_w2k_all.json 2> $O/r03c_w2k_all.err; echo "w2k rc=$?"; cat $O/r03c_w2k_all.json; tail -3 $O/r03c_w2k_all.err
PIECE_BENCH_VARIANTS="4" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 128 2 0 > $O/r03c_w2k_128.json 2>&1; cat $O/r03c_w2k_128.json

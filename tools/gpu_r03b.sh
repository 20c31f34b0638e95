#!/bin/bash
# round 3, second GPU call: inflate v2 (16-bit tables, 16-byte far loads, readlane tokens, dword flush), two window sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/r03b; mkdir -p $D
$R/tools/_build/mdk_synth -o $D/s32 -L 32000000 -c 30 -s 1589478401 > $D/s32.json
PIECE_BENCH_VARIANTS="0 1 3 4" timeout 600 $R/tools/_build/piece_bench $D/s32.bam 128 2 1 > $O/r03b_w4k_128.json 2> $O/r03b_w4k_128.err; echo "w4k rc=$?"; cat $O/r03b_w4k_128.json; tail -3 $O/r03b_w4k_128.err
PIECE_BENCH_VARIANTS="1 3" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 600 1 0 > $O/r03b_w4k_all.json 2>&1; cat $O/r03b_w4k_all.json
export LD_LIBRARY_PATH=$R/methyldackel_amd/_exp_w2k
PIECE_BENCH_VARIANTS="0 1 3 4" timeout 600 $R/tools/_build/piece_bench $D/s32.bam 128 2 1 > $O/r03b_w2k_128.json 2> $O/r03b_w2k_128.err; echo "w2k rc=$?"; cat $O/r03b_w2k_128.json; tail -3 $O/r03b_w2k_128.err
PIECE_BENCH_VARIANTS="1 3" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 600 1 0 > $O/r03b_w2k_all.json 2>&1; cat $O/r03b_w2k_all.json
PIECE_BENCH_VARIANTS="1" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 32 4 0 > $O/r03b_w2k_32.json 2>&1; cat $O/r03b_w2k_32.json

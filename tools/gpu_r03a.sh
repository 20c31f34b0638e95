#!/bin/bash
# round 3, first GPU call: the device inflate on its own (correctness vs zlib, kernel variants, piece sizes) + the feed probe
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/r03a; mkdir -p $D
rocminfo | grep -m3 "Marketing Name" > $O/r03a_box.txt; nproc >> $O/r03a_box.txt
( time $R/tools/_build/mdk_synth -o $D/s32 -L 32000000 -c 30 -s 1589478401 ) > $D/s32.json 2> $O/r03a_synth_time.txt
ls -la $D >> $O/r03a_box.txt
timeout 600 $R/tools/_build/piece_bench $D/s32.bam 64 3 1 > $O/r03a_piece64.json 2> $O/r03a_piece64.err; echo "piece64 rc=$?"; cat $O/r03a_piece64.json; tail -3 $O/r03a_piece64.err
PIECE_BENCH_VARIANTS="1" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 16 4 0 > $O/r03a_piece16.json 2>&1; echo "piece16 rc=$?"; cat $O/r03a_piece16.json
PIECE_BENCH_VARIANTS="1 4" timeout 300 $R/tools/_build/piece_bench $D/s32.bam 256 2 0 > $O/r03a_piece256.json 2>&1; echo "piece256 rc=$?"; cat $O/r03a_piece256.json
timeout 300 $R/tools/_build/pin_probe 512 $D/s32.bam > $O/r03a_pin_probe.json 2>&1; echo "pin rc=$?"; cat $O/r03a_pin_probe.json
cd /tmp; export TMPDIR=/tmp
PIECE_BENCH_VARIANTS="1" rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03a_prof_kt -o kt -- $R/tools/_build/piece_bench $D/s32.bam 64 3 0 > /dev/null 2>&1
find $O/r03a_prof_kt -name "*kernel_stats.csv" | head -1 | xargs cat

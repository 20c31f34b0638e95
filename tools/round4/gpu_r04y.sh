#!/bin/bash
# round 4, GPU call y: k_inflate with two token places per lane (128 tokens per batch): parity on the device, HIP-event and rocprofv3 timings
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q > $O/r04y_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/r04y_pytest.log
D=/tmp/mdk_r04; mkdir -p $D; tools/_build/mdk_synth -o $D/s32 -L 32000000 -c 30 -s 11 > /dev/null 2>&1
timeout 100 tools/_build/piece_bench $D/s32.bam 1024 3 0 > $O/r04y_piece_bench_whole.json 2> $O/r04y_piece_bench.err; cat $O/r04y_piece_bench_whole.json | cut -c1-900
timeout 100 tools/_build/piece_bench $D/s32.bam 64 3 1 > $O/r04y_piece_bench_64.json 2>> $O/r04y_piece_bench.err; cut -c1-900 $O/r04y_piece_bench_64.json
cd /tmp; export TMPDIR=/tmp; rm -rf $O/r04y_inflate_kt
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04y_inflate_kt -o kt -- $R/tools/_build/piece_bench $D/s32.bam 1024 1 0 > /dev/null 2>&1
head -4 $(find $O/r04y_inflate_kt -name kt_kernel_stats.csv | head -1)

#!/bin/bash
# round 4, GPU call n: k_inflate with the prefix sum in DPP, the leaner walk and the branch-free bit window
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q > $O/r04n_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04n_pytest.log
D=/tmp/mdk_r04; mkdir -p $D
tools/_build/mdk_synth -o $D/s32000000 -L 32000000 -c 30 -s 11 > /dev/null 2>&1
timeout 300 tools/_build/piece_bench $D/s32000000.bam 64 3 1 > $O/r04n_piece_bench.json 2> $O/r04n_piece_bench.err; echo "piece_bench rc=$?"; cat $O/r04n_piece_bench.json; tail -2 $O/r04n_piece_bench.err
timeout 300 tools/_build/piece_bench $D/s32000000.bam 1024 3 0 > $O/r04n_piece_bench_whole.json 2>> $O/r04n_piece_bench.err; cat $O/r04n_piece_bench_whole.json
timeout 600 python tools/round4/r04n.py > $O/r04n_stdout.txt 2> $O/r04n_stderr.txt; echo rc=$?
grep "^==" $O/r04n_e2e.txt | cut -c1-300; tail -3 $O/r04n_stderr.txt

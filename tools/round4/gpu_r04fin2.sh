#!/bin/bash
# round 4: the bench alone on the final tree (after the isolation of the end-to-end runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python bench.py --data-dir /tmp/mdk_bench_data > $O/r04fin2_bench.json 2> $O/r04fin2_bench.err; echo "bench rc=$?"; tail -12 $O/r04fin2_bench.err

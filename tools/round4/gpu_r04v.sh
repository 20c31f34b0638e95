#!/bin/bash
# round 4, GPU call v: the bench on the final tree (end-to-end runs one second apart)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python bench.py --data-dir /tmp/mdk_bench_data > $O/r04fin4_bench.json 2> $O/r04fin4_bench.err; echo "bench rc=$?"; grep "default:\|large_default\|large_inplace\|xl_default\|xl_inplace\|oracle allcore\|oracle large_all\|oracle xl\|done in" $O/r04fin4_bench.err

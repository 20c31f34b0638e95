#!/bin/bash
# round 4, GPU call u: as t with the previous command's child recognised by name (a process taking its address space down has no command line),
# then the bench on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for m in B A; do timeout 200 python tools/round4/r04t.py $m >> $O/r04u_e2e.txt 2>> $O/r04u_stderr.txt; done
grep "^==" $O/r04u_e2e.txt
timeout 1200 python bench.py --data-dir /tmp/mdk_bench_data > $O/r04fin3_bench.json 2> $O/r04fin3_bench.err; echo "bench rc=$?"; grep "default:\|large_default\|large_inplace\|xl_default\|xl_inplace\|oracle allcore\|oracle large_all\|oracle xl" $O/r04fin3_bench.err

#!/bin/bash
# round 6: the kernels of the configs[2] command (128 Mb, --CHG --CHH with trimming): rocprofv3 kernel trace, per-kernel sums
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; D=/tmp/c3; mkdir -p $D; cd $D; export TMPDIR=/tmp
[ -f s.bam ] || $R/tools/_build/mdk_synth -o s -L 128000000 -c 30 -s 1234 -j 16 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel; X="--CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146"
$M extract s.fa s.bam -@ 64 $X -o warm > /dev/null 2>&1
rm -rf /tmp/c3_kt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c3_kt -o kt -- $M extract s.fa s.bam -@ 64 $X -o out > /dev/null 2> err.txt
python3 $R/tools/round6/trace_summary.py $(find /tmp/c3_kt -name '*kernel_trace.csv' | head -1) 2>&1 | head -24 | tee $O/r06c6_cfg3_trace_summary.txt

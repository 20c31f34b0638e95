#!/bin/bash
# mbias on the GPU box: end-to-end time of `MethylDackel mbias` next to the CPU oracle, and the kernel time of k_mbias
# (rocprofv3 --kernel-trace --stats) for CpG-only and all-context runs on a synthetic sample.  usage: gpu_mbias.sh <tag> [length]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-run}; LEN=${2:-1000000}
W=/tmp/mb_work; mkdir -p $W; cd $W; export TMPDIR=/tmp
$R/tools/_build/mdk_synth -o S -L $LEN -c 30 -s 0x5EED0001 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel; OR=$R/oracle/_build/mdk_oracle
TIMEFORMAT="%R s"
{
  echo "sample: $LEN bp, 30x PE 2x150, $(nproc) host cores"
  for ctx in "" "--CHG --CHH"; do
    for i in 1 2; do echo -n "gpu mbias [$ctx] wall "; time timeout 120 $M mbias S.fa S.bam --noSVG -@ 16 $ctx > g.txt; done
    echo -n "oracle mbias [$ctx] wall "; time timeout 300 $OR mbias S.fa S.bam --noSVG $ctx > o.txt
    [ -s g.txt ] && cmp g.txt o.txt && echo "tables identical [$ctx] ($(wc -l < g.txt) lines)"
  done
} > $O/${TAG}_mbias_e2e.txt 2>&1
cat $O/${TAG}_mbias_e2e.txt
# under the profiler the command runs through the library entry point (normal teardown instead of the command's fast exit)
cat > drv.py <<PY
import sys; sys.path.insert(0, "$R")
import ctypes as C, methyldackel_amd as mdk
a = ["mbias", "S.fa", "S.bam", "--noSVG", "-@", "16"] + sys.argv[1:]
sys.exit(mdk.lib_extract().mbias_main(len(a), mdk._argv(a)) & 255)
PY
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_mbias_cpg -o kt -- python drv.py > /dev/null 2>&1; echo "rocprof cpg rc=$?"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_mbias_all -o kt -- python drv.py --CHG --CHH > /dev/null 2>&1; echo "rocprof all rc=$?"
for d in cpg all; do f=$(find $O/${TAG}_prof_mbias_$d -name "*kernel_stats.csv" 2>/dev/null | head -1); echo "== $d: $f"; [ -n "$f" ] && head -5 $f; done

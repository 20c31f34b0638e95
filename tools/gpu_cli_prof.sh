#!/bin/bash
# rocprofv3 kernel trace of the `MethylDackel extract` command itself (device preparation + pileup), per-kernel statistics.
# usage: tools/gpu_cli_prof.sh TAG [length] [extra extract options]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-cli}; L=${2:-16000000}; shift; shift
mkdir -p $O /tmp/cliprof; cd /tmp/cliprof
[ -f s.bam ] || $R/tools/_build/mdk_synth -o s -L $L -c 30 -s 77 > /dev/null
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cli_kt -o kt -- $R/methyldackel_amd/_build/MethylDackel extract s.fa s.bam -@ 32 -o out "$@" > /dev/null 2>&1
cat $(find $O/${TAG}_cli_kt -name kt_kernel_stats.csv)

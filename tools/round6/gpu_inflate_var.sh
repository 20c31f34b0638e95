#!/bin/bash
# round 6: k_inflate of build variants: the whole file as one launch and 96 MB pieces (kernel times by HIP events), every byte verified on the 64 MB run.  usage: gpu_inflate_var.sh TAG variant...   ("-": the default build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift; L=32000000; D=/tmp/inf; mkdir -p $D; cd $D
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
for v in "$@"; do
  lib=$R/methyldackel_amd/_build; [ "$v" != "-" ] && lib=$R/methyldackel_amd/_exp_$v
  for mode in "4000 1 0" "96 8 0" "64 3 1"; do
    LD_LIBRARY_PATH=$lib timeout 300 $R/tools/_build/piece_bench s$L.bam $mode 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_only']['v0']; p=d.get('pass1', d.get('pass2'))
print('[$v] pieces of $mode: k_inflate %.3f ms, %.2f GB/s compressed; pipelined %.2f GB/s; verified %s' % (k['inflate_ms'], k['GBps_compressed'], p['GBps_compressed'], d.get('verified')))"
  done
done 2>&1 | tee $O/${TAG}_inflate_variants.txt

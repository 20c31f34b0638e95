#!/bin/bash
# round 6: where k_inflate's wavefronts spend their time (the -DINF_PROFILE build: s_memtime per phase).  usage: tools/round6/gpu_inflate_prof.sh TAG [LIBDIR...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06prof}; shift; L=32000000; D=/tmp/inf; mkdir -p $D; cd $D
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
for lib in "$@"; do
  echo "--- $lib"; LD_LIBRARY_PATH=$R/methyldackel_amd/$lib timeout 300 $R/tools/_build/piece_bench s$L.bam 4000 1 ${VERIFY:-0} 2> $O/${TAG}_$lib.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_only'])"; grep "inf profile" $O/${TAG}_$lib.err | tail -1
done

#!/bin/bash
# round 6: the device inflate alone on the GPU box -- piece_bench (every byte against zlib, record offsets, digests; kernel times), the inflate tests.
# usage: tools/round6/gpu_inflate.sh TAG [LENGTH]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06inf}; L=${2:-32000000}; D=/tmp/inf; mkdir -p $D; cd $D
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
[ -f z6.bam ] || $R/tools/_build/mdk_synth -o z6 -L 4000000 -c 30 -s 7 -z 6 > /dev/null
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
echo "--- 64 MB pieces, 3 in flight, verified"; timeout 300 $R/tools/_build/piece_bench s$L.bam 64 3 1 > $O/${TAG}_piece64.json 2> $O/${TAG}_piece64.err; echo "rc=$?"; cat $O/${TAG}_piece64.json; tail -3 $O/${TAG}_piece64.err
echo "--- level-6 file, verified"; timeout 300 $R/tools/_build/piece_bench z6.bam 64 3 1 > $O/${TAG}_piece_z6.json 2> $O/${TAG}_piece_z6.err; echo "rc=$?"; cat $O/${TAG}_piece_z6.json; tail -3 $O/${TAG}_piece_z6.err
echo "--- whole file as one piece"; timeout 300 $R/tools/_build/piece_bench s$L.bam 4000 1 0 > $O/${TAG}_piece_whole.json 2> $O/${TAG}_piece_whole.err; echo "rc=$?"; cat $O/${TAG}_piece_whole.json; tail -3 $O/${TAG}_piece_whole.err
echo "--- 96 MB pieces, 8 in flight"; timeout 300 $R/tools/_build/piece_bench s$L.bam 96 8 0 > $O/${TAG}_piece96.json 2> $O/${TAG}_piece96.err; echo "rc=$?"; cat $O/${TAG}_piece96.json
cd $R
if [ "${3:-tests}" = tests ]; then timeout 900 python -m pytest tests/test_gpu_inflate.py -m gpu -q -x > $O/${TAG}_pytest_inflate.log 2>&1; echo "pytest rc=$?"; tail -5 $O/${TAG}_pytest_inflate.log; fi

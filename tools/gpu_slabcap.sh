#!/bin/bash
# effect of the reader's look-ahead (slabs) and of pinned slabs on the command's wall clock: tools/gpu_slabcap.sh LENGTH
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=${1:-128000000}; D=/tmp/e2e; mkdir -p $D; cd $D
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel
for pin in 16000000000 1; do for cap in 48 16 10 6; do for rep in 1 2; do sleep 0.6; t0=$(date +%s.%N); MDK_PIN_MIN_BYTES=$pin MDK_SLAB_CAP=$cap MDK_HOST_PROFILE=1 $M extract s$L.fa s$L.bam -@ 48 -o out 2>&1 | grep -E "mdk main\] plan" | sed 's/chunks prepared.*//'; t1=$(date +%s.%N); python3 -c "print('pin_min $pin cap $cap wall %.3f s' % ($t1 - $t0))"; done; done; done

#!/bin/bash
# start-up cost experiments on the GPU box: wall-clock of `MethylDackel extract` over a 32 Mb / 64 Mb sample under env variations
R=$GRAFT_REPO_ROOT; W=/tmp/if_work; mkdir -p $W; cd $W
TIMEFORMAT="%R"
for L in 32000000 64000000; do
  $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
  for mode in "env MDK_PIN_MIN_BYTES=0" "env -u MDK_PIN_MIN_BYTES" "env MDK_PIN_MIN_BYTES=0" "env -u MDK_PIN_MIN_BYTES"; do
    t=$( { time MDK_HOST_PROFILE=1 $mode $R/methyldackel_amd/_build/MethylDackel extract s$L.fa s$L.bam -o o -@ 64 2> err.txt; } 2>&1 )
    echo "$L [$mode]: $t s | $(grep 'mdk main' err.txt | cut -c1-130)"
  done
done

#!/bin/bash
R=$GRAFT_REPO_ROOT; W=/tmp/if_work; mkdir -p $W; cd $W
$R/tools/_build/mdk_synth -o s -L 32000000 -c 30 -s 99 > /dev/null
TIMEFORMAT="%R"
for mode in "env -u MDK_NO_PIN" "env MDK_NO_PIN=1" "env -u MDK_NO_PIN" "env MDK_NO_PIN=1"; do
  t=$( { time MDK_HOST_PROFILE=1 $mode $R/methyldackel_amd/_build/MethylDackel extract s.fa s.bam -o o -@ 64 2> err.txt; } 2>&1 )
  echo "INIT_FIRST='$mode': $t s | $(grep 'mdk main' err.txt | cut -c1-120)"
done

#!/bin/bash
# round 6: the pipeline's queues over time (MDK_WATCHDOG=<ms>: a line per interval) for the K x 128 Mb run.  usage: tools/round6/gpu_wd.sh TAG [COPIES] [MS] [ENV...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; K=${2:-4}; MS=${3:-5}; shift 3
D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl $K > /dev/null 2>&1
M=$R/methyldackel_amd/_build/MethylDackel
$M extract xl.fa xl.bam -@ 64 -o warm > /dev/null 2>&1
sleep 1
env "$@" MDK_WATCHDOG=$MS MDK_HOST_PROFILE=1 $M extract xl.fa xl.bam -@ 64 -o out 2> $O/${TAG}_wd.err
grep -c "^\[wd\]" $O/${TAG}_wd.err; grep -E "plan open|reader:" $O/${TAG}_wd.err | cut -c1-300

#!/bin/bash
# round 6: what the command's exit costs against the size of its input: wall clock of the caller, the clock inside the process, what it held when it left.
# usage: gpu_exit.sh TAG "ENV.." sizes_in_Mb...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; ENVS=$2; shift 2
D=/tmp/mdk_exit_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
M=$R/methyldackel_amd/_build/MethylDackel
for L in "$@"; do
  $R/tools/_build/mdk_synth -o s$L -L ${L}000000 -c 30 -s 1234 -j 16 > /dev/null
  $M extract s$L.fa s$L.bam -@ 64 -o warm > /dev/null 2>&1
  for rep in 1 2 3; do
    sleep 1; t0=$(date +%s.%N); env $ENVS MDK_HOST_PROFILE=1 $M extract s$L.fa s$L.bam -@ 64 -o out 2> err.txt; t1=$(date +%s.%N)
    ins=$(grep -o "total [0-9.]*s" err.txt | head -1 | tr -dc '0-9.'); lv=$(grep -o "leaving at epoch [0-9.]*" err.txt | tr -dc '0-9.'); res=$(grep -o "(resident [^)]*)" err.txt | tail -1)
    python3 -c "print('[%s] %d Mb: wall %.3f inside %s exit %.3f %s' % ('$ENVS', $L, $t1-$t0, '$ins', $t1-float('$lv'), '$res'))"
  done
  rm -f s$L.bam s$L.fa s$L.bam.bai
done 2>&1 | tee -a $O/${TAG}_exit.txt

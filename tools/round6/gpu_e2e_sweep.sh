#!/bin/bash
# round 6: the 512 Mb (K copies of 128 Mb) run under settings of the feed's knobs; wall clock + the command's own clock, RUNS runs each.
# usage: tools/round6/gpu_e2e_sweep.sh TAG COPIES RUNS "ENV1=.. ENV2=.." "..." ...   (each further argument is one setting; "-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; K=$2; RUNS=$3; shift 3
D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
if [ "$K" -gt 1 ]; then $R/tools/_build/mdk_replicate s128 xl $K > /dev/null 2>&1; F=xl; else F=s128; fi
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 -o warm > /dev/null 2>&1
for setting in "$@"; do
  [ "$setting" = "-" ] && setting=""
  line=""
  for rep in $(seq $RUNS); do
    sleep 1; t0=$(date +%s.%N); env $setting MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ ${THREADS:-64} -o out 2> err.txt; rc=$?; t1=$(date +%s.%N)
    inner=$(grep -o "total [0-9.]*s" err.txt | head -1 | tr -dc '0-9.')
    line="$line $(python3 -c "print('%.3f/%s' % ($t1-$t0, '$inner'))")"
    [ $rc != 0 ] && line="$line rc=$rc"
  done
  echo "[$setting] wall/inside:$line" | tee -a $O/${TAG}_sweep.txt
  grep -E "reader:" err.txt | sed 's/.*reader:/   reader:/' | cut -c1-200
  grep -E "framing the pieces|device teams, summed" err.txt | sed 's/^\[mdk host\]/  /' | cut -c1-330
  grep -E "plan open" err.txt | sed 's/.*device ready/   device ready/' | cut -c1-260
done

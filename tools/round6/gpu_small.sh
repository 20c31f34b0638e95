#!/bin/bash
# round 6: the fixed costs -- the 32 Mb and 128 Mb runs under settings (wall / inside), 5 runs each
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06small}
D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -j 16 -o s32 -L 32000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_synth -j 16 -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel
$M extract s32.fa s32.bam -@ 64 -o warm > /dev/null 2>&1
for F in s32 s128; do
for setting in "-" "MDK_NO_RESERVE_HINT=1" "MDK_NO_REAP=1" "THREADS=128" "MDK_PIECE_LANES=0" "MDK_NO_PACK=1"; do
  T=64; [ "$setting" = "THREADS=128" ] && T=128
  [ "$setting" = "-" ] && setting=""
  line=""
  for rep in 1 2 3 4 5; do
    sleep 1; t0=$(date +%s.%N); env $setting MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ $T -o out 2> err.txt; rc=$?; t1=$(date +%s.%N)
    inner=$(grep -o "total [0-9.]*s" err.txt | head -1 | tr -dc '0-9.')
    line="$line $(python3 -c "print('%.3f/%s' % ($t1-$t0, '$inner'))")"
    [ $rc != 0 ] && line="$line rc=$rc"
  done
  echo "$F [$setting] wall/inside:$line" | tee -a $O/${TAG}_sweep.txt
  grep -E "plan open" err.txt | cut -c1-300 | sed 's/^/    /' | tee -a $O/${TAG}_sweep.txt
done; done

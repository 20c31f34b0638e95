#!/bin/bash
# round 6: with k_sites_pack in: where a group's time goes (MDK_HOST_PROFILE's group lines), the queues over time, hardware queues, a kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06pb}; K=${2:-4}; RUNS=${3:-2}
D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -j 16 -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl $K > /dev/null 2>&1; F=xl
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 -o warm > /dev/null 2>&1
for setting in ${SETTINGS:-"-" "MDK_PIECE_PRIO=0" "MDK_NO_PACK=1" "-" "MDK_PIECE_PRIO=0" "MDK_GROUPS_IN_FLIGHT=5"}; do
  [ "$setting" = "-" ] && setting=""
  line=""
  for rep in $(seq $RUNS); do
    sleep 1; t0=$(date +%s.%N); env $setting MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> err.txt; rc=$?; t1=$(date +%s.%N)
    inner=$(grep -o "total [0-9.]*s" err.txt | head -1 | tr -dc '0-9.')
    line="$line $(python3 -c "print('%.3f/%s' % ($t1-$t0, '$inner'))")"
    [ $rc != 0 ] && line="$line rc=$rc"
  done
  echo "[$setting] wall/inside:$line" | tee -a $O/${TAG}_sweep.txt
  grep -E "host threads inside|plan open|reader:|teams, summed" err.txt | cut -c1-700 | tee -a $O/${TAG}_sweep.txt
done
sleep 1
MDK_WATCHDOG=5 MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> $O/${TAG}_wd.err
export TMPDIR=/tmp; rm -rf /tmp/busy_kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/busy_kt -o kt -- $M extract $F.fa $F.bam -@ 64 -o x > /dev/null 2> $O/${TAG}_cmd.err
mkdir -p $O/${TAG}_trace; find /tmp/busy_kt -name '*kernel_trace.csv' -exec cp {} $O/${TAG}_trace/kernel_trace.csv \;

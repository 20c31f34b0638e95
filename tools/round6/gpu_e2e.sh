#!/bin/bash
# round 6: the command end to end on the GPU box: 128 Mb sample and K copies of it (in RAM-backed storage), with the host profile lines.
# usage: tools/round6/gpu_e2e.sh TAG [COPIES=4] [RUNS=3] [ORACLE=1]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06e2e}; K=${2:-4}; RUNS=${3:-3}; ORA=${4:-1}
D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
t0=$(date +%s.%N); $R/tools/_build/mdk_synth -o s128 -L 128000000 -c 30 -s 1234 > /dev/null; t1=$(date +%s.%N)
$R/tools/_build/mdk_replicate s128 xl $K > /dev/null 2>&1; t2=$(date +%s.%N)
python3 -c "print('synth %.1f s, replicate x$K %.1f s' % ($t1-$t0, $t2-$t1))"; ls -la $D | head
M=$R/methyldackel_amd/_build/MethylDackel
for f in s128 xl; do
  for rep in $(seq $RUNS); do
    sleep 1; t0=$(date +%s.%N); MDK_HOST_PROFILE=1 $M extract $f.fa $f.bam -@ ${THREADS:-64} -o out_$f 2> $O/${TAG}_${f}_$rep.err; rc=$?; t1=$(date +%s.%N)
    python3 -c "print('$f run $rep rc=$rc wall %.3f s' % ($t1-$t0))"; grep -E "mdk main|reader:|teams, summed" $O/${TAG}_${f}_$rep.err | cut -c1-700
  done
done
if [ "$ORA" = 1 ]; then
  t0=$(date +%s.%N); MDK_ORACLE_PROFILE=1 $R/oracle/_build/mdk_oracle extract xl.fa xl.bam -@ 32 --chunkSize 250000 -o ora 2> $O/${TAG}_oracle.err; t1=$(date +%s.%N)
  python3 -c "print('oracle xl -@32 wall %.3f s' % ($t1-$t0))"; grep oracle $O/${TAG}_oracle.err | head -20
  cmp ora_CpG.bedGraph out_xl_CpG.bedGraph && echo IDENTICAL
fi

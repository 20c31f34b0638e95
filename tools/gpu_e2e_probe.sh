#!/bin/bash
# e2e probe: where does the command's wall clock go (GPU box).  usage: tools/gpu_e2e_probe.sh LENGTH
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=${1:-64000000}; D=/tmp/e2e; mkdir -p $D; cd $D
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel
for th in ${THREADS:-16 32 64}; do
  for rep in 1 2 3; do
    t0=$(date +%s.%N); MDK_HOST_PROFILE=1 $M extract s$L.fa s$L.bam -@ $th -o out 2>&1 | grep -E "mdk main|mdk host" | sed -e "s/^\[mdk main\] \(entered\|leaving\)/[t0 $t0] \1/" | sed 's/; records found.*reader:/; reader:/'; t1=$(date +%s.%N)
    python3 -c "print('threads $th wall %.3f s (t1 epoch %.3f)' % ($t1 - $t0, $t1))"
  done
done
echo "--- pinned staging forced"
for th in 32 64; do t0=$(date +%s.%N); MDK_PIN_MIN_BYTES=1 MDK_HOST_PROFILE=1 $M extract s$L.fa s$L.bam -@ $th -o out 2>&1 | grep -E "mdk main" ; t1=$(date +%s.%N); python3 -c "print('pinned threads $th wall %.3f s' % ($t1 - $t0))"; done
echo "--- oracle all-core phases"
MDK_ORACLE_PROFILE=1 $R/oracle/_build/mdk_oracle extract s$L.fa s$L.bam -@ $(nproc) --chunkSize 62500 -o oo 2>&1 | grep oracle
echo "--- oracle 64 threads"
MDK_ORACLE_PROFILE=1 $R/oracle/_build/mdk_oracle extract s$L.fa s$L.bam -@ 64 --chunkSize 250000 -o oo 2>&1 | grep oracle

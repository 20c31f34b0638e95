#!/bin/bash
# round 5, the last seconds: a 64 Mb sample (1.1 GB BAM, a dozen 96 MB device pieces) through the command and through the oracle, outputs compared
R=${GRAFT_REPO_ROOT:-$(pwd)}; D=/dev/shm/mdk_end; mkdir -p $D; cd $D
$R/tools/_build/mdk_synth -o s -L 64000000 -c 30 -s 4242 > /dev/null 2>&1
$R/oracle/_build/mdk_oracle extract -@ 64 --chunkSize 250000 s.fa s.bam -o ref > /dev/null 2>&1 &
MDK_HOST_PROFILE=1 $R/methyldackel_amd/_build/MethylDackel extract -@ 64 s.fa s.bam -o ours 2>&1 | grep -h "pieces inflated" | sed 's/.*pieces inflated/pieces inflated/'
wait
cmp <(tail -n +2 ref_CpG.bedGraph) <(tail -n +2 ours_CpG.bedGraph) && echo "IDENTICAL beyond the track line (which names the -o prefix): $(wc -l < ours_CpG.bedGraph) lines"

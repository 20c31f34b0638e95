#!/bin/bash
# round 6: where the configs[2] command (--CHG --CHH with trimming, 128 Mb) spends its time: the host profile of three runs, and the same with more emitter threads
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; D=/tmp/c3; mkdir -p $D; cd $D
$R/tools/_build/mdk_synth -o s -L 128000000 -c 30 -s 1234 -j 16 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel; X="--CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146"
$M extract s.fa s.bam -@ 64 $X -o warm > /dev/null 2>&1
for e in "" "MDK_NO_OUTPUT=1"; do
  for i in 1 2; do sleep 1; t0=$(date +%s.%N); env $e MDK_HOST_PROFILE=1 $M extract s.fa s.bam -@ 64 $X -o out 2> err.txt; t1=$(date +%s.%N)
    python3 -c "print('[$e] wall %.3f' % ($t1-$t0))"; grep -E "plan open|inflate\+frame|handing" err.txt | cut -c1-420; done
done 2>&1 | tee $O/r06c3_profile.txt
ls -la out_C*.bedGraph | awk '{print $5, $9}'

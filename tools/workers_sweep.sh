#!/bin/bash
# host pipeline depth sweep on the GPU box: wall-clock of `MethylDackel extract` over one synthetic BAM for several worker caps
# usage: workers_sweep.sh <length> <threads> <cap,cap,...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=${1:-64000000}; TH=${2:-64}; CAPS=${3:-12,24,48}
W=/tmp/ws_work; mkdir -p $W; cd $W
[ -f s.bam ] || $R/tools/_build/mdk_synth -o s -L $L -c 30 -s 99 > /dev/null
ls -la s.bam | awk '{print "bam bytes", $5}'
TIMEFORMAT="%R"
for cap in ${CAPS//,/ }; do
  for rep in 1 2; do
    t=$( { time MDK_WORKERS=$cap MDK_HOST_PROFILE=1 timeout 300 $R/methyldackel_amd/_build/MethylDackel extract s.fa s.bam -o out_$cap -@ $TH 2> err_$cap.txt; } 2>&1 )
    echo "workers<=$cap threads $TH rep $rep: $t s | $(grep 'mdk main' err_$cap.txt | sed 's/.*loop: //') | $(grep 'mdk host' err_$cap.txt | sed 's/.*mdk host. //')"
  done
done
cmp out_$(echo $CAPS | cut -d, -f1)_CpG.bedGraph out_$(echo $CAPS | awk -F, '{print $NF}')_CpG.bedGraph && echo "outputs identical across caps"

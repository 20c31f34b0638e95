#!/bin/bash
# round 6: settings compared by INTERLEAVED runs (A B C A B C ...: the box's drift hits all alike), medians of wall / inside.
# (MDK_AB_ARGS: extra options of the command, e.g. "--CHG --CHH")
# usage: tools/round6/gpu_ab.sh TAG COPIES(0: 32 Mb, 1: 128 Mb, K: K x 128 Mb) REPS "ENV.." "ENV.." ...   ("-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; K=$2; REPS=$3; shift 3
D=${MDK_AB_DIR:-/tmp}/mdk_ab_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
if [ "$K" = 0 ]; then $R/tools/_build/mdk_synth -o s -L 32000000 -c 30 -s 1234 > /dev/null; F=s
else $R/tools/_build/mdk_synth -o s -L 128000000 -c 30 -s 1234 > /dev/null; F=s; if [ "$K" -gt 1 ]; then $R/tools/_build/mdk_replicate s xl $K > /dev/null 2>&1; F=xl; fi; fi
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 $MDK_AB_ARGS -o warm > /dev/null 2>&1
n=0; for setting in "$@"; do n=$((n+1)); : > w_$n.txt; : > i_$n.txt; done
for rep in $(seq $REPS); do
  n=0
  for setting in "$@"; do
    n=$((n+1)); [ "$setting" = "-" ] && setting=""
    sleep 1; t0=$(date +%s.%N); env $setting MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 $MDK_AB_ARGS -o out 2> err_$n.txt; rc=$?; t1=$(date +%s.%N)
    python3 -c "print('%.3f' % ($t1-$t0))" >> w_$n.txt
    grep -o "total [0-9.]*s" err_$n.txt | head -1 | tr -dc '0-9.\n' >> i_$n.txt
    [ $rc != 0 ] && echo "rc=$rc for [$setting]"
  done
done
n=0
for setting in "$@"; do
  n=$((n+1))
  python3 - "$setting" w_$n.txt i_$n.txt <<'PY' | tee -a $O/${TAG}_ab.txt
import sys, statistics as st
w = [float(x) for x in open(sys.argv[2]).read().split()]; i = [float(x) for x in open(sys.argv[3]).read().split()]
print("[%s] wall median %.3f (min %.3f) inside median %.3f (min %.3f)  runs: %s" % (sys.argv[1], st.median(w), min(w), st.median(i), min(i), " ".join("%.3f/%.3f" % p for p in zip(w, i))))
PY
done

#!/bin/bash
# round 6: the preparation kernels on a library with a sequencer's read names (38 letters) and a bisulfite aligner's aux fields (XM:Z, 150 letters):
# the 96-byte windows (every record read from HBM by its lane), the 128-byte windows, and the handle's own choice.  usage: gpu_prep_illumina.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$1; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PREP_BENCH_SYNTH=--illumina
for w in 0 1 ""; do
  if [ -n "$w" ]; then export MDK_SCAN_WIDE=$w; else unset MDK_SCAN_WIDE; fi
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== MDK_SCAN_WIDE=[${w:-unset}] $(cat /tmp/pl.json)"
  python $R/tools/round5/kt_largest.py /tmp/pl_kt k_prep
done 2>&1 | tee $O/${T}_prep_illumina.txt

#!/bin/bash
# round 6: the dense-context pileup of build variants (kbench: 8 resident S1 intervals, --CHG --CHH with trimming), each after the dense parity tests.  usage: gpu_dense.sh TAG variant...
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$1; shift; O=$R/gpurun_out; mkdir -p $O; cd $R
V="default:"
for v in "$@"; do
  V="$V;$v:MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v"
  MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q -x -k "CH or edge or fixture" 2>&1 | tail -1
done
timeout 600 python tools/kbench.py --resident 16 --variants "$V" --cmds "dense:--CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146" 2>&1 | tee $O/${T}_kbench_dense.txt

#!/bin/bash
# round 5, call p: the dense-context kernel compiled for 6 waves per SIMD (68 VGPRs, no spilled registers, three workgroups per CU) against 8 (64, five spilled)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
timeout 600 python tools/kbench.py --resident 16 --variants "default:;qw6:MDK_BUILD_DIR=$R/methyldackel_amd/_exp_qw6" --cmds 'dense:--CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146;cpg:' 2>&1 | tee $O/kbench_dense.txt
MDK_BUILD_DIR=$R/methyldackel_amd/_exp_qw6 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1

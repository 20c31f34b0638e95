#!/bin/bash
# round 3, GPU call: dense-context kernel with fewer instructions per site pair (adjacent list entries per lane, 24-bit multiplies in the
# overlap rule, strand-0 check hoisted) against the previous build (methyldackel_amd/_exp_base), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/kbench.py --resident 16 --variants "new:;base:MDK_BUILD_DIR=$R/methyldackel_amd/_exp_base;new2:;base2:MDK_BUILD_DIR=$R/methyldackel_amd/_exp_base" --cmds "cpg:;all:--CHG --CHH;dense6:--CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146" 2>&1 | tee $O/r03o_kbench.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r03o_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r03o_pytest_gpu.log

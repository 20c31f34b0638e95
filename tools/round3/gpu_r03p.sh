#!/bin/bash
# round 3, GPU call: pileup kernels without the spills across the main loop (scan addresses, context codes), QL variants; parity subset
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
E=$R/methyldackel_amd
timeout 900 python tools/kbench.py --resident 16 --variants "new:;base:MDK_BUILD_DIR=$E/_exp_base;ql4:MDK_BUILD_DIR=$E/_exp_ql4;ql16:MDK_BUILD_DIR=$E/_exp_ql16;new2:" --cmds "cpg:;all:--CHG --CHH" 2>&1 | tee $O/r03p_kbench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_zoo.py tests/test_gpu_scaled_configs.py -m gpu -q -x > $O/r03p_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r03p_pytest.log

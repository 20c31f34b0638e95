#!/bin/bash
# round 5, call b: the whole -m gpu suite on the hardened tree (hot path first, stress test last)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log

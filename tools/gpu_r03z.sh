#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 300 python -m pytest tests/test_gpu_mbias.py -m gpu -q -x 2>&1 | tail -2 | tr '\n' ' '; echo " [stage run $i]"; done | tee $O/r03z_mbias.txt

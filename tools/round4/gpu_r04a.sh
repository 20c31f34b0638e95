#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
lscpu | head -20 > gpurun_out/r04a_box.txt; numactl -H >> gpurun_out/r04a_box.txt 2>&1
timeout 1500 python tools/round4/r04a.py > gpurun_out/r04a_stdout.txt 2> gpurun_out/r04a_stderr.txt; echo rc=$?
tail -5 gpurun_out/r04a_stderr.txt

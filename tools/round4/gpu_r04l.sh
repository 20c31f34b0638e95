#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
R04_VARIANTS=xl,xl_agent,xl_norel,large timeout 1200 python tools/round4/r04l.py > gpurun_out/r04l_stdout.txt 2> gpurun_out/r04l_stderr.txt; echo rc=$?
cat gpurun_out/r04l_e2e.txt | cut -c1-700; tail -3 gpurun_out/r04l_stderr.txt

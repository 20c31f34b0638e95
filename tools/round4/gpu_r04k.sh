#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python tools/round4/r04k.py > gpurun_out/r04k_stdout.txt 2> gpurun_out/r04k_stderr.txt; echo rc=$?
grep "^==\|^## \|replicate\|synth" gpurun_out/r04k_e2e.txt | cut -c1-160; tail -3 gpurun_out/r04k_stderr.txt

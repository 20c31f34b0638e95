#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/e2e_wall_probe.py 32000000 > $O/r03n_wall32.json 2> $O/r03n_wall32.err; echo "rc=$?"; cat $O/r03n_wall32.err | tail -12

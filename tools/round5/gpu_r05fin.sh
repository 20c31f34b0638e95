#!/bin/bash
# round 5, the round's record: whole -m gpu suite, smoke, bench.py as the driver runs it, rocprofv3 kernel trace + counter passes of the same
# command (tools/gpu_round.sh), then the long form of the stress loop on this tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
bash tools/gpu_round.sh r05fin tests 2>&1 | tail -40
mkdir -p $O/r05fin_stress
timeout 900 python tools/round5/stress.py $O/r05fin_stress 60 mbias_hostprep extract_hostprep perread_hostprep mbias_default extract_default perread_default 2>&1 | tail -8

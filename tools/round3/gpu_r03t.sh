#!/bin/bash
# round 3, GPU call: staging blocks registered by a background thread -- parity subset, then the 128 Mb and 32 Mb commands (own clock)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_inflate.py tests/test_gpu_multi.py -m gpu -x -q > $O/r03t_pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/r03t_pytest.log
D=/tmp/mdk_big; mkdir -p $D; tools/_build/mdk_synth -o $D/s -L 128000000 -c 30 -s 1 > /dev/null 2>&1; tools/_build/mdk_synth -o $D/t -L 32000000 -c 30 -s 2 > /dev/null 2>&1
M=$R/methyldackel_amd/_build/MethylDackel
for f in s t; do for cfg in "" "MDK_HOST_INFLATE=1"; do for rep in 1 2 3 4; do
    mkdir -p $D/o; ( cd $D/o; sleep 0.5; env $cfg MDK_HOST_PROFILE=1 timeout 90 $M extract $D/$f.fa $D/$f.bam -@ 64 -o out 2> err.txt; echo "$f [$cfg] rep $rep rc $? $(grep -o 'total [0-9.]*s' err.txt | head -1) $(grep -o 'staging blocks registered: [0-9]* ([0-9]* MB) in [0-9.]*s' err.txt) $(grep -o 'wait-for-chunk [0-9.]*s submit [0-9.]*s download [0-9.]*s' err.txt) $(md5sum out_CpG.bedGraph | cut -c1-8)" )
done; done; done 2>&1 | tee $O/r03t_register_bg.txt

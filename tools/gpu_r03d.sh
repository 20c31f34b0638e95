#!/bin/bash
# round 3, GPU call 4: device inflate integrated into the command: new tests, a sample of the old suite, end-to-end timings
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q > $O/r03d_pytest_inflate.log 2>&1; echo "inflate tests rc=$?"; tail -15 $O/r03d_pytest_inflate.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prep.py tests/test_gpu_multi.py -m gpu -x -q > $O/r03d_pytest_some.log 2>&1; echo "some tests rc=$?"; tail -5 $O/r03d_pytest_some.log
D=/tmp/r03d; mkdir -p $D; cd $D
$R/tools/_build/mdk_synth -o $D/s32 -L 32000000 -c 30 -s 1589478401 > $D/s32.json
M=$R/methyldackel_amd/_build/MethylDackel
for mode in "" "MDK_HOST_INFLATE=1" "MDK_DEVICE_INFLATE_ONLY=1"; do for rep in 1 2 3; do sleep 0.5; t0=$(date +%s.%N); env $mode MDK_HOST_PROFILE=1 $M extract s32.fa s32.bam -@ 64 -o out_$rep 2> err.txt; rc=$?; t1=$(date +%s.%N); python3 -c "print('mode [$mode] rep $rep rc $rc wall %.3f s' % ($t1 - $t0))"; grep -E "mdk main\] plan|pieces inflated" err.txt | sed 's/.*records found/records found/' | cut -c1-400; done; md5sum out_1_CpG.bedGraph; done 2>&1 | tee $O/r03d_e2e32.txt

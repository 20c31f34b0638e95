#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/r03e; mkdir -p $D; cd $D
$R/tools/_build/mdk_synth -o $D/s2 -L 2000000 -c 30 -s 5 > $D/s2.json
M=$R/methyldackel_amd/_build/MethylDackel
MDK_DEVICE_INFLATE_ONLY=1 MDK_GPU_PIECE_MB=3 MDK_HOST_PROFILE=1 $M extract s2.fa s2.bam -@ 8 -o dev > $O/r03e_dev.out 2> $O/r03e_dev.err; echo "dev rc=$?"; tail -8 $O/r03e_dev.err
MDK_HOST_INFLATE=1 $M extract s2.fa s2.bam -@ 8 -o host 2>/dev/null; md5sum dev_CpG.bedGraph host_CpG.bedGraph; wc -l dev_CpG.bedGraph host_CpG.bedGraph
cd $R; timeout 900 python -m pytest tests/test_gpu_inflate.py -m gpu -q > $O/r03e_pytest_inflate.log 2>&1; echo "inflate tests rc=$?"; tail -40 $O/r03e_pytest_inflate.log

#!/bin/bash
# round 5, call x: how busy the device is over the 512 Mb run (rocprofv3 kernel trace of the command itself; the command now leaves the
# ordinary way under a profiler, so that the trace is written), then the whole -m gpu suite on that tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
D=/dev/shm/mdk_e2e; mkdir -p $D; L=128000000
[ -f $D/s$L.bam.bai ] || tools/_build/mdk_synth -o $D/s$L -L $L -c 30 -s $((0x5EED0001 + 1000)) > /dev/null 2>&1
[ -f $D/x${L}x4.bam ] || tools/_build/mdk_replicate $D/s$L $D/x${L}x4 4 > /dev/null 2>&1
mkdir -p $D/out; cd $D/out; export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1 MDK_NO_RANKS=1 MDK_HOST_PROFILE=1
$R/methyldackel_amd/_build/MethylDackel extract $D/x${L}x4.fa $D/x${L}x4.bam -@ 64 -o warm > /dev/null 2>&1
rm -rf /tmp/busy_kt
s=$(date +%s.%N)
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/busy_kt -o kt -- $R/methyldackel_amd/_build/MethylDackel extract $D/x${L}x4.fa $D/x${L}x4.bam -@ 64 -o x > /dev/null 2> $O/cmd.err
e=$(date +%s.%N)
grep -h "total\|pieces inflated\|teams, summed" $O/cmd.err | cut -c1-400
find /tmp/busy_kt -name "*.csv" | head -5
python $R/tools/round5/gpu_busy.py /tmp/busy_kt $(python -c "print($e - $s)") | tee $O/gpu_busy_512Mb.json
cd $R; unset MDK_HOST_PROFILE MDK_NO_RANKS
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log

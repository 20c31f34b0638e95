cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; D=/tmp/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1 TMPDIR=/tmp
$R/tools/_build/mdk_synth -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl 8 > /dev/null 2>&1; F=xl
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 -o warm > /dev/null 2>&1
sleep 1; MDK_WATCHDOG=10 MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> $O/r06pu_wd.err
grep -E "total|host threads inside|teams, summed|reader:" $O/r06pu_wd.err | cut -c1-700
rm -rf /tmp/busy_kt
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/busy_kt -o kt -- $M extract $F.fa $F.bam -@ 64 -o x > /dev/null 2> $O/r06pu_cmd.err
mkdir -p $O/r06pu_trace; find /tmp/busy_kt -name '*kernel_trace.csv' -exec cp {} $O/r06pu_trace/kernel_trace.csv \; ; find /tmp/busy_kt -name '*memory_copy_trace.csv' -exec cp {} $O/r06pu_trace/memory_copy_trace.csv \;
python3 $R/tools/round6/trace_summary.py $O/r06pu_trace/kernel_trace.csv

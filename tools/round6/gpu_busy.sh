#!/bin/bash
# round 6: how busy the device is over the 512 Mb run (rocprofv3 kernel + memory-copy trace of the command itself).  usage: tools/round6/gpu_busy.sh TAG [COPIES]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06busy}; K=${2:-4}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/dev/shm/mdk_e2e; mkdir -p $D; L=128000000
[ -f $D/s$L.bam.bai ] || tools/_build/mdk_synth -o $D/s$L -L $L -c 30 -s $((0x5EED0001 + 1000)) > /dev/null 2>&1
[ -f $D/x${L}x$K.bam ] || tools/_build/mdk_replicate $D/s$L $D/x${L}x$K $K > /dev/null 2>&1
mkdir -p $D/out; cd $D/out; export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1 MDK_NO_RANKS=1 MDK_HOST_PROFILE=1
$R/methyldackel_amd/_build/MethylDackel extract $D/x${L}x$K.fa $D/x${L}x$K.bam -@ 64 -o warm > /dev/null 2>&1
rm -rf /tmp/busy_kt
s=$(date +%s.%N)
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/busy_kt -o kt -- $R/methyldackel_amd/_build/MethylDackel extract $D/x${L}x$K.fa $D/x${L}x$K.bam -@ 64 -o x > /dev/null 2> $O/${TAG}_cmd.err
e=$(date +%s.%N)
grep -h "total\|pieces inflated\|teams, summed" $O/${TAG}_cmd.err | cut -c1-500
python $R/tools/round5/gpu_busy.py /tmp/busy_kt $(python -c "print($e - $s)") | tee $O/${TAG}_gpu_busy.json
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/busy_kt/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        k = r.get("Direction", "?"); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9; b = int(r.get("Bytes", r.get("Size", 0)) or 0)
        by.setdefault(k, [0, 0.0, 0]); by[k][0] += 1; by[k][1] += d; by[k][2] += b
    print({k: {"copies": v[0], "sum_s": round(v[1], 4), "GB": round(v[2] / 1e9, 3)} for k, v in by.items()})
PY
mkdir -p $O/${TAG}_trace; find /tmp/busy_kt -name '*kernel_trace.csv' -exec cp {} $O/${TAG}_trace/kernel_trace.csv \; ; find /tmp/busy_kt -name '*memory_copy_trace.csv' -exec cp {} $O/${TAG}_trace/memory_copy_trace.csv \;
rm -rf $D

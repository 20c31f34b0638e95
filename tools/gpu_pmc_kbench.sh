#!/bin/bash
# PMC passes over tools/kbench.py's child for one option set: tools/gpu_pmc_kbench.sh TAG "<extract options>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=$1; EX="$2"; mkdir -p $O /tmp/kbench
cd /tmp; export TMPDIR=/tmp
python $R/tools/kbench.py --resident 16 --cmds "x:$EX" > /dev/null 2>&1     # makes the data
export KB_EXTRA="$EX"
B="python $R/tools/kbench.py --child --resident 16 --data /tmp/kbench"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/${TAG}_sq -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/${TAG}_sq2 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum -d $O/${TAG}_fetch -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum -d $O/${TAG}_tcp -o p -- $B > /dev/null 2>&1
python - <<PY
import csv, collections, glob
for d in ("sq","sq2","fetch","tcp"):
    fs = glob.glob("$O/${TAG}_"+d+"/**/p_counter_collection.csv", recursive=True)
    if not fs: print(d, "no data"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = "multi" if "k_pileup_multi" in r["Kernel_Name"] else ("single" if "k_pileup" in r["Kernel_Name"] else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        for c,v in agg[k].items(): print(d, k, c, round(sum(v)/len(v)), len(v))
PY

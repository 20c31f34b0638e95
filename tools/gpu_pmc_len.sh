#!/bin/bash
# PMC passes for a large interval: tools/gpu_pmc_len.sh <length> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; L=$1; TAG=$2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --length $L"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- $B > $O/${TAG}_bench.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/${TAG}_sq -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_fetch -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_wl -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/${TAG}_cache -o p -- $B > /dev/null 2>&1
python - <<PY
import csv, collections
print(open("$O/${TAG}_kt/kt_kernel_stats.csv").read().splitlines()[1])
for d in ("sq","fetch","wl","cache"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open("$O/${TAG}_"+d+"/p_counter_collection.csv")):
        if "k_pileup" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,sum(v)/len(v))
PY
cat $O/${TAG}_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline'], d['config']['segments_per_gpu'])"

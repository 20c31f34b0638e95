#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --passes 8 > $O/r03last_bench.json 2> $O/r03last_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/r03last_bench.json')); r=d['roofline']; print(d['value'], r['family'], r['kernel_ms_per_chunk'], r['frac'], r['traffic'], r['traffic_source']['file'])"; tail -2 $O/r03last_bench.err

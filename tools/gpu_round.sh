#!/bin/bash
# One GPU-box round: parity tests, smoke, headline bench, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-run}
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu.log; tail -3 $O/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${TAG}_smoke.log; tail -2 $O/${TAG}_smoke.log
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cat $O/${TAG}_bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_kt -o kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/${TAG}_prof_pmc_sq -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_prof_pmc_fetch -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_prof_pmc_write_lds -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/${TAG}_prof_pmc_cache -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
find $O -name "*.csv" | head -30

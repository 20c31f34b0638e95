#!/bin/bash
# One GPU-box round: parity tests, smoke, headline bench, rocprofv3 kernel trace + PMC passes.
# Outputs under gpurun_out/; tools/summarize_prof.py turns them into the summaries kept in profiles/.
# usage: tools/gpu_round.sh TAG [tests|notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-run}; D=/tmp/mdk_bench_data
cd $R
if [ "${2:-tests}" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu.log; tail -5 $O/${TAG}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${TAG}_smoke.log; tail -2 $O/${TAG}_smoke.log
fi
timeout 1200 python bench.py --data-dir $D > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cat $O/${TAG}_bench.json; tail -5 $O/${TAG}_bench.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --data-dir $D --no-cpu-baseline --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_kt -o kt -- $B --steps 4 --warmup 1 --passes 16 > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
P="$B --steps 2 --warmup 1 --passes 4"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/${TAG}_prof_pmc_sq -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_prof_pmc_fetch -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_prof_pmc_write_lds -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/${TAG}_prof_pmc_cache -o p -- $P > /dev/null 2>&1
find $O -name "*.csv" | grep ${TAG} | head -30

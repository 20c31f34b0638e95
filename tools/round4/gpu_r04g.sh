#!/bin/bash
# round 4, GPU call g: the wave-parallel k_inflate (tests, piece_bench), k_prep_scan with and without the LDS window (per-kernel times), the command
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_parity.py -m gpu -x -q > $O/r04g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04g_pytest.log
D=/tmp/mdk_r04; mkdir -p $D
[ -f $D/s32000000.bam ] || tools/_build/mdk_synth -o $D/s32000000 -L 32000000 -c 30 -s 11 > /dev/null 2>&1
timeout 300 tools/_build/piece_bench $D/s32000000.bam 64 3 1 > $O/r04g_piece_bench.json 2> $O/r04g_piece_bench.err; echo "piece_bench rc=$?"; cat $O/r04g_piece_bench.json; tail -2 $O/r04g_piece_bench.err
timeout 300 tools/_build/piece_bench $D/s32000000.bam 1024 1 0 > $O/r04g_piece_bench_whole.json 2>> $O/r04g_piece_bench.err; cat $O/r04g_piece_bench_whole.json
( cd /tmp; export TMPDIR=/tmp PREP_BENCH_FAST=1
for v in "" stage0; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  rm -rf /tmp/pl_kt
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  f=$(find /tmp/pl_kt -name '*kernel_stats.csv' | head -1)
  echo "== variant [${v:-default}] $(python -c "import json; d=json.load(open('/tmp/pl.json')); print('prep us/chunk', round(d['prep_ms_per_chunk_8_per_launch']*1000,1))")"
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
done ) 2>&1 | tee $O/r04g_prep_variants.txt
R04_VARIANTS=default timeout 600 python tools/round4/r04b.py r04g 128000000 > $O/r04g_stdout.txt 2> $O/r04g_stderr.txt; echo rc=$?
grep "^==\|^## " $O/r04g_e2e.txt | cut -c1-120; grep -m2 "host threads inside\|uploader:" $O/r04g_e2e.txt | cut -c1-500

#!/bin/bash
# round 5, call d: the preparation without scratch (rare path inlined), record offsets loaded once, ticket sums in one round trip;
# variants: 96 SGPRs for k_prep_segs (8 waves/SIMD), 128-thread workgroups, no LDS window.  Per variant: test_gpu_prep, per-kernel times.
# Then counters of the default build's preparation kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
( cd /tmp; export TMPDIR=/tmp
for v in "" sgpr96 pb128 stage0; do
  if [ -n "$v" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  ( cd $R; timeout 400 python -m pytest tests/test_gpu_prep.py -m gpu -q -x 2>&1 | tail -1 )
  rm -rf /tmp/pl_kt
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_kt -o kt -- python $R/tools/prep_bench.py 16 > /tmp/pl.json 2> /dev/null
  echo "== variant [${v:-default}] $(cat /tmp/pl.json)"
  f=$(find /tmp/pl_kt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep "k_prep" "$f" | awk -F, '{printf "   %-28s calls %s avg %.1f us max %.1f us\n", $1, $2, $4/1000, $7/1000}'
done ) 2>&1 | tee $O/prep_variants.txt
cd /tmp; export TMPDIR=/tmp; unset MDK_BUILD_DIR
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pp_$i
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pp_$i -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1 || echo "prep group $i failed: $grp"
done
python $R/tools/round4/pmc_table.py /tmp/pp_* > $O/prep_pmc.txt; grep -A30 "k_prep_s" $O/prep_pmc.txt | head -80

#!/bin/bash
# round 6: counters of the preparation kernels for build variants (separate --pmc passes).  usage: gpu_preppmc.sh TAG variant...  ("-" = the default build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$1; shift; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != "-" ]; then export MDK_BUILD_DIR=$R/methyldackel_amd/_exp_$v; else unset MDK_BUILD_DIR; fi
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_ATOMIC_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1)); rm -rf /tmp/pp_${v}_$i
    PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pp_${v}_$i -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1 || echo "group $i failed for [$v]: $grp"
  done
  echo "===== variant [$v]"; python $R/tools/round4/pmc_table.py /tmp/pp_${v}_* | grep -A16 -E "^k_prep_(scan|segs)$"
done 2>&1 | tee $O/${T}_prep_pmc.txt

#!/bin/bash
# round 4, GPU call h: where k_inflate and the preparation kernels spend their cycles (instruction mix, waits, LDS) -- counters only, one pass per group
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
D=/tmp/mdk_r04; mkdir -p $D
[ -f $D/s32000000.bam ] || $R/tools/_build/mdk_synth -o $D/s32000000 -L 32000000 -c 30 -s 11 > /dev/null 2>&1
PB="$R/tools/_build/piece_bench $D/s32000000.bam 1024 1 0"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pi_$i /tmp/pp_$i
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pi_$i -o p -- $PB > /dev/null 2>&1 || echo "inflate group $i failed: $grp"
  PREP_BENCH_FAST=1 timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pp_$i -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2>&1 || echo "prep group $i failed: $grp"
done
python $R/tools/round4/pmc_table.py /tmp/pi_* > $O/r04h_inflate_pmc.txt; python $R/tools/round4/pmc_table.py /tmp/pp_* > $O/r04h_prep_pmc.txt
cat $O/r04h_inflate_pmc.txt | head -60; grep -A40 "k_prep_scan" $O/r04h_prep_pmc.txt | head -80

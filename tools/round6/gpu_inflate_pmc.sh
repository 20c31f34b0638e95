#!/bin/bash
# round 6: k_inflate's counters (whole file as one piece: the device full) and its kernel trace.  usage: tools/round6/gpu_inflate_pmc.sh TAG [LENGTH]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06pmc}; L=${2:-32000000}; D=/tmp/inf; mkdir -p $D; cd $D
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
PB="$R/tools/_build/piece_bench s$L.bam 4000 1 0"
timeout 300 $PB > $O/${TAG}_piece_whole.json 2>&1; cat $O/${TAG}_piece_whole.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- $D/../inf/../../$PB > /dev/null 2>&1 || (cd $D && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- $PB > /dev/null 2>&1)
cd $D
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/${TAG}_pmc1 -o p -- $PB > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $O/${TAG}_pmc2 -o p -- $PB > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM FETCH_SIZE WRITE_SIZE -d $O/${TAG}_pmc3 -o p -- $PB > /dev/null 2>&1
python3 $R/tools/round4/pmc_table.py $O/${TAG}_pmc1 $O/${TAG}_pmc2 $O/${TAG}_pmc3 > $O/${TAG}_pmc.txt; sed -n '/k_inflate/,$p' $O/${TAG}_pmc.txt | head -40
find $O/${TAG}_kt -name "*kernel_stats.csv" | head -1 | xargs cat | head -8

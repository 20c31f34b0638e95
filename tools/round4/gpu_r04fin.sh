#!/bin/bash
# round 4, final GPU call: tools/gpu_round.sh (parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes of the bench step) and the same
# evidence for the device inflate (piece_bench under rocprofv3: kernel trace with stats; counters in separate passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/gpu_round.sh r04fin tests
cd /tmp; export TMPDIR=/tmp
D=/tmp/mdk_bench_data; S=$(ls $D/cpu_sample_32000000_*.bam 2>/dev/null | head -1)
if [ -z "$S" ]; then mkdir -p /tmp/mdk_r04; $R/tools/_build/mdk_synth -o /tmp/mdk_r04/s32 -L 32000000 -c 30 -s 11 > /dev/null 2>&1; S=/tmp/mdk_r04/s32.bam; fi
PB="$R/tools/_build/piece_bench $S 1024 1 0"; P64="$R/tools/_build/piece_bench $S 64 3 0"
rm -rf $O/r04fin_inflate_kt $O/r04fin_inflate64_kt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04fin_inflate_kt -o kt -- $PB > $O/r04fin_piece_bench_whole_under_rocprof.json 2> /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04fin_inflate64_kt -o kt -- $P64 > $O/r04fin_piece_bench_64_under_rocprof.json 2> /dev/null
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pi_$i
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pi_$i -o p -- $PB > /dev/null 2>&1 || echo "inflate group $i failed: $grp"
done
python $R/tools/round4/pmc_table.py /tmp/pi_* > $O/r04fin_inflate_pmc.txt
find $O/r04fin_inflate_kt $O/r04fin_inflate64_kt -name "kt_kernel_stats.csv" | head; head -12 $(find $O/r04fin_inflate_kt -name "kt_kernel_stats.csv" | head -1)
grep -A26 "^k_inflate" $O/r04fin_inflate_pmc.txt | head -30

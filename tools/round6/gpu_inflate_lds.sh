#!/bin/bash
# round 6: k_inflate's LDS counters with the per-byte state addressed through the XOR (the build) and straight (-DINF_AUX_STRAIGHT)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; L=32000000; D=/tmp/inf; mkdir -p $D; cd $D; export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
for v in - auxstraight; do
  lib=$R/methyldackel_amd/_build; [ "$v" != "-" ] && lib=$R/methyldackel_amd/_exp_$v
  rm -rf /tmp/il_$v; LD_LIBRARY_PATH=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d /tmp/il_$v -o p -- $R/tools/_build/piece_bench s$L.bam 4000 1 0 > /dev/null 2>&1
  echo "===== [$v]"; python3 $R/tools/round4/pmc_table.py /tmp/il_$v | sed -n '/^k_inflate/,/^k_/p' | head -8
done 2>&1 | tee $O/r06i5_inflate_lds_counters.txt

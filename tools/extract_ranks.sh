#!/bin/bash
# `MethylDackel extract` on N GPUs of this node, one process per GPU (csrc/host/mdk_ranks.c):
#     tools/extract_ranks.sh N [extract options] ref.fa aln.bam
# Rank k takes chunks k, k+N, ... of the schedule and GPU k; rank 0 collects the site buffers (ncclSend/ncclRecv) and writes the files.
# The same thing under torchrun:  MDK_TORCHRUN=1 python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
#     --master-port P methyldackel_amd/_build/MethylDackel extract [options] ref.fa aln.bam      (the ranks meet on port P+1; MDK_PORT overrides)
set -u
N=${1:?number of GPUs}; shift
BIN="$(cd "$(dirname "$0")/.." && pwd)/methyldackel_amd/_build/MethylDackel"
export MDK_WORLD=$N MASTER_ADDR=${MASTER_ADDR:-127.0.0.1} MASTER_PORT=${MASTER_PORT:-$((20000 + RANDOM % 20000))} HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 1; r < N; r++)); do MDK_RANK=$r "$BIN" extract "$@" > /dev/null & pids+=($!); done
MDK_RANK=0 "$BIN" extract "$@"; rc=$?
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc

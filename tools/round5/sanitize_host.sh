#!/bin/bash
# TEST TOOL (CPU): the command's host code (csrc/host/*.c, main.c) built with ThreadSanitizer and with AddressSanitizer + UBSan, the device
# stand-in linked in (tools/dev_standin.c; tools/round5/san_stubs.c for the entry points it lacks), and run under the round's fuzzers:
#   tools/round5/sanitize_host.sh OUTDIR [runs]
# Findings of round 5 (all fixed): unlocked looks at flags another thread sets (mdk_bam.dev, header_done, io_end, the lazily set crc_wanted,
# pslot.prepared), a signed overflow where a member that starts inside a record is walked speculatively (note_records).
R=$(cd "$(dirname "$0")/../.." && pwd); O=${1:-/tmp/mdk_san}; N=${2:-40}; mkdir -p $O
SRC="$R/methyldackel_amd/csrc/host/main.c $(ls $R/methyldackel_amd/csrc/host/mdk_*.c) $R/tools/dev_standin.c $R/tools/round5/san_stubs.c"
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=thread -fPIC -pthread -I$R/include -o $O/MethylDackel_tsan $SRC -lz -lm -ldl || exit 1
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -pthread -I$R/include -o $O/MethylDackel_asan $SRC -lz -lm -ldl || exit 1
export TSAN_OPTIONS="report_thread_leaks=0 exitcode=66" ASAN_OPTIONS="detect_leaks=0:exitcode=67" UBSAN_OPTIONS="halt_on_error=1:exitcode=68"
rc=0
for b in tsan asan; do
  MDK_FUZZ_CLI=$O/MethylDackel_$b python3 $R/tools/round5/fuzz_options.py 7$N $N $O/opt_$b | tail -1 || rc=1
  MDK_FUZZ_CLI=$O/MethylDackel_$b python3 $R/tools/round5/fuzz_ranks.py 8$N $((N / 2)) $O/ranks_$b | tail -1 || rc=1
done
exit $rc

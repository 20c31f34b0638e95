#!/bin/bash
# round 3, GPU call: preparation kernels with wide loads (v3) -- prep tests, timings, per-kernel trace; ranks-mode tests; the 128 Mb command
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py tests/test_gpu_inflate.py -m gpu -x -q > $O/r03i_pytest_prep.log 2>&1; echo "prep+parity+inflate tests rc=$?"; tail -6 $O/r03i_pytest_prep.log
timeout 300 python tools/prep_bench.py 16 > $O/r03i_prep_bench.json 2> $O/r03i_prep_bench.err; echo "prep_bench rc=$?"; cat $O/r03i_prep_bench.json; tail -3 $O/r03i_prep_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o p -- python $R/tools/prep_bench.py 16 > /dev/null 2> $O/r03i_prof.err; f=$(find /tmp/prof_i -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/r03i_prep_kernel_stats.csv && head -12 "$f" | cut -c1-200 )
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_scaled_configs.py -m gpu -x -q > $O/r03i_pytest_ranks.log 2>&1; echo "ranks tests rc=$?"; tail -15 $O/r03i_pytest_ranks.log
# the command on a 128 Mb sample (hybrid inflate with many device pieces): must end, three times, identical outputs to host-only inflate
D=/tmp/mdk_big; mkdir -p $D; tools/_build/mdk_synth -o $D/s -L 128000000 -c 30 -s 1 > /dev/null 2>&1
for mode in "" "MDK_HOST_INFLATE=1"; do for rep in 1 2; do
  mkdir -p $D/o$rep; ( cd $D/o$rep; s=$(date +%s.%N); env $mode MDK_HOST_PROFILE=1 timeout 90 $R/methyldackel_amd/_build/MethylDackel extract $D/s.fa $D/s.bam -@ 64 -o out 2> err.txt; rc=$?; e=$(date +%s.%N); echo "128Mb mode [$mode] rep $rep rc $rc wall $(echo "$e - $s" | bc) s"; grep -h "mdk main\|pieces" err.txt | cut -c1-400; md5sum out_CpG.bedGraph )
done; done 2>&1 | tee $O/r03i_128mb.txt

#!/bin/bash
# round 4, GPU call o: where the kernel spends the command's exit (kernel stacks sampled between _exit and reaping)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=/tmp/mdk_r04; mkdir -p $D
tools/_build/mdk_synth -o $D/s128 -L 128000000 -c 30 -s 11 > /dev/null 2>&1
cd $D
for v in "-" "MDK_GPU_INFLATE_TEAMS=8" "MDK_HOST_INFLATE=1" "MDK_NO_MMAP=1"; do
  echo "===== env $v" >> $O/r04o_exit_stacks.txt
  timeout 200 python $R/tools/exit_stack_probe.py 5 "$v" -- extract $D/s128.fa $D/s128.bam -@ 64 -o $D/out >> $O/r04o_exit_stacks.txt 2>&1
done
grep "^###\|^=====" $O/r04o_exit_stacks.txt

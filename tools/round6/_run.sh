cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; D=/tmp/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -o s -L 32000000 -c 30 -s 1234 > /dev/null
M=$R/methyldackel_amd/_build/MethylDackel
$M extract s.fa s.bam -@ 64 -o warm > /dev/null 2>&1
for i in 1 2 3; do sleep 1; t0=$(date +%s.%N); MDK_HOST_PROFILE=1 $M extract s.fa s.bam -@ 64 -o out 2> err.txt; t1=$(date +%s.%N); python3 -c "print('wall %.3f, ended at epoch %.3f' % ($t1-$t0, $t1))"; grep -E "reaper|leaving at|entered at|total|slow|device closed" err.txt | cut -c1-300; done

cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -j 16 -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl 4 > /dev/null 2>&1; F=xl
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 -o warm > /dev/null 2>&1
for i in 1 2; do sleep 1; MDK_WATCHDOG=5 MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> $O/r06pg_wd$i.err; grep -E "slow|total" $O/r06pg_wd$i.err | cut -c1-330; done
cmp out_CpG.bedGraph warm_CpG.bedGraph | head -2
for setting in "-" "MDK_NO_REAP=1" "-" "MDK_GROUPS_IN_FLIGHT=4"; do
  [ "$setting" = "-" ] && setting=""
  line=""
  for rep in 1 2 3; do
    sleep 1; t0=$(date +%s.%N); env $setting MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> err.txt; rc=$?; t1=$(date +%s.%N)
    inner=$(grep -o "total [0-9.]*s" err.txt | head -1 | tr -dc '0-9.')
    line="$line $(python3 -c "print('%.3f/%s' % ($t1-$t0, '$inner'))")"
    [ $rc != 0 ] && line="$line rc=$rc"
  done
  echo "[$setting] wall/inside:$line" | tee -a $O/r06pg_sweep.txt
  grep -E "plan open" err.txt | cut -c1-400 | tee -a $O/r06pg_sweep.txt
done

#!/bin/bash
# round 6: k_sites_pack (the device puts a group's ordered sites and status into pinned host memory) against the collector's copies:
# the command-level parity tests, then the 512 Mb run both ways with the host profile's collector lines, outputs compared.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r06pack}; K=${2:-4}; RUNS=${3:-3}
cd $R; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_scaled_configs.py -m gpu -x -q 2>&1 | tail -5 | tee $O/${TAG}_pytest.log
D=/dev/shm/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -j 16 -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl $K > /dev/null 2>&1; F=xl
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 -o warm > /dev/null 2>&1
for setting in "-" "MDK_NO_PACK=1" "-" "MDK_NO_PACK=1" "MDK_GROUPS_IN_FLIGHT=5"; do
  [ "$setting" = "-" ] && setting=""
  line=""
  for rep in $(seq $RUNS); do
    sleep 1; t0=$(date +%s.%N); env $setting MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> err.txt; rc=$?; t1=$(date +%s.%N)
    inner=$(grep -o "total [0-9.]*s" err.txt | head -1 | tr -dc '0-9.')
    line="$line $(python3 -c "print('%.3f/%s' % ($t1-$t0, '$inner'))")"
    [ $rc != 0 ] && line="$line rc=$rc"
  done
  echo "[$setting] wall/inside:$line" | tee -a $O/${TAG}_sweep.txt
  grep -E "host threads inside|plan open|reader:|teams, summed" err.txt | cut -c1-600 | tee -a $O/${TAG}_sweep.txt
  cmp out_CpG.bedGraph warm_CpG.bedGraph && echo "  same output as the first run" | tee -a $O/${TAG}_sweep.txt
done
MDK_NO_PACK=1 $M extract $F.fa $F.bam -@ 64 -o np > /dev/null 2>&1; cmp np_CpG.bedGraph warm_CpG.bedGraph && echo "packed == copied" | tee -a $O/${TAG}_sweep.txt
# dense contexts: buffers grow past the first guess
$M extract s128.fa s128.bam -@ 64 --CHG --CHH -o d1 2> /dev/null; MDK_NO_PACK=1 $M extract s128.fa s128.bam -@ 64 --CHG --CHH -o d2 2>/dev/null
cmp d1_CpG.bedGraph d2_CpG.bedGraph && cmp d1_CHH.bedGraph d2_CHH.bedGraph && cmp d1_CHG.bedGraph d2_CHG.bedGraph && echo "dense: packed == copied" | tee -a $O/${TAG}_sweep.txt
for s in "" "MDK_NO_PACK=1"; do t0=$(date +%s.%N); env $s $M extract s128.fa s128.bam -@ 64 --CHG --CHH -o d1 2> /dev/null; t1=$(date +%s.%N); python3 -c "print('dense 128 Mb [$s] %.3f s' % ($t1-$t0))" | tee -a $O/${TAG}_sweep.txt; done

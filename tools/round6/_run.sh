cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; D=/tmp/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
$R/tools/_build/mdk_synth -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl 4 > /dev/null 2>&1; F=xl
M=$R/methyldackel_amd/_build/MethylDackel
$M extract $F.fa $F.bam -@ 64 -o warm > /dev/null 2>&1
for setting in "-" "MDK_SERIAL_FRAMING=1 MDK_NO_POPULATE=1" "-" "MDK_SERIAL_FRAMING=1 MDK_NO_POPULATE=1"; do
  [ "$setting" = "-" ] && setting=""
  sleep 1; env $setting MDK_WATCHDOG=10 MDK_HOST_PROFILE=1 $M extract $F.fa $F.bam -@ 64 -o out 2> err.txt
  echo "== [$setting]"; grep -E "total|host threads inside|teams, summed|reader:" err.txt | grep -v "^\[wd\]" | cut -c1-600
  cp err.txt $O/r06pp_$(echo "$setting" | tr -c 'A-Z_=1\n' '_' | cut -c1-20)_$RANDOM.err
done

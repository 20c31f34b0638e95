cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; D=/tmp/mdk_e2e_$$; mkdir -p $D; trap "rm -rf $D" EXIT; cd $D
$R/tools/_build/mdk_synth -o s128 -L 128000000 -c 30 -s 1234 > /dev/null
$R/tools/_build/mdk_replicate s128 xl 4 > /dev/null 2>&1
cat xl.bam > /dev/null
$R/tools/_build/read_probe xl.bam 8; $R/tools/_build/read_probe xl.bam 2; df -h /tmp | tail -1; mount | grep -E " /tmp | / " | head -3

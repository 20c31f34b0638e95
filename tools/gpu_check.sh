#!/bin/bash
# the GPU suite and the smoke test on the current tree (no bench, no profiles): tools/gpu_check.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-check}; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
timeout 120 python tools/prep_bench.py 16 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k: round(v*1000,1) for k,v in d.items() if k.startswith('prep_ms') or k.startswith('pileup_ms')})"

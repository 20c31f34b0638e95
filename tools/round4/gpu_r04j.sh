#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04j_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r04j_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04j_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r04j_smoke.log
s=$(date +%s); timeout 1500 python bench.py > $O/r04j_bench.json 2> $O/r04j_bench.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"; tail -25 $O/r04j_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04j_bench.json"))
print("value", d["value"], "ms/chunk", d["config"]["ms_per_chunk"])
for k,v in d["roofline"]["kernels"].items(): print(k, round(v["kernel_ms"],4), round(v["frac"],4))
for k in ("cpu_baseline","e2e_cli","e2e_large","e2e_xl","inflate","legs_error"):
    if k in d: print(k, json.dumps(d[k])[:900])
PY

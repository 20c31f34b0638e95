#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_prep.py tests/test_gpu_parity.py tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py tests/test_zoo.py -m gpu -x -q > gpurun_out/r04f_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04f_pytest.log
timeout 600 python bench.py --no-cpu-baseline --data-dir /tmp/mdk_bench_data > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f_bench.json"))
print("value", d["value"], "ms/chunk", d["config"]["ms_per_chunk"])
for k,v in d["roofline"]["kernels"].items(): print(k, v["kernel_ms"], v["frac"])
print("dense", d.get("dense_contexts",{}).get("kernel_ms"), d.get("dense_contexts",{}).get("frac"))
PY
R04_VARIANTS=default,gteams4,slowexit timeout 900 python tools/round4/r04b.py r04f 128000000 > gpurun_out/r04f_stdout.txt 2> gpurun_out/r04f_stderr.txt; echo rc=$?
grep "^==\|^## " gpurun_out/r04f_e2e.txt | cut -c1-120; tail -3 gpurun_out/r04f_stderr.txt

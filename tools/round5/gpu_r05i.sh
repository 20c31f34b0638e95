#!/bin/bash
# round 5, call i: whole -m gpu suite on the tree with the slab reaper, mbias over the device inflate, the new preparation; then bench.py as the
# driver runs it (in-place wall clock on top, queue leg, XXL sample, live counter passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
t0=$(date +%s); timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; tail -c 600 $O/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r05i/bench.json"))
print({k: d.get(k) for k in ("value","ms_per_step")}, d["roofline"]["family"], round(d["roofline"]["frac"],4), d["roofline"]["kernel_ms"], d["roofline"]["traffic"], (d["roofline"]["traffic_source"] or {}).get("measured"))
for k in ("e2e_cli","e2e_large","e2e_xl","e2e_xxl"):
    e=d.get(k)
    if e: print(k, round(e["seconds"],3), e["runs"], "x", round(e.get("speedup_vs_cpu_all_cores", e.get("speedup_vs_cpu_baseline",0)),2), "detached", round(e["detached"]["seconds"],3), "queue", (e.get("queue") or {}).get("runs"), "ident", e["identical_to_oracle"])
print("dense", round(d["dense_contexts"]["frac"],4), "inflate", d["inflate"].get("device_full",{}).get("GBps_compressed"), d.get("legs_error"))
P

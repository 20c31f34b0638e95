#!/usr/bin/env python3
"""MEASUREMENT TOOL: the device preparation and the pileup over R resident 1 Mb chunks of the S1 workload, one chunk and eight chunks
per launch (HIP events inside libmdk_hip).  usage: prep_bench.py [R=16] [extra extract options]"""
import ctypes as C, json, os, subprocess, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
extra = sys.argv[2:]
work = Path(tempfile.mkdtemp(prefix="mdk_prepbench_"))
mdk.build()
subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(work / "s"), "-L", str(1_000_000 * R), "-c", "30", "-s", str(0x5EED0001)] + os.environ.get("PREP_BENCH_SYNTH", "").split(), check=True, capture_output=True)      # PREP_BENCH_SYNTH: extra generator options (--illumina: long read names and aux fields)
plan = mdk.Plan([str(work / "s.fa"), str(work / "s.bam"), "--chunkSize", "1000000", "-@", "16"] + extra + ["-o", str(work / "o")])
plan.set_prep(1)
cfg = plan.dev_cfg(); cfg.n_slots = R
dev = mdk.Device(cfg); dev.set_prep(plan.prep_cfg())
L = mdk.lib_hip()
recs = rawb = 0
for i in range(R):
    c = plan.next_chunk(); plan.ensure_reference(dev, c.tid); dev.upload_raw(i, c.raw); dev.launch(i); dev.download(i)
    recs += c.raw.n_records; rawb += sum(c.raw.range[k].bytes for k in range(c.raw.n_ranges))
slots = (C.c_int * R)(*range(R)); ms = C.c_float(0)
out = {"R": R, "records_per_chunk": recs // R, "record_bytes_per_chunk": rawb // R, "extra": extra}
import os
FAST = bool(os.environ.get("PREP_BENCH_FAST"))           # only the 8-chunk launches (per-kernel traces of build variants); 2: the pileup's 8-chunk launches too (bench.py's counter passes)
if not FAST:
    assert L.md_dev_bench_prep(dev.h, 0, 3, 30, C.byref(ms)) == 0; out["prep_ms_one_chunk_same_slot"] = ms.value
for per in ((8,) if FAST else (1, 2, 4, 8)):
    assert L.md_dev_bench_prep_rotate(dev.h, slots, R, per, 2 * (R // per), 20 * (R // per), C.byref(ms)) == 0, L.md_dev_last_error()
    out[f"prep_ms_per_chunk_{per}_per_launch"] = ms.value / per
for per in ((8,) if os.environ.get("PREP_BENCH_FAST") == "2" else () if FAST else (1, 8)):
    br = dev.bench_rotate(list(range(R)), 8, 100, per_launch=per)
    out[f"pileup_ms_per_chunk_{per}_per_launch"] = br.ms_pileup / per
out["prep_GBps_8_per_launch"] = (rawb / R) / (out["prep_ms_per_chunk_8_per_launch"] / 1e3) / 1e9
print(json.dumps(out))
dev.close(); plan.close()

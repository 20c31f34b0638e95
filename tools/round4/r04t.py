#!/usr/bin/env python3
"""round 4, GPU call t: why the command's runs scatter more when bench.py starts them than in a stand-alone series.
mode A: this process holds what bench.py holds at that point (a torch context, a plan with its threads, a device handle with resident chunks);
mode B: it holds nothing; mode C: as A, handles closed before the runs."""
import os, re, statistics, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
mode = sys.argv[1]
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
T = REPO / "tools/_build"
if not (work / "s128.bam").exists(): subprocess.run([str(T / "mdk_synth"), "-o", str(work / "s128"), "-L", "128000000", "-c", "30", "-s", "11"], check=True, capture_output=True)
if not (work / "s16.bam").exists(): subprocess.run([str(T / "mdk_synth"), "-o", str(work / "s16"), "-L", "16000000", "-c", "30", "-s", "5"], check=True, capture_output=True)
held = None
if mode in ("A", "C"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
    plan = mdk.Plan([str(work / "s16.fa"), str(work / "s16.bam"), "--chunkSize", "1000000", "-@", "32", "-o", str(work / "held")])
    plan.set_prep(1); cfg = plan.dev_cfg(); cfg.n_slots = 18
    dev = mdk.Device(cfg, device=0); dev.set_prep(plan.prep_cfg())
    for k in range(16):
        ch = plan.next_chunk(); plan.ensure_reference(dev, ch.tid); dev.upload_raw(k, ch.raw); dev.launch(k); dev.download(k)
    held = (plan, dev)
    if mode == "C": dev.close(); plan.close(); held = None
def gone(marker):
    t_end = time.time() + 8
    while time.time() < t_end:
        alive = False
        for pid in os.listdir("/proc"):
            if pid.isdigit() and int(pid) != os.getpid():
                try:
                    if open(f"/proc/{pid}/comm").read().strip() == "MethylDackel" and open(f"/proc/{pid}/stat").read().rsplit(") ", 1)[1][0] != "Z": alive = True; break
                except OSError: pass
        if not alive: return
        time.sleep(0.02)
walls, ins, rdy = [], [], []
for rep in range(7):
    time.sleep(0.3); gone(str(work / "s128.bam"))
    d = work / f"t_{mode}"; d.mkdir(exist_ok=True)
    t0 = time.perf_counter()
    r = mdk.run_cli([str(work / "s128.fa"), str(work / "s128.bam"), "-@", "64", "-o", "out"], cwd=d, env={"MDK_HOST_PROFILE": "1"}, timeout=120)
    walls.append(time.perf_counter() - t0)
    m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr); ins.append(float(m.group(1)) if m else -1)
    m = re.search(r"device ready at ([0-9.]+)s", r.stderr); rdy.append(float(m.group(1)) if m else -1)
    if rep in (1, 5):
        for l in r.stderr.splitlines():
            if l.startswith("[mdk hip] warm-up") or l.startswith("[mdk main] staging") or l.startswith("[mdk main] plan open") or "first chunk" in l or "reader:" in l: print("     ", l[:700])
print(f"== mode {mode}: walls {' '.join('%.3f' % w for w in walls)} | median {statistics.median(walls):.3f} | inside {' '.join('%.3f' % w for w in ins)} | device ready {' '.join('%.3f' % w for w in rdy)}", flush=True)

#!/usr/bin/env python3
"""round 4, GPU call x: three against five groups of chunks in flight (MDK_GROUPS_IN_FLIGHT) at 128 and 512 Mb, runs one second apart"""
import os, re, statistics, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / "r04x_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
T = REPO / "tools/_build"
subprocess.run([str(T / "mdk_synth"), "-o", str(work / "s128"), "-L", "128000000", "-c", "30", "-s", "11"], check=True, capture_output=True)
subprocess.run([str(T / "mdk_replicate"), str(work / "s128"), str(work / "y4"), "4"], check=True, capture_output=True, text=True)
def ours(name, env, tag, reps):
    walls, ins = [], []
    for rep in range(reps):
        time.sleep(1.0)
        d = work / f"o_{tag}"; d.mkdir(exist_ok=True)
        t0 = time.perf_counter()
        r = mdk.run_cli([str(work / f"{name}.fa"), str(work / f"{name}.bam"), "-@", "64", "-o", "out"], cwd=d, env=dict(env, MDK_HOST_PROFILE="1"), timeout=120)
        walls.append(time.perf_counter() - t0)
        m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr); ins.append(float(m.group(1)) if m else -1)
        if rep == 1:
            for l in r.stderr.splitlines():
                if l.startswith("[mdk main] plan open") or "host threads inside" in l: say("     ", l[:900])
    say(f"== {name} [{tag}] walls {' '.join('%.3f' % w for w in walls)} | median {statistics.median(walls):.3f} | inside {' '.join('%.3f' % w for w in ins)} | median {statistics.median(ins):.3f}")
for rep in range(2):
    for g in ("3", "5"):
        ours("s128", {"MDK_GROUPS_IN_FLIGHT": g}, f"one128_g{g}_{rep}", 4)
    for g in ("3", "5"):
        ours("y4", {"MDK_GROUPS_IN_FLIGHT": g}, f"xl512_g{g}_{rep}", 3)

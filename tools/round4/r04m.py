#!/usr/bin/env python3
"""round 4, GPU call m: after the removal of the persistent preparation kernels -- repeated end-to-end runs (watchdog on) on the one-contig
128 Mb input, its 4-contig 512 Mb replica and the 16-contig 8.7 GB input, with 3 and 5 device inflate teams"""
import os, re, statistics, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / "r04m_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
T = REPO / "tools/_build"
subprocess.run([str(T / "mdk_synth"), "-o", str(work / "s32"), "-L", "32000000", "-c", "30", "-s", "11"], check=True, capture_output=True)
subprocess.run([str(T / "mdk_synth"), "-o", str(work / "s128"), "-L", "128000000", "-c", "30", "-s", "11"], check=True, capture_output=True)
subprocess.run([str(T / "mdk_replicate"), str(work / "s32"), str(work / "x16"), "16"], check=True, capture_output=True, text=True)
subprocess.run([str(T / "mdk_replicate"), str(work / "s128"), str(work / "y4"), "4"], check=True, capture_output=True, text=True)
def ours(name, env, tag, reps, limit=40):
    walls, ins = [], []
    for rep in range(reps):
        time.sleep(0.3)
        d = work / f"o_{tag}_{rep}"; d.mkdir(exist_ok=True)
        e = dict(os.environ); e.update(env); e.update({"MDK_HOST_PROFILE": "1", "MDK_WATCHDOG": "1", "MDK_NO_RANKS": "1"})
        t0 = time.perf_counter()
        with open(d / "err.txt", "w") as ef:
            p = subprocess.Popen([str(mdk.CLI), "extract", str(work / f"{name}.fa"), str(work / f"{name}.bam"), "-@", "64", "-o", "out"], cwd=d, env=e, stdout=subprocess.DEVNULL, stderr=ef)
            try: rc = p.wait(timeout=limit)
            except subprocess.TimeoutExpired: p.kill(); p.wait(); rc = "TIMEOUT"
        wall = time.perf_counter() - t0
        err = (d / "err.txt").read_text()
        m = re.search(r"total ([0-9.]+)s; chunks prepared", err)
        walls.append(wall); ins.append(float(m.group(1)) if m else -1)
        slow = rc != 0 or (len(walls) > 1 and wall > 1.6 * min(walls))
        if slow or rep == 0:
            say(f"## {name} [{tag}] rep {rep} rc {rc} wall {wall:.3f} inside {ins[-1]}")
            lines = [l for l in err.splitlines() if l.startswith("[mdk")]
            for l in (lines if len(lines) < 24 else lines[:8] + ["..."] + lines[-14:]): say("     ", l[:900])
    say(f"== {name} [{tag}] walls {' '.join('%.3f' % w for w in walls)} | median {statistics.median(walls):.3f} | inside {' '.join('%.3f' % w for w in ins)}")
ours("s128", {}, "one128", 10)
ours("s128", {"MDK_GPU_INFLATE_TEAMS": "5"}, "one128_t5", 6)
ours("y4", {}, "xl512", 5)
ours("y4", {"MDK_GPU_INFLATE_TEAMS": "5"}, "xl512_t5", 5)
ours("x16", {}, "x16", 4)
ours("x16", {"MDK_GPU_INFLATE_TEAMS": "5"}, "x16_t5", 3)

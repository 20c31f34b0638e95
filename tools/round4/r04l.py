#!/usr/bin/env python3
"""round 4, GPU call l: the stall seen on the 16-contig XL input, with the watchdog's account of where the pipeline stands (MDK_WATCHDOG=1)"""
import os, re, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / "r04l_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(work / "s32"), "-L", "32000000", "-c", "30", "-s", "11"], check=True, capture_output=True)
for k in (4, 16):
    subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(work / "s32"), str(work / f"x{k}"), str(k)], check=True, capture_output=True, text=True)
def ours(name, env, tag, reps, limit=40):
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
        say(f"## {name} [{tag}] rep {rep} rc {rc} wall {wall:.3f} inside {m.group(1) if m else '?'}")
        if rc != 0 or wall > 3:
            lines = [l for l in err.splitlines() if l.startswith("[mdk")]
            for l in lines[:6] + ["..."] + lines[-14:]: say("     ", l[:700])
VAR = os.environ.get("R04_VARIANTS", "xl,xl_agent,large").split(",")
if "xl" in VAR: ours("x16", {}, "xl", 2)
if "xl_agent" in VAR: ours("x16", {"MDK_PREP_AGENT_SCOPE": "1"}, "xl_agent", 2)
if "xl_norel" in VAR: ours("x16", {"MDK_NO_EARLY_RELEASE": "1"}, "xl_norel", 2)
if "large" in VAR: ours("x4", {}, "large", 6)
if "large_agent" in VAR: ours("x4", {"MDK_PREP_AGENT_SCOPE": "1"}, "large_agent", 4)

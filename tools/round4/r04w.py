#!/usr/bin/env python3
"""round 4, GPU call w (two collector threads vs one; the rest as s: : staging blocks registered from runtime-up, shared piece streams, open waits for warm streams): code objects of the preparation / inflate kernels, copy-engine queues and carved device blocks made ahead on side threads of
the warm-up -- the first upload and the first group (75 ms and 60 ms in call p's time series) and the 128 Mb / 512 Mb walls"""
import os, re, statistics, subprocess, sys, time, shutil
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / "r04w_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
T = REPO / "tools/_build"
subprocess.run([str(T / "mdk_synth"), "-o", str(work / "s128"), "-L", "128000000", "-c", "30", "-s", "11"], check=True, capture_output=True)
subprocess.run([str(T / "mdk_replicate"), str(work / "s128"), str(work / "y4"), "4"], check=True, capture_output=True, text=True)
def ours(name, env, tag, reps, keep=None, limit=60):
    walls, ins = [], []
    for rep in range(reps):
        time.sleep(0.8)
        d = work / f"o_{tag}_{rep}"; d.mkdir(exist_ok=True)
        e = dict(os.environ); e.update(env); e.update({"MDK_HOST_PROFILE": "1", "MDK_NO_RANKS": "1"})
        with open(d / "err.txt", "w") as ef:
            t0 = time.perf_counter()
            p = subprocess.Popen([str(mdk.CLI), "extract", str(work / f"{name}.fa"), str(work / f"{name}.bam"), "-@", "64", "-o", "out"], cwd=d, env=e, stdout=subprocess.DEVNULL, stderr=ef)
            try:
                while p.poll() is None:
                    if time.perf_counter() - t0 > limit: raise subprocess.TimeoutExpired("x", limit)
                    time.sleep(0.001)
                rc = p.returncode
            except subprocess.TimeoutExpired: p.kill(); p.wait(); rc = "TIMEOUT"
            wall = time.perf_counter() - t0
        time.sleep(0.4)
        err = (d / "err.txt").read_text()
        m = re.search(r"total ([0-9.]+)s; chunks prepared", err)
        walls.append(wall); ins.append(float(m.group(1)) if m else -1)
        if rc != 0 or rep == 1:
            say(f"## {name} [{tag}] rep {rep} rc {rc} wall {wall:.3f} inside {ins[-1]}")
            for l in [l for l in err.splitlines() if l.startswith("[mdk")][:16]: say("     ", l[:1300])
        if keep and rep < 2: shutil.copy(d / "err.txt", O / f"r04w_{keep}_{rep}.txt")
    say(f"== {name} [{tag}] walls {' '.join('%.3f' % w for w in walls)} | median {statistics.median(walls):.3f} | inside {' '.join('%.3f' % w for w in ins)} | median {statistics.median(ins):.3f}")
for rep in range(2):
    ours("s128", {}, f"one128_c2_{rep}", 4)
    ours("s128", {"MDK_COLLECTORS": "1"}, f"one128_c1_{rep}", 4)
    ours("y4", {}, f"xl512_c2_{rep}", 2)
    ours("y4", {"MDK_COLLECTORS": "1"}, f"xl512_c1_{rep}", 2)

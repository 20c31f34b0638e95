#!/usr/bin/env python3
"""round 4, GPU call k: the command on an XL input (32 Mb sample x 16 contigs = 8.7 GB of BAM) and on a 128 Mb-sized one (x 4), with its own account of
where the time goes; looking for the steady-state rate and for the occasional run that stalls."""
import os, re, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / "r04k_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
t0 = time.time(); subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(work / "s32"), "-L", "32000000", "-c", "30", "-s", "11"], check=True, capture_output=True); say(f"synth 32 Mb: {time.time() - t0:.1f} s")
for k in (4, 16):
    t0 = time.time(); r = subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(work / "s32"), str(work / f"x{k}"), str(k)], check=True, capture_output=True, text=True); say(f"replicate x{k}: {time.time() - t0:.1f} s {r.stdout.strip()}")
def ours(name, env, tag, reps, threads="64"):
    res = []
    for rep in range(reps):
        time.sleep(0.3)
        d = work / f"o_{tag}_{rep}"; d.mkdir(exist_ok=True)
        t0 = time.perf_counter()
        r = mdk.run_cli([str(work / f"{name}.fa"), str(work / f"{name}.bam"), "-@", threads, "-o", "out"], cwd=d, env=dict(env, MDK_HOST_PROFILE="1"), timeout=300)
        wall = time.perf_counter() - t0
        m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr); inside = float(m.group(1)) if m else -1
        res.append((wall, inside, r.stderr))
        say(f"## {name} [{tag}] rep {rep} rc {r.returncode} wall {wall:.3f} inside {inside:.3f} md5 {subprocess.run(['md5sum', str(d / 'out_CpG.bedGraph')], capture_output=True, text=True).stdout[:12]}")
    ins = sorted(x[1] for x in res); med = ins[len(ins) // 2]
    for k, (wall, inside, err) in enumerate(res):
        if k == 0 or inside > 1.5 * med:
            say(f"   profile of rep {k}:")
            for l in err.splitlines():
                if l.startswith("[mdk"): say("     ", l[:900])
    say(f"== {name} [{tag}] inside median {med:.3f}, walls {['%.3f' % x[0] for x in res]}")
VAR = os.environ.get("R04_VARIANTS", "xl,large").split(",")
if "xl" in VAR:
    ours("x16", {}, "xl", 3)
    ours("x16", {"MDK_PREP_AGENT_SCOPE": "1"}, "xl_agent", 2)
    ours("x16", {"MDK_GPU_INFLATE_TEAMS": "5"}, "xl_teams5", 2)
    ours("x16", {"MDK_HOST_INFLATE": "1"}, "xl_hostinflate", 1)
if "large" in VAR:
    ours("x4", {}, "large", 10)
    ours("x4", {"MDK_PREP_AGENT_SCOPE": "1"}, "large_agent", 6)

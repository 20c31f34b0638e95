#!/usr/bin/env python3
"""round 4, GPU call b: the command after the consumer was split into uploader / collector / reference threads (3 groups in flight on 3
streams), staging blocks registered next to the uploader, file pages and staging memory given back before the exit."""
import os, re, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04b"
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / f"{TAG}_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
work = Path("/tmp/mdk_r04"); work.mkdir(exist_ok=True)
synth = str(REPO / "tools/_build/mdk_synth"); oracle = str(REPO / "oracle/_build/mdk_oracle")
sizes = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["128000000", "32000000"])]
for L in sizes:
    if not (work / f"s{L}.bam.bai").exists():
        t0 = time.time(); subprocess.run([synth, "-o", str(work / f"s{L}"), "-L", str(L), "-c", "30", "-s", "11"], check=True, capture_output=True); say(f"synth {L}: {time.time() - t0:.1f} s")
    d = work / f"oracle{L}"
    if not (d / "out_CpG.bedGraph").exists():
        d.mkdir(exist_ok=True); t0 = time.time()
        subprocess.run([oracle, "extract", str(work / f"s{L}.fa"), str(work / f"s{L}.bam"), "-@", "64", "--chunkSize", "250000", "-o", "out"], cwd=d, check=True, capture_output=True); say(f"oracle {L} -@ 64: {time.time() - t0:.2f} s")
def ours(L, env, tag, reps=3, threads="64"):
    walls = []
    for rep in range(reps):
        time.sleep(0.3)
        d = work / f"o_{tag}_{rep}"; d.mkdir(exist_ok=True)
        t0 = time.perf_counter()
        r = mdk.run_cli([str(work / f"s{L}.fa"), str(work / f"s{L}.bam"), "-@", threads, "-o", "out"], cwd=d, env=dict(env, MDK_HOST_PROFILE="1"), timeout=120)
        wall = time.perf_counter() - t0; walls.append(wall); t_end = time.time()
        m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr)
        same = (d / "out_CpG.bedGraph").exists() and (d / "out_CpG.bedGraph").read_bytes() == (work / f"oracle{L}" / "out_CpG.bedGraph").read_bytes()
        ml = re.search(r"leaving at epoch ([0-9.]+)", r.stderr)
        say(f"## {L} [{tag}] rep {rep} rc {r.returncode} wall {wall:.3f} inside {m.group(1) if m else '?'} exit {t_end - float(ml.group(1)) if ml else -1:.3f} identical {same}")
        for l in r.stderr.splitlines():
            if l.startswith("[mdk"): say("   ", l[:700])
        if r.returncode: say(r.stderr[-1500:])
    say(f"== {L} [{tag}] walls {['%.3f' % w for w in walls]} median {sorted(walls)[len(walls) // 2]:.3f}")
VAR = os.environ.get("R04_VARIANTS", "default,notrim,hostinflate").split(",")
for L in sizes:
    if "default" in VAR: ours(L, {}, f"default{L}", 5)
    if "notrim" in VAR: ours(L, {"MDK_NO_TRIM": "1"}, f"notrim{L}", 3)
    if "cap12" in VAR: ours(L, {"MDK_SLAB_CAP": "12"}, f"cap12_{L}", 3)
    if "cap20" in VAR: ours(L, {"MDK_SLAB_CAP": "20"}, f"cap20_{L}", 3)
    if "norelease" in VAR: ours(L, {"MDK_NO_EARLY_RELEASE": "1"}, f"norelease{L}", 3)
    if "norectab" in VAR: ours(L, {"MDK_NO_RECTAB": "1"}, f"norectab{L}", 3)
    if "cap24" in VAR: ours(L, {"MDK_SLAB_CAP": "24"}, f"cap24_{L}", 3)
    if "cap8" in VAR: ours(L, {"MDK_SLAB_CAP": "8"}, f"cap8_{L}", 3)
    if "slowexit" in VAR: ours(L, {"HSA_TOOLS_LIB": ""}, f"slowexit{L}", 3)
    if "noprereg" in VAR: ours(L, {"MDK_NO_PREREG": "1"}, f"noprereg{L}", 2)
    if "hostinflate" in VAR: ours(L, {"MDK_HOST_INFLATE": "1"}, f"hostinflate{L}", 2)
    if "gteams4" in VAR: ours(L, {"MDK_GPU_INFLATE_TEAMS": "4"}, f"gteams4_{L}", 2)
    if "noarena" in VAR: ours(L, {"MDK_NO_ARENA": "1"}, f"noarena{L}", 2)
    if "nocrc" in VAR: ours(L, {"MDK_NO_CRC": "1"}, f"nocrc{L}", 3)
    if "t32" in VAR: ours(L, {}, f"t32_{L}", 2, threads="32")
    if "t128" in VAR: ours(L, {}, f"t128_{L}", 2, threads="128")

#!/usr/bin/env python3
"""round 4, GPU call a: where the command's wall clock goes (start-up itemised, host threads inside the device library, exit vs THP) and what the
CPU baseline's best setting is.  Writes gpurun_out/r04a_*.txt"""
import json, os, re, subprocess, sys, tempfile, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
O = REPO / "gpurun_out"; O.mkdir(exist_ok=True)
out = open(O / "r04a_e2e.txt", "w")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
def vm():
    d = {}
    for l in open("/proc/vmstat"):
        k, v = l.split()
        if k in ("thp_fault_alloc", "thp_fault_fallback", "thp_collapse_alloc", "compact_stall"): d[k] = int(v)
    return d
say("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip(), "cpus:", os.cpu_count())
say(subprocess.run(["free", "-g"], capture_output=True, text=True).stdout)
subprocess.run(["gcc", "-O2", "-o", "/tmp/exit_probe", str(REPO / "tools/exit_probe.c")], check=True)
for gb, thp, pre in ((8, 1, 0), (8, 0, 0), (8, 1, 1), (8, 0, 1)):
    v0 = vm(); r = subprocess.run(["/tmp/exit_probe", str(gb), str(thp), str(pre)], capture_output=True, text=True); v1 = vm()
    say(r.stdout.strip(), {k: v1[k] - v0[k] for k in v0})
work = Path("/tmp/mdk_r04a"); work.mkdir(exist_ok=True)
synth = str(REPO / "tools/_build/mdk_synth")
for L in (32_000_000, 128_000_000):
    if not (work / f"s{L}.bam.bai").exists():
        t0 = time.time(); subprocess.run([synth, "-o", str(work / f"s{L}"), "-L", str(L), "-c", "30", "-s", "11"], check=True, capture_output=True); say(f"synth {L}: {time.time() - t0:.1f} s")
def ours(L, env, tag, reps=3, threads="64"):
    for rep in range(reps):
        time.sleep(0.3)
        d = work / f"o_{tag}_{rep}"; d.mkdir(exist_ok=True)
        v0 = vm(); t0 = time.perf_counter()
        r = mdk.run_cli([str(work / f"s{L}.fa"), str(work / f"s{L}.bam"), "-@", threads, "-o", "out"], cwd=d, env=dict(env, MDK_HOST_PROFILE="1"), timeout=120)
        wall = time.perf_counter() - t0; v1 = vm()
        m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr)
        say(f"## {L} [{tag}] rep {rep} rc {r.returncode} wall {wall:.3f} inside {m.group(1) if m else '?'} vmstat {({k: v1[k] - v0[k] for k in v0})}")
        for l in r.stderr.splitlines():
            if l.startswith("[mdk"): say("   ", l[:600])
for L in (128_000_000, 32_000_000):
    ours(L, {}, f"default{L}", 4)
    ours(L, {"MDK_NO_THP": "1"}, f"nothp{L}", 2)
    ours(L, {"MDK_SLAB_CAP": "12"}, f"cap12_{L}", 2)
    ours(L, {"MDK_NO_PIN": "1"}, f"nopin{L}", 2)
    ours(L, {}, f"t128_{L}", 2, threads="128")
    ours(L, {}, f"t32_{L}", 2, threads="32")
# CPU baseline sweep
oracle = str(REPO / "oracle/_build/mdk_oracle")
best = {}
for L in (32_000_000, 128_000_000):
    for thr in (32, 64, 128, 256):
        for ck in (50_000, 250_000, 1_000_000):
            if L == 128_000_000 and ck == 50_000 and thr < 64: continue
            d = work / f"c_{L}_{thr}_{ck}"; d.mkdir(exist_ok=True)
            ts = []
            for rep in range(2 if L == 32_000_000 else 1):
                t0 = time.perf_counter(); subprocess.run([oracle, "extract", str(work / f"s{L}.fa"), str(work / f"s{L}.bam"), "-@", str(thr), "--chunkSize", str(ck), "-o", "out"], cwd=d, check=True, capture_output=True); ts.append(time.perf_counter() - t0)
            say(f"oracle L={L} -@ {thr} --chunkSize {ck}: {['%.3f' % t for t in ts]}")
            best[(L, thr, ck)] = min(ts)
say("best:", sorted((v, k) for k, v in best.items())[:6])

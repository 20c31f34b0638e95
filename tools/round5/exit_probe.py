#!/usr/bin/env python3
"""MEASUREMENT TOOL (GPU box): what the process exit of `MethylDackel extract` costs with the teardown in place, under switches that change what
is left to take down: wall - (the command's own clock from entry to outputs closed).  usage: exit_probe.py OUTDIR [length=64000000]"""
import json, os, re, statistics, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
out = Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
D = Path("/dev/shm/mdk_exit" if os.path.isdir("/dev/shm") else "/tmp/mdk_exit"); D.mkdir(exist_ok=True)
sp = D / f"s{L}"
if not Path(str(sp) + ".bam.bai").exists():
    subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(sp), "-L", str(L), "-c", "30", "-s", "77"], check=True, capture_output=True)
CLI = REPO / "methyldackel_amd/_build/MethylDackel"
res = {}
for name, env, threads in (("default", {}, "64"), ("no_reap", {"MDK_NO_REAP": "1"}, "64"), ("no_arena", {"MDK_NO_ARENA": "1"}, "64"), ("no_warm_side", {"MDK_NO_WARM_SIDE": "1"}, "64"),
                           ("no_pin", {"MDK_NO_PIN": "1"}, "64"), ("host_inflate", {"MDK_HOST_INFLATE": "1"}, "64"), ("threads128", {}, "128"), ("threads192", {}, "192")):
    ws, ins, last = [], [], ""
    for _ in range(4):
        time.sleep(1.2)
        e = dict(os.environ, MDK_HOST_PROFILE="1", MDK_NO_RANKS="1", MDK_NO_DETACH="1", HSA_DISABLE_COREDUMP_ON_EXCEPTION="1"); e.update(env)
        t = time.perf_counter(); r = subprocess.run([str(CLI), "extract", str(sp) + ".fa", str(sp) + ".bam", "-@", threads, "-o", "x"], cwd=D, env=e, capture_output=True, text=True, timeout=300); w = time.perf_counter() - t
        assert r.returncode == 0, r.stderr[-800:]
        m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr); ws.append(w); ins.append(float(m.group(1)))
        last = [l[:300] for l in r.stderr.splitlines() if "reaper" in l or "leaving at" in l or "device ready" in l]
    res[name] = {"wall": [round(x, 3) for x in ws], "inside": ins, "outside_median": round(statistics.median(a - b for a, b in zip(ws, ins)), 3), "lines": last}
    print(name, res[name]["wall"], ins, "outside", res[name]["outside_median"], flush=True)
(out / "exit_probe.json").write_text(json.dumps(res, indent=1))

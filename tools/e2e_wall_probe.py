#!/usr/bin/env python3
"""MEASUREMENT TOOL: whole-process wall clock of `MethylDackel extract` against the time it reports from inside (MDK_HOST_PROFILE), for
several pauses between runs and with / without registered staging memory.  usage: e2e_wall_probe.py [length_bp=32000000]"""
import json, os, re, subprocess, sys, tempfile, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
work = Path(tempfile.mkdtemp(prefix="mdk_wall_"))
subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(work / "s"), "-L", str(L), "-c", "30", "-s", "11"], check=True, capture_output=True)
out = []
for mode, env in (("default", {}), ("no_pin", {"MDK_NO_PIN": "1"}), ("host_inflate", {"MDK_HOST_INFLATE": "1"}), ("host_inflate_no_pin", {"MDK_HOST_INFLATE": "1", "MDK_NO_PIN": "1"})):
    for pause in (0.3, 1.5):
        walls, inside = [], []
        for rep in range(4):
            time.sleep(pause)
            d = work / f"o_{mode}_{pause}_{rep}"; d.mkdir()
            t0 = time.perf_counter()
            r = mdk.run_cli([str(work / "s.fa"), str(work / "s.bam"), "-@", "64", "-o", "out"], cwd=d, env=dict(env, MDK_HOST_PROFILE="1"), timeout=120)
            walls.append(round(time.perf_counter() - t0, 3))
            m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr)
            inside.append(float(m.group(1)) if m else None)
            assert r.returncode == 0, r.stderr[-500:]
        out.append({"mode": mode, "pause_s": pause, "wall_s": walls, "inside_process_s": inside})
        print(out[-1], file=sys.stderr)
print(json.dumps({"length_bp": L, "runs": out}))

#!/usr/bin/env python3
"""End-to-end wall-clock of `MethylDackel extract` (this build, GPU) vs the CPU oracle on one synthetic BAM (GPU box).
usage: tools/e2e.py <length> <threads[,threads...]> [extra extract options...]"""
import filecmp, json, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, ths, extra = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
d = tempfile.mkdtemp(prefix="mdk_e2e_")
t = time.time(); subprocess.run([f"{REPO}/tools/_build/mdk_synth", "-o", f"{d}/s", "-L", str(L), "-c", "30", "-s", "99"], check=True, capture_output=True); tg = time.time() - t
os.makedirs(f"{d}/o"); os.makedirs(f"{d}/g")
res = {"length": L, "extra": extra, "bam_bytes": os.path.getsize(f"{d}/s.bam"), "synth_s": round(tg, 2)}
t = time.time(); subprocess.run([f"{REPO}/oracle/_build/mdk_oracle", "extract", f"{d}/s.fa", f"{d}/s.bam", "-o", "out"] + extra, cwd=f"{d}/o", check=True, capture_output=True); res["oracle_s"] = round(time.time() - t, 3)
for th in ths:
    best = None
    for rep in range(2):
        t = time.time(); r = subprocess.run([f"{REPO}/methyldackel_amd/_build/MethylDackel", "extract", f"{d}/s.fa", f"{d}/s.bam", "-o", "out", "-@", th] + extra, cwd=f"{d}/g", capture_output=True, text=True, env=dict(os.environ, MDK_HOST_PROFILE="1"))
        dt = time.time() - t; best = dt if best is None else min(best, dt)
    ident = all(filecmp.cmp(f"{d}/o/{f}", f"{d}/g/{f}", shallow=False) for f in os.listdir(f"{d}/o"))
    res[f"gpu_cli_threads_{th}"] = {"seconds": round(best, 3), "rc": r.returncode, "identical": ident, "speedup_vs_oracle": round(res["oracle_s"] / best, 2),
                                    "host_profile": [l for l in r.stderr.strip().splitlines() if l.startswith("[mdk")]}
print(json.dumps(res))

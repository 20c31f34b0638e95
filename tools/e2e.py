#!/usr/bin/env python3
"""End-to-end wall-clock of `MethylDackel extract` (this build, GPU) vs the CPU oracle on one synthetic BAM (GPU box).
usage: tools/e2e.py <length> <threads[,threads...]> [extra extract options...]
env: E2E_MODES=dev,host (device / host chunk preparation), E2E_DATA=dir (keep and reuse the synthetic input)"""
import filecmp, json, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, ths, extra = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
d = os.environ.get("E2E_DATA") or tempfile.mkdtemp(prefix="mdk_e2e_")
os.makedirs(d, exist_ok=True)
s = f"{d}/s{L}"
tg = 0.0
if not os.path.exists(s + ".bam.bai"):
    t = time.time(); subprocess.run([f"{REPO}/tools/_build/mdk_synth", "-o", s, "-L", str(L), "-c", "30", "-s", "99"], check=True, capture_output=True); tg = time.time() - t
w = tempfile.mkdtemp(prefix="mdk_e2e_out_")
res = {"length": L, "extra": extra, "bam_bytes": os.path.getsize(s + ".bam"), "synth_s": round(tg, 2), "host_cores": os.cpu_count()}
for name, opt in (("oracle_1", ["-@", "1"]), ("oracle_all", ["-@", str(os.cpu_count()), "--chunkSize", str(max(50000, L // (4 * (os.cpu_count() or 1))))])):
    os.makedirs(f"{w}/{name}")
    t = time.time(); subprocess.run([f"{REPO}/oracle/_build/mdk_oracle", "extract", s + ".fa", s + ".bam", "-o", "out"] + opt + extra, cwd=f"{w}/{name}", check=True, capture_output=True); res[name + "_s"] = round(time.time() - t, 3)
for mode in os.environ.get("E2E_MODES", "dev,host").split(","):
    env = dict(os.environ, MDK_HOST_PROFILE="1")
    if mode == "host":
        env["MDK_HOST_PREP"] = "1"
    for th in ths:
        g = f"{w}/g_{mode}_{th}"; os.makedirs(g)
        best = None
        for rep in range(3):
            time.sleep(0.6)      # back-to-back commands wait for the previous process' GPU teardown
            t = time.time(); r = subprocess.run([f"{REPO}/methyldackel_amd/_build/MethylDackel", "extract", s + ".fa", s + ".bam", "-o", "out", "-@", th] + extra, cwd=g, capture_output=True, text=True, env=env)
            dt = time.time() - t
            if best is None or dt < best:
                best, prof = dt, [l for l in r.stderr.strip().splitlines() if l.startswith("[mdk")]
        ident = all(filecmp.cmp(f"{w}/oracle_1/{f}", f"{g}/{f}", shallow=False) for f in os.listdir(f"{w}/oracle_1"))
        res[f"cli_{mode}_prep_threads_{th}"] = {"seconds": round(best, 3), "rc": r.returncode, "identical": ident, "x_vs_oracle_1": round(res["oracle_1_s"] / best, 2),
                                              "x_vs_oracle_all": round(res["oracle_all_s"] / best, 2), "host_profile": prof}
print(json.dumps(res, indent=1))

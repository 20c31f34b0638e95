#!/usr/bin/env python3
"""MEASUREMENT TOOL (GPU box): the XL sample (4 x 128 Mb) through `MethylDackel extract` under different splits of the inflate between the host's
threads and the device.  usage: e2e_sweep.py OUTDIR"""
import json, os, re, statistics, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
out = Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
D = Path("/dev/shm/mdk_e2e" if os.path.isdir("/dev/shm") else "/tmp/mdk_e2e"); D.mkdir(exist_ok=True)
L = 128_000_000; sp = D / f"s{L}"; sx = D / f"x{L}x4"
if not Path(str(sp) + ".bam.bai").exists():
    subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(sp), "-L", str(L), "-c", "30", "-s", str(0x5EED0001 + 1000)], check=True, capture_output=True)
if not Path(str(sx) + ".bam").exists():
    subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(sp), str(sx), "4"], check=True, capture_output=True)
CLI = REPO / "methyldackel_amd/_build/MethylDackel"
cfgs = [("default_64", "64", {}), ("gpu12", "64", {"MDK_GPU_INFLATE_TEAMS": "12"}), ("gpu16", "64", {"MDK_GPU_INFLATE_TEAMS": "16"}),
        ("gpu16_piece32", "64", {"MDK_GPU_INFLATE_TEAMS": "16", "MDK_GPU_PIECE_MB": "32"}), ("gpu12_host2", "64", {"MDK_GPU_INFLATE_TEAMS": "12", "MDK_INFLATE_TEAMS": "2"}),
        ("gpu16_t96", "96", {"MDK_GPU_INFLATE_TEAMS": "16"})]
if os.environ.get("E2E_CFGS"): cfgs = [tuple(c) for c in json.loads(os.environ["E2E_CFGS"])]      # [[name, threads, {env}], ...]
res = {}
w = D / "out"; w.mkdir(exist_ok=True)
for name, th, env in cfgs:
    ws, ins, last = [], [], []
    for _ in range(3):
        time.sleep(1.0)
        e = dict(os.environ, MDK_HOST_PROFILE="1", MDK_NO_RANKS="1", HSA_DISABLE_COREDUMP_ON_EXCEPTION="1"); e.update(env)
        t = time.perf_counter(); r = subprocess.run([str(CLI), "extract", str(sx) + ".fa", str(sx) + ".bam", "-@", th, "-o", "x"], cwd=w, env=e, capture_output=True, text=True, timeout=300); ws.append(time.perf_counter() - t)
        assert r.returncode == 0, r.stderr[-800:]
        ins.append(float(re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr).group(1)))
        last = [l[:500] for l in r.stderr.splitlines() if "pieces inflated by" in l or "uploader:" in l or "reaper" in l or "teams, summed" in l]
    res[name] = {"wall": [round(x, 3) for x in ws], "inside": ins, "lines": last}
    print(name, res[name]["wall"], ins, flush=True)
    for l in last: print("    ", l[:500], flush=True)
(out / "e2e_sweep.json").write_text(json.dumps(res, indent=1))

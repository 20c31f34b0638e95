#!/usr/bin/env python3
"""MEASUREMENT TOOL (GPU box): `MethylDackel extract` of this build end to end on the bench's large and XL samples, the teardown in place
(MDK_NO_DETACH=1: the honest wall clock) and detached, with the command's own profile lines; the CPU oracle once per sample at the
setting round 4's sweep found best; `mbias` on the large sample.  usage: e2e.py OUTDIR [copies=4] [build_dir]"""
import json, os, re, statistics, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
out = Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = Path(sys.argv[3]) if len(sys.argv) > 3 else REPO / "methyldackel_amd" / "_build"
CLI = B / "MethylDackel"
D = Path("/dev/shm/mdk_e2e" if os.path.isdir("/dev/shm") else "/tmp/mdk_e2e"); D.mkdir(exist_ok=True)
L = 128_000_000
sp = D / f"s{L}"
if not Path(str(sp) + ".bam.bai").exists():
    t = time.time(); subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(sp), "-L", str(L), "-c", "30", "-s", str(0x5EED0001 + 1000)], check=True, capture_output=True); print(f"synth {time.time() - t:.1f}s", flush=True)
sx = D / f"x{L}x{copies}"
if copies > 1 and not Path(str(sx) + ".bam").exists():
    t = time.time(); subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(sp), str(sx), str(copies)], check=True, capture_output=True); print(f"replicate {time.time() - t:.1f}s", flush=True)


def gone():
    for _ in range(400):
        alive = False
        for pid in os.listdir("/proc"):
            if pid.isdigit():
                try:
                    if open(f"/proc/{pid}/comm").read().strip() == "MethylDackel" and open(f"/proc/{pid}/stat").read().rsplit(") ", 1)[1][0] != "Z":
                        alive = True; break
                except OSError:
                    pass
        if not alive:
            return
        time.sleep(0.02)


def ours(s, cmd, extra, env, runs, gap=1.0):
    ts, inner, prof = [], [], None
    w = D / "out"; w.mkdir(exist_ok=True)
    for _ in range(runs):
        gone(); time.sleep(gap)
        e = dict(os.environ, MDK_HOST_PROFILE="1", MDK_NO_RANKS="1", HSA_DISABLE_COREDUMP_ON_EXCEPTION="1"); e.update(env)
        t = time.perf_counter()
        r = subprocess.run([str(CLI), cmd, str(s) + ".fa", str(s) + ".bam", "-@", "64"] + extra, cwd=w, env=e, capture_output=True, text=True, timeout=600)
        ts.append(time.perf_counter() - t)
        assert r.returncode == 0, r.stderr[-1500:]
        m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr); inner.append(float(m.group(1)) if m else None)
        prof = [l[:600] for l in r.stderr.splitlines() if l.startswith("[mdk")]
    return {"median": statistics.median(ts), "runs": [round(x, 3) for x in ts], "inside": inner, "profile_last": prof}


res = {"build": str(B)}
for name, s in (("large", sp),) + ((("xl", sx),) if copies > 1 else ()):
    t = time.perf_counter()
    subprocess.run([str(REPO / "oracle/_build/mdk_oracle"), "extract", str(s) + ".fa", str(s) + ".bam", "-@", "64", "--chunkSize", "50000", "-o", "gpu"], cwd=D, check=True, capture_output=True)
    cpu = time.perf_counter() - t
    a = ours(s, "extract", ["-o", "gpu"], {}, 3)
    b = ours(s, "extract", ["-o", "gpu"], {"MDK_DETACH": "1"}, 3)
    ident = all((D / "out" / f"gpu_{c}.bedGraph").read_bytes() == (D / f"gpu_{c}.bedGraph").read_bytes() for c in ("CpG",))
    q = ours(s, "extract", ["-o", "gpu"], {}, 4, gap=0.0)          # a queue of samples: back to back, no pause
    res[name] = {"bam_bytes": os.path.getsize(str(s) + ".bam"), "cpu_64x50000_s": round(cpu, 2), "in_place": a, "detached": b, "queue_in_place": q, "identical": ident,
                 "x_in_place": round(cpu / a["median"], 2), "x_detached": round(cpu / b["median"], 2), "x_queue": round(cpu / q["median"], 2)}
    print(name, json.dumps({k: v for k, v in res[name].items() if k not in ("in_place", "detached", "queue_in_place")}), a["runs"], b["runs"], q["runs"], flush=True)
mb = ours(sp, "mbias", ["--noSVG", "--txt"], {}, 3)
mbh = ours(sp, "mbias", ["--noSVG", "--txt"], {"MDK_HOST_INFLATE": "1"}, 2)
res["mbias_large"] = {"in_place": mb, "host_inflate_only": mbh}
print("mbias", mb["runs"], "host inflate only", mbh["runs"], flush=True)
(out / "e2e.json").write_text(json.dumps(res, indent=1))

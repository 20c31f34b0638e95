#!/usr/bin/env python3
"""Kernel-level experiment loop (GPU box): k_pileup time on S1-sized intervals for several option sets and tile sizes.
usage: tools/kbench.py [--resident R] [--variants 'name:env=val,env=val;...'] [--cmds 'cpg:;all:--CHG --CHH']
Each variant runs in a fresh subprocess (the tile size and experimental switches are read at md_dev_open)."""
import argparse, json, os, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent


def child(a):
    sys.path.insert(0, str(REPO))
    import ctypes as C
    import methyldackel_amd as mdk
    R = a.resident
    prefix = Path(a.data) / f"k_{R}"
    plan = mdk.Plan([str(prefix) + ".fa", str(prefix) + ".bam", "--chunkSize", "1000000", "-@", "32"] + a.extra.split() + ["-o", "/tmp/kb_out"])
    plan.set_prep(1)
    cfg = plan.dev_cfg(); cfg.n_slots = R
    dev = mdk.Device(cfg); dev.set_prep(plan.prep_cfg())
    for i in range(R):
        c = plan.next_chunk(); plan.ensure_reference(dev, c.tid); dev.upload_raw(i, c.raw); dev.launch(i); dev.download(i)
    one = dev.bench(0, 10, 300)
    rot = dev.bench_rotate(list(range(R)), 2 * R, 50 * R)
    K = int(os.environ.get("KB_GROUP", "8"))
    grp = dev.bench_rotate(list(range(R)), 8, 100, per_launch=K) if R % K == 0 and R >= K else None
    pm = C.c_float(0); dev.L.md_dev_bench_prep(dev.h, 0, 3, 20, C.byref(pm))
    print(json.dumps({"cached_us": round(one.ms_pileup * 1e3, 2), "hbm_us": round(rot.ms_pileup * 1e3, 2), "algo_MB": round(rot.algo_bytes / 1e6, 2),
                      "frac_hbm": round(rot.algo_bytes / (rot.ms_pileup / 1e3) / 8e12, 4), "tile": one.tile, "tiles": one.n_tiles, "sites": int(rot.n_sites), "prep_us": round(pm.value * 1e3, 1),
                      "group": None if grp is None else {"chunks": K, "us_per_launch": round(grp.ms_pileup * 1e3, 1), "us_per_chunk": round(grp.ms_pileup * 1e3 / K, 2), "frac_hbm": round(grp.algo_bytes / (grp.ms_pileup / 1e3) / 8e12, 4)}}))
    dev.close(); plan.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resident", type=int, default=8)
    ap.add_argument("--variants", default="default:")
    ap.add_argument("--cmds", default="cpg:;all:--CHG --CHH")
    ap.add_argument("--data", default="/tmp/kbench")
    ap.add_argument("--child", action="store_true"); ap.add_argument("--extra", default="")
    a = ap.parse_args()
    if a.child:
        if os.environ.get("KB_EXTRA") is not None:
            a.extra = os.environ["KB_EXTRA"]
        return child(a)
    os.makedirs(a.data, exist_ok=True)
    prefix = Path(a.data) / f"k_{a.resident}"
    if not Path(str(prefix) + ".bam").exists():
        subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(prefix), "-L", str(1_000_000 * a.resident), "-c", "30", "-s", str(0x5EED0001)], check=True, capture_output=True)
    for v in a.variants.split(";"):
        name, _, envs = v.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("="); env[k] = val
        for cm in a.cmds.split(";"):
            cname, _, extra = cm.partition(":")
            r = subprocess.run([sys.executable, __file__, "--child", "--resident", str(a.resident), "--data", a.data, "--extra=" + extra], env=env, capture_output=True, text=True)
            print(f"{name:>14} {cname:>5}  {r.stdout.strip() or r.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Tuning helper (GPU box): run bench.py under a grid of MDK_TILE settings and print one line each.
usage: tools/sweep.py <tile,tile,...> [unused] ["extra extract options"] ["extra mdk_synth options"]"""
import itertools, json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tiles = sys.argv[1].split(",") if len(sys.argv) > 1 else ["256", "512", "768", "1024"]
budgets = sys.argv[2].split(",") if len(sys.argv) > 2 else ["80896"]
extra = sys.argv[3] if len(sys.argv) > 3 else ""
synth_args = sys.argv[4] if len(sys.argv) > 4 else ""
for t, b in itertools.product(tiles, budgets):
    env = dict(os.environ, MDK_LDS_BUDGET=b)
    if t != "auto":
        env["MDK_TILE"] = t
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline"]
    if extra:
        cmd += ["--extra", extra]
    if synth_args:
        cmd += ["--synth-args", synth_args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1]); c = d["config"]; rf = d["roofline"]
        print(f"tile {t:>5} budget {b:>6} -> tile {c['tile']:>5} tiles {c['tiles']:<5} lds {c['lds_bytes_per_workgroup']:>6} "
              f"ms/step {d['ms_per_step']:.4f} pileup_ms {rf['kernel_ms']:.4f} all_ms {rf['all_kernels_ms']:.4f} frac {rf['frac']:.4f} value {d['value']:.3e}", flush=True)
    except Exception as e:
        print("FAILED", t, b, e, r.stderr[-500:], flush=True)

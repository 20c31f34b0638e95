#!/bin/bash
# where a `MethylDackel extract` process spends its wall-clock outside its own main(): start-up before main, and exit after it
# (GPU box).  usage: tools/gpu_exit_probe.sh <length> [extract options...]   env: PROBE_THREADS=n (-@), PROBE_PREFIX="taskset -c ..." (command prefix)
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=${1:-32000000}; shift
W=/tmp/exit_probe; mkdir -p $W; cd $W
[ -f s$L.bam ] || $R/tools/_build/mdk_synth -o s$L -L $L -c 30 -s 99 > /dev/null
for out in /tmp/exit_probe/o /dev/shm/mdk_o; do
  for rep in 1 2 3; do
    sleep 0.7
    python3 - "$R" "$L" "$out" "$@" <<'PY'
import subprocess, sys, time, os, re
R, L, out = sys.argv[1:4]; extra = sys.argv[4:]
env = dict(os.environ, MDK_HOST_PROFILE="1")
t0 = time.time()
r = subprocess.run(os.environ.get("PROBE_PREFIX", "").split() + [f"{R}/methyldackel_amd/_build/MethylDackel", "extract", f"s{L}.fa", f"s{L}.bam", "-o", out, "-@", os.environ.get("PROBE_THREADS", "64")] + extra, capture_output=True, text=True, env=env)
t1 = time.time()
ent = float(re.search(r"entered at epoch ([0-9.]+)", r.stderr).group(1)); lea = float(re.search(r"leaving at epoch ([0-9.]+)", r.stderr).group(1))
sz = sum(os.path.getsize(f) for f in [out + s for s in ("_CpG.bedGraph", "_CHG.bedGraph", "_CHH.bedGraph", ".cytosine_report.txt")] if os.path.exists(f))
rs = re.search(r"\(resident [^)]*", r.stderr); cl = re.search(r"device closed[^\n]*", r.stderr); print("   ", cl.group(0)) if cl else None; print(f"[{rs.group(0) if rs else ''}] out={out} extra={' '.join(extra) or '-'}: wall {t1-t0:.3f} = before main {ent-t0:.3f} + main {lea-ent:.3f} + after main {t1-lea:.3f}   (output {sz/1e6:.0f} MB, rc {r.returncode})", flush=True)
PY
  done
done

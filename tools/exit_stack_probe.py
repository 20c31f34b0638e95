#!/usr/bin/env python3
"""MEASUREMENT TOOL (not part of the product): where the kernel spends the time between the command's _exit and the moment its parent can reap it.
Starts `MethylDackel extract ...` with MDK_HOST_PROFILE=1, waits for the "[mdk main] leaving" line, then samples /proc/PID/stack, the state of
every remaining task and the resident size every ~1 ms until the process is gone; prints the timeline compressed to runs of equal samples.
  exit_stack_probe.py REPS ENVSPEC -- extract args...      ENVSPEC: comma-separated K=V (or "-")"""
import os, sys, time, subprocess, threading, collections
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import methyldackel_amd as mdk

def top_frames(pid, tid=None):
    p = f"/proc/{pid}/stack" if tid is None else f"/proc/{pid}/task/{tid}/stack"
    try:
        with open(p) as f: fr = [l.split()[-1].split("+")[0] for l in f.read().splitlines() if l.strip()]
        return fr
    except OSError as e:
        # not root: the symbol the task sleeps in (wchan), its state and its name are readable by the owner
        base = f"/proc/{pid}" if tid is None else f"/proc/{pid}/task/{tid}"
        try:
            wch = open(base + "/wchan").read().strip() or "0"
            st = open(base + "/stat").read(); comm = st[st.index("(") + 1:st.rindex(")")]; state = st[st.rindex(")") + 2]
            return [f"{comm}:{state}:{wch}"]
        except OSError:
            return [f"<{e.__class__.__name__}>"]

def one(args, env, rep):
    e = dict(os.environ); e.update(env); e.setdefault("MDK_FAST_EXIT", "1"); e.update({"MDK_HOST_PROFILE": "1", "MDK_NO_RANKS": "1"})
    t0 = time.perf_counter()
    p = subprocess.Popen([str(mdk.CLI)] + args, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, bufsize=1)
    t_leave = None; lines = []
    for l in p.stderr:
        lines.append(l)
        if l.startswith("[mdk main] leaving"): t_leave = time.perf_counter(); break
    samples = []
    while p.poll() is None:
        t = time.perf_counter()
        try: tids = os.listdir(f"/proc/{p.pid}/task")
        except OSError: tids = []
        fr = top_frames(p.pid)
        # the deepest frames that say something
        key = ">".join(fr[:6]) if fr else "-"
        others = collections.Counter()
        for tid in tids[:200]:
            if int(tid) == p.pid: continue
            f2 = top_frames(p.pid, tid)
            others[">".join(f2[:3]) if f2 else "-"] += 1
        try:
            with open(f"/proc/{p.pid}/statm") as f: rss = int(f.read().split()[1]) * 4096 // (1 << 20)
        except (OSError, ValueError, IndexError): rss = -1
        samples.append((t, len(tids), rss, key, tuple(others.most_common(2))))
        time.sleep(0.0005)
    t_end = time.perf_counter()
    rest = p.stderr.read()
    if t_leave is None: print("   (no 'leaving' line)", "".join(lines)[-600:], rest[-300:]); t_leave = t_end
    print(f"### rep {rep}: start->leaving {0 if t_leave is None else t_leave - t0:.3f}s, leaving->reaped {0 if t_leave is None else t_end - t_leave:.3f}s, wall {t_end - t0:.3f}s, {len(samples)} samples")
    # compress
    prev = None; start = None; n = 0
    def flush():
        if prev is not None: print(f"   +{start - t_leave:7.3f}s x{n:4d}  tasks {prev[0]:3d} rss {prev[1]:5d} MB  main: {prev[2][:230]}  | others: {prev[3]}")
    for (t, nt, rss, key, oth) in samples:
        cur = (nt, rss // 256 * 256, key, oth[:1])
        if prev is None or cur[0] != prev[0] or cur[2] != prev[2]:
            flush(); prev = cur; start = t; n = 0
        n += 1
    flush()
    sys.stdout.flush()

if __name__ == "__main__":
    reps = int(sys.argv[1]); envspec = sys.argv[2]; assert sys.argv[3] == "--"
    env = dict(kv.split("=", 1) for kv in envspec.split(",")) if envspec != "-" else {}
    for r in range(reps): one(sys.argv[4:], env, r); time.sleep(0.3)

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X `MethylDackel extract` hot path.

Metric (BASELINE.json): CpG calls/s, synthetic 30x paired-end WGBS, CpG-only extract, in 1 Mb chunks (configs[1], "S1").

Workload.  Every rank holds R (default 16) DIFFERENT S1-sized intervals resident in HBM as `MethylDackel extract` has them after
the upload: the chunks' BAM records as they lie in the inflated file (about 56 MB each, 0.9 GB together -- beyond the 256 MiB
Infinity Cache, so every launch streams its inputs from HBM).  One STEP is one pass of the hot path over a batch of P x R chunks
(the R resident intervals presented P times in rotation), and for every chunk the WHOLE device work the command does for it:
the preparation kernels (admission, strand, file-order compaction, name table, pairing, CIGAR -> segments: k_prep_zero /
k_prep_scan / k_prep_segs) and the pileup (k_pileup_multi), eight chunks per launch of each kernel as extract_main launches them
(md_dev_launch_group).  Inside a step every launch is issued and collected (site counts read back) with two launches queued on
one in-order stream; with N ranks the kernels write into send buffers and the results of a launch travel to rank 0 with one
ncclSend/ncclRecv exchange (libmdk_hip's md_comm, RCCL over xGMI) while the next launch is computed.  The loop is libmdk_hip's
md_bench_run (C); Python only brackets it.

`python bench.py --gpus N` starts its N ranks itself when it was not started by torch.distributed.run (WORLD_SIZE unset).

Also on the JSON line:
  roofline     -- the kernel family that takes the most device time per chunk, and under "kernels" every family: algorithmic bytes
                  per launch / HIP-event time per launch while rotating over the R intervals on one stream, vs 8 TB/s
  cpu_baseline -- the CPU oracle (`oracle/`, "port") end to end on a 32 Mb sample of the same workload with all host cores and
                  with one thread; 3 runs each, median (BASELINE.md's protocol); e2e_cli = `MethylDackel extract` of this build on
                  the same file under the same protocol
  e2e_large    -- the same comparison on a 128 Mb sample (all cores only), where start-up and teardown no longer dominate
  streamed     -- the resident loop's chunks with H2D upload + kernels + D2H of the sites per chunk (registered staging, two slots)
"""
import argparse
import shutil
import ctypes as C
import json
import os
import statistics
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
S1_SEED = 0x5EED0001
GROUP = 8                    # chunks per launch of each kernel (md_dev_launch_group) = chunks per exchange


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, 127.0.0.1 rendezvous), relay rank 0's line"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out); sys.stdout.flush()
    sys.exit(max(abs(x) for x in rcs))


def median_run(fn, runs=3):
    ts = []
    for _ in range(runs):
        t1 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t1)
    return statistics.median(ts), ts


# ------------------------------------------------------------------------------------------------------------------------------------
# The CPU baseline and the end-to-end legs (rank 0 of an N = 1 run).  Every leg has a price in seconds; a leg the run's time budget
# (--budget-s) no longer holds is left out and named in "legs_skipped" -- the line itself must always come out.
# ------------------------------------------------------------------------------------------------------------------------------------
HUMAN_LADDER = [248, 242, 198, 190, 181, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]      # hg38's chr1..22, X, Y in Mb


def _wait_gone(limit=8.0):
    t_end = time.time() + limit
    while time.time() < t_end:
        alive = False
        for pid in os.listdir("/proc"):
            if not pid.isdigit() or int(pid) == os.getpid():
                continue
            try:      # (by name, not by command line: a process that is taking its address space down has none any more)
                if open(f"/proc/{pid}/comm").read().strip() == "MethylDackel" and open(f"/proc/{pid}/stat").read().rsplit(") ", 1)[1][0] != "Z":
                    alive = True; break
            except OSError:
                pass
        if not alive:
            return
        time.sleep(0.02)


def _calls_of(d, name="out_CpG.bedGraph"):
    n = 0
    for line in open(Path(d) / name):
        f = line.split("\t")
        if len(f) == 6:
            n += int(f[4]) + int(f[5])
    return n


def _mem_available_gb():
    """what this process's control group may still take (files in /dev/shm are charged to it as well), or the machine's MemAvailable if that is less:
    a box with 3 TB of memory ran this under a 300 GiB limit, and a leg sized by MemAvailable alone took the box down"""
    avail = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / 1e6
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"), ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                avail = min(avail, max(0.0, (int(v) - int(open(cur).read().strip())) / 1e9))
        except (OSError, ValueError):
            pass
    return avail


class Legs:
    def __init__(self, args, result, data, work, extra, t_start, mdk):
        self.args, self.result, self.data, self.work, self.extra, self.t_start, self.mdk = args, result, Path(data), Path(work), extra, t_start, mdk
        self.oracle = REPO / "oracle/_build/mdk_oracle"
        self.ncores = os.cpu_count() or 1
        self.threads = str(min(64, self.ncores))
        self.skipped = []; self.slow_runs = []; self.n = 0

    def left(self):
        return self.args.budget_s - (time.time() - self.t_start)

    def fits(self, name, price):
        if self.left() >= price:
            return True
        self.skipped.append({"leg": name, "needs_s": price, "left_s": round(self.left(), 1)})
        log(f"[bench] leg {name} left out: needs ~{price:.0f} s, {self.left():.0f} s of the budget left")
        return False

    def synth(self, prefix, lens, cov, seed, par=0, more=()):
        if not Path(str(prefix) + ".bam.bai").exists():
            t1 = time.time()
            subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(prefix), "-L", lens, "-c", str(cov), "-s", str(seed)] + (["-j", str(par)] if par else []) + list(more) + self.args.synth_args.split(),
                           capture_output=True, text=True, check=True)
            log(f"[bench] sample {Path(prefix).name} written in {time.time() - t1:.1f} s")
        return prefix

    def run_oracle(self, sp, name, thr, ck, runs, opts=(), timeout=1800):
        """-> (median seconds, runs, directory, phases of the last run)"""
        self.n += 1
        d = self.work / f"co_{self.n}_{name}"; d.mkdir()
        o = ["-@", str(thr)] + (["--chunkSize", str(ck)] if ck else [])
        ts = []; phases = {}
        for _ in range(runs):
            t1 = time.perf_counter()
            r = subprocess.run([str(self.oracle), "extract", str(sp) + ".fa", str(sp) + ".bam"] + o + list(opts) + self.extra + ["-o", "out"], check=True, capture_output=True, text=True, cwd=d, timeout=timeout,
                               env=dict(os.environ, MDK_ORACLE_PROFILE="1"))
            ts.append(time.perf_counter() - t1)
            phases = {}
            for line in r.stderr.splitlines():
                m = re.match(r"\[oracle\] (.+?)\s+([0-9.]+) s(\s+\(one thread\))?$", line)
                if m:
                    phases[m.group(1).strip()] = {"seconds": float(m.group(2)), "serial": bool(m.group(3))}
                m = re.match(r"\[oracle\] serial phases ([0-9.]+) s of ([0-9.]+) s", line)
                if m:
                    phases["_serial_fraction"] = float(m.group(1)) / max(float(m.group(2)), 1e-9)
        log(f"[bench] oracle {name}: {[round(t, 3) for t in ts]}")
        return statistics.median(ts), ts, d, phases

    def run_ours(self, sp, name, env=None, runs=3, gap=1.0, opts=(), ranks=None, timeout=600):
        """-> (median seconds, runs, directory, all ok, the command's own clock per run)"""
        self.n += 1
        d = self.work / f"cg_{self.n}_{name}"; d.mkdir(); rcs = []; ts = []; inner = []; profs = []
        for _ in range(runs):
            # (outside the clock) runs are measured in isolation: the next one starts a second after the previous command's last process has gone
            # (the driver goes on releasing a process's GPU state for a while; gap = 0: a queue of samples, back to back)
            if gap > 0:
                _wait_gone(); time.sleep(gap)
            t1 = time.perf_counter()
            r = self.mdk.run_cli([str(sp) + ".fa", str(sp) + ".bam", "-@", self.threads] + list(opts) + self.extra + ["-o", "out"], cwd=d, env=dict(env or {}, MDK_HOST_PROFILE="1"), timeout=timeout, ranks=ranks)
            ts.append(time.perf_counter() - t1); rcs.append(r.returncode)
            m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr)          # the command's own clock, entry of extract_main to outputs closed
            inner.append(float(m.group(1)) if m else None)
            profs.append([l[:400] for l in r.stderr.splitlines() if l.startswith("[mdk")])
        log(f"[bench] {name}: wall {[round(t, 3) for t in ts]} inside the process {inner}")
        med = statistics.median(ts)
        for k, t in enumerate(ts):          # a run far off the others is kept with its own account of where the time went, not just as a number
            if t > 1.6 * med and t - med > 0.3:
                self.slow_runs.append({"leg": name, "run": k, "seconds": t, "median": med, "profile": profs[k]})
        return med, ts, d, all(r == 0 for r in rcs), inner

    @staticmethod
    def same(d1, d2):
        fs = sorted(os.listdir(d2))
        return bool(fs) and all((Path(d1) / f).exists() and (Path(d1) / f).read_bytes() == (Path(d2) / f).read_bytes() for f in fs)

    def cpu_setting_sweep(self, sp, tag, settings, runs_best):
        """one run per setting, then `runs_best` runs at the fastest -> (median, runs, directory, phases, setting, sweep)"""
        sweep = []
        for thr, ck in settings:
            if thr > self.ncores:
                continue
            t1, _, _, _ = self.run_oracle(sp, f"{tag}_sweep_{thr}_{ck}", thr, ck, 1)
            sweep.append({"threads": thr, "chunk_size": ck, "seconds": t1})
        best = min(sweep, key=lambda q: q["seconds"])
        t, ts, d, ph = self.run_oracle(sp, f"{tag}_best", best["threads"], best["chunk_size"], runs_best)
        ts = ts + [best["seconds"]]
        return statistics.median(ts), ts, d, ph, {"threads": best["threads"], "chunk_size": best["chunk_size"]}, sweep

    def e2e(self, key, sp, cpu_cfg, note, opts=(), oracle_opts=None, runs=3, cpu_runs=1, extras=None, detached_runs=0, queue_runs=0, price=0.0, out_name="out_CpG.bedGraph"):
        """one end-to-end comparison: the oracle at cpu_cfg, this build `runs` times (in place); identical outputs asserted into the entry"""
        t_c, ts_c, d_c, ph = self.run_oracle(sp, key + "_cpu", cpu_cfg["threads"], cpu_cfg["chunk_size"], cpu_runs, opts=oracle_opts if oracle_opts is not None else opts)
        t_g, ts_g, d_g, ok_g, in_g = self.run_ours(sp, key + "_inplace", {}, runs=runs, opts=opts)
        calls = _calls_of(d_c, out_name); bam = os.path.getsize(str(sp) + ".bam")
        e = {"bam_bytes": bam, "calls": calls, "calls_counted_in": out_name, "cpu_all_cores_seconds": t_c, "cpu_runs": ts_c, "cpu_setting": cpu_cfg, "cpu_phases": ph, "cpu_serial_fraction": ph.get("_serial_fraction"),
             "seconds": t_g, "runs": ts_g, "value": calls / t_g, "unit": "CpG calls/s" if out_name.endswith("CpG.bedGraph") else "calls/s", "speedup_vs_cpu_all_cores": t_c / t_g, "identical_to_oracle": bool(ok_g and self.same(d_g, d_c)),
             "inside_process_runs": in_g, "bam_GBps": bam / t_g / 1e9, "options": list(opts), "note": note,
             "protocol": f"CPU: {cpu_runs} run(s) at the setting named, MDK_ORACLE_PROFILE phases of the last; this build: {runs} runs, median; the caller's wall clock around the command, teardown in place, each run a second after the previous command's last process has gone"}
        if detached_runs:
            t_d, ts_d, _, ok_d, _ = self.run_ours(sp, key + "_detached", {"MDK_DETACH": "1"}, runs=detached_runs, opts=opts)
            e["detached"] = {"seconds": t_d, "runs": ts_d, "speedup_vs_cpu_all_cores": t_c / t_d, "ok": bool(ok_d), "note": "MDK_DETACH=1 (opt-in): the work is done by a child, the command returns when the child reports its outputs closed"}
        if queue_runs:
            t_q, ts_q, _, ok_q, _ = self.run_ours(sp, key + "_queue", {}, runs=queue_runs, gap=0.0, opts=opts)
            e["queue"] = {"seconds_per_sample": t_q, "runs": ts_q, "speedup_vs_cpu_all_cores": t_c / t_q, "ok": bool(ok_q), "note": "runs back to back with no pause, teardown in place: what a queue of samples gets per sample"}
        if extras:
            e.update(extras)
        self.result[key] = e
        return e, d_c


def e2e_legs(args, result, data, work, extra, headline, t_start, mdk):
    G = Legs(args, result, data, work, extra, t_start, mdk)
    cov = args.coverage
    # ---- cpu_baseline + e2e_cli on the 32 Mb sample ----
    sp = G.synth(G.data / f"cpu_sample_{args.cpu_sample_length}_{cov}", str(args.cpu_sample_length), cov, S1_SEED + 1000)
    t_single, ts_single, d_single, ph_single = G.run_oracle(sp, "single", 1, None, 1)
    t_all, ts_all, d_all, ph_all, best, sweep = G.cpu_setting_sweep(sp, "small", ((32, 250_000), (64, 50_000), (64, 250_000), (128, 1_000_000), (G.ncores, max(50_000, args.cpu_sample_length // (4 * G.ncores)))), 2)
    calls = _calls_of(d_single)
    result["cpu_baseline"] = {"value": calls / t_all, "unit": "CpG calls/s", "cores": best["threads"], "kind": "port",
                              "sample": f"oracle/mdk_oracle extract -@ {best['threads']} --chunkSize {best['chunk_size']} -- the fastest of a sweep over worker threads x chunk size on this box's {G.ncores} hardware threads "
                                        f"(C restatement of the reference with its chunk-parallel worker threads, extract.c:325-350,1479-1486; end to end from the BAM file: the file mapped, inflate with CRC32 check by all threads, pileup, text) "
                                        f"on a {args.cpu_sample_length} bp / {cov}x sample of the same synthetic workload; median of 3 runs {t_all:.2f} s, {calls} CpG calls; the reference binary itself cannot be built here (no htslib)",
                              "seconds": t_all, "runs": ts_all, "cpg_calls": calls, "identical_to_single_thread": G.same(d_all, d_single), "host_threads": G.ncores, "sweep": sweep,
                              "phases": ph_all, "serial_fraction": ph_all.get("_serial_fraction"),
                              "single_thread": {"value": calls / t_single, "seconds": t_single, "runs": ts_single, "cores": 1, "phases": ph_single}}
    t_g, ts_g, d_g, ok_g, in_g = G.run_ours(sp, "inplace", {}, runs=5)      # (five: a run this short ends in one of the exit's two modes -- 3 ms or ~0.3 s, DESIGN.md 8.5 -- and three runs make a coin toss of the median)
    t_gd, ts_gd, _, ok_gd, _ = G.run_ours(sp, "detached", {"MDK_DETACH": "1"}, runs=2)
    result["e2e_cli"] = {"seconds": t_g, "runs": ts_g, "value": calls / t_g, "unit": "CpG calls/s", "threads": int(G.threads), "protocol": "5 runs, median, whole-process wall clock with the teardown in place (the CPU baseline's protocol)",
                         "speedup_vs_cpu_baseline": t_all / t_g, "speedup_vs_single_thread": t_single / t_g, "identical_to_oracle": bool(ok_g and ok_gd and G.same(d_g, d_single)),
                         "inside_process_runs": in_g, "bam_bytes": os.path.getsize(str(sp) + ".bam"),
                         "detached": {"seconds": t_gd, "runs": ts_gd, "speedup_vs_cpu_baseline": t_all / t_gd, "note": "MDK_DETACH=1 (opt-in): the command's work is done by a child, the command returns when the child reports its outputs closed"},
                         "note": "`MethylDackel extract` of this build on the same file, one process (start-up, HIP init, inflate on the host's threads and -- once the device is up -- on the device, chunk preparation, kernels, D2H, text, teardown)"}
    # ---- the device inflate on the record ----
    try:
        pb = subprocess.run([str(REPO / "tools/_build/piece_bench"), str(sp) + ".bam", "96", "8", "0"], capture_output=True, text=True, timeout=300)
        pj = json.loads(pb.stdout); kv = pj["kernel_only"]["v0"]; inf_bytes = kv["comp_bytes"] + kv["out_bytes"]
        result["inflate"] = {"workload": f"the {args.cpu_sample_length} bp sample's BAM: {pj['members']} BGZF members, {pj['file_MB']:.0f} MB -> {pj['inflated_MB']:.0f} MB",
                             "pipelined": {"GBps_compressed": pj["pass1"]["GBps_compressed"], "GBps_inflated": pj["pass1"]["GBps_inflated"], "seconds": pj["pass1"]["seconds"],
                                           "note": "whole file through md_piece_submit / md_piece_wait: staging copy, H2D of the compressed bytes, k_inflate, k_crc32, k_walk, digests back; 96 MB pieces, 8 in flight (what the command keeps)"},
                             "GBps_compressed": kv["GBps_compressed"], "GBps_inflated": kv["GBps_inflated"],
                             "roofline": {"kernel": "k_inflate", "bound": "hbm", "kernel_ms": kv["inflate_ms"], "algo_bytes_per_launch": inf_bytes, "achieved": inf_bytes / (kv["inflate_ms"] * 1e6), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": inf_bytes / (kv["inflate_ms"] * 1e6) / HBM_PEAK_GBS, "members_per_launch": pj["kernel_only"]["piece_members"],
                                          "note": "algorithmic bytes = compressed bytes read + inflated bytes written, one 96 MB piece per launch; the kernel is bound by instruction issue and LDS round trips of its chains and pointer jumps, not by HBM (DESIGN.md 4)"},
                             "crc32_ms": kv["crc32_ms"], "walk_ms": kv["walk_ms"], "crc32_GBps": kv["out_bytes"] / (kv["crc32_ms"] * 1e6) if kv["crc32_ms"] > 0 else None}
        pw = subprocess.run([str(REPO / "tools/_build/piece_bench"), str(sp) + ".bam", "4000", "1", "0"], capture_output=True, text=True, timeout=300)
        kw = json.loads(pw.stdout)["kernel_only"]["v0"]
        result["inflate"]["device_full"] = {"members_per_launch": json.loads(pw.stdout)["kernel_only"]["piece_members"], "kernel_ms": kw["inflate_ms"], "GBps_compressed": kw["GBps_compressed"], "GBps_inflated": kw["GBps_inflated"],
                                            "crc32_ms": kw["crc32_ms"], "walk_ms": kw["walk_ms"], "identical_to_zlib": kw.get("identical_to_zlib"),
                                            "note": "k_inflate over all members of the file in one launch; HIP events"}
    except Exception as ex:
        result["inflate"] = {"error": repr(ex)[:300]}
    if not (args.large_sample_length and headline):
        return
    # ---- 128 Mb: the CPU's best setting at this size, the command in place / detached / as a queue ----
    spl = G.synth(G.data / f"cpu_sample_{args.large_sample_length}_{cov}", str(args.large_sample_length), cov, S1_SEED + 1000)
    alt = (64, 250_000) if best["threads"] == 32 else (32, 250_000)
    t_la, ts_la, d_la, ph_la, cfg_l, sweep_l = G.cpu_setting_sweep(spl, "large", ((best["threads"], best["chunk_size"]), alt, (64, 1_000_000)), 1)
    t_lg, ts_lg, d_lg, ok_lg, in_lg = G.run_ours(spl, "large_inplace", {}, runs=5)
    t_ld, ts_ld, _, ok_ld, _ = G.run_ours(spl, "large_detached", {"MDK_DETACH": "1"}, runs=2)
    t_lq, ts_lq, _, ok_lq, _ = G.run_ours(spl, "large_queue", {}, runs=3, gap=0.0)
    calls_l = _calls_of(d_la); bam_l = os.path.getsize(str(spl) + ".bam")
    result["e2e_large"] = {"sample_bp": args.large_sample_length, "bam_bytes": bam_l, "cpg_calls": calls_l, "cpu_all_cores_seconds": t_la, "cpu_runs": ts_la, "cpu_setting": cfg_l, "cpu_sweep": sweep_l, "cpu_phases": ph_la,
                           "cpu_serial_fraction": ph_la.get("_serial_fraction"), "seconds": t_lg, "runs": ts_lg, "value": calls_l / t_lg, "unit": "CpG calls/s", "speedup_vs_cpu_all_cores": t_la / t_lg,
                           "identical_to_oracle": bool(ok_lg and ok_ld and ok_lq and G.same(d_lg, d_la)), "inside_process_runs": in_lg, "bam_GBps": bam_l / t_lg / 1e9,
                           "detached": {"seconds": t_ld, "runs": ts_ld, "speedup_vs_cpu_all_cores": t_la / t_ld},
                           "queue": {"seconds_per_sample": t_lq, "runs": ts_lq, "speedup_vs_cpu_all_cores": t_la / t_lq, "note": "three runs back to back with no pause between them, teardown in place"},
                           "protocol": "CPU: a sweep of three settings at this size, the fastest once more, median; this build: 5 runs, median; the caller's wall clock around the command with the teardown in place (the default)"}
    # ---- BASELINE configs[2]: --CHG --CHH with --OT/--OB trimming, end to end (20x the output lines through the host's formatter) ----
    if G.fits("e2e_cfg3", 45):
        o3 = ["--CHG", "--CHH", "--OT", "6,146,6,146", "--OB", "6,146,6,146"]
        G.e2e("e2e_cfg3", spl, cfg_l, "BASELINE.json configs[2] at 128 Mb: all three contexts with mbias trimming; calls = CpG calls (the CHG / CHH files are compared byte for byte as well)", opts=o3, runs=3)
    # ---- 512 Mb: K copies of the large sample as K contigs ----
    spx = None
    if args.xl_copies > 1 and G.fits("e2e_xl", 75):
        spx = G.data / f"xl_{args.large_sample_length}x{args.xl_copies}_{cov}"
        if not Path(str(spx) + ".bam").exists():
            t1 = time.time()
            subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(spl), str(spx), str(args.xl_copies)], check=True, capture_output=True, timeout=600)
            log(f"[bench] xl sample written in {time.time() - t1:.1f} s")
        alt_x = (64, 250_000) if cfg_l["threads"] == 32 else (32, 250_000)
        t_x2, _, _, _ = G.run_oracle(spx, "xl_alt", alt_x[0], alt_x[1], 1)
        e, d_xc = G.e2e("e2e_xl", spx, cfg_l, f"{args.xl_copies} copies of the 128 Mb sample as {args.xl_copies} contigs (tools/mdk_replicate)", runs=3, cpu_runs=2, detached_runs=2,
                     extras={"sample_bp": args.large_sample_length * args.xl_copies, "contigs": args.xl_copies})
        if t_x2 < e["cpu_all_cores_seconds"]:       # the other setting was the faster one at this size: the baseline is the faster
            e["cpu_other_setting"] = {"threads": alt_x[0], "chunk_size": alt_x[1], "seconds": t_x2, "note": "one run; faster than the setting above, and what the speed-up is computed from"}
            e["cpu_all_cores_seconds"] = t_x2; e["speedup_vs_cpu_all_cores"] = t_x2 / e["seconds"]
            if "detached" in e:
                e["detached"]["speedup_vs_cpu_all_cores"] = t_x2 / e["detached"]["seconds"]
        else:
            e["cpu_other_setting"] = {"threads": alt_x[0], "chunk_size": alt_x[1], "seconds": t_x2}
        if args.ranks_leg > 1 and G.fits("e2e_ranks", 20):
            try:
                ranks_on_sample(G, result, spx, args.ranks_leg, e, d_xc)
            except Exception as ex:
                result["e2e_ranks"] = {"error": repr(ex)[:300]}
    # ---- the metric's own size: a human-like 30x sample of 24 DISTINCT contigs (mdk_synth -j), as large as this box holds ----
    shm = Path("/dev/shm")
    if args.human_gb > 0 and G.fits("e2e_human", 60 + 55 * args.human_gb):
        scale = asked = args.human_gb / 3.1
        # what the leg holds at once: the BAM (17.1 MB per Mb), the generator's records (~3.5x that) while it runs, then the oracle's inflated copy + record table (~4.5x)
        need_gb = 17.1e-3 * 3100 * scale * 6.0
        room = min(_mem_available_gb(), shutil.disk_usage(shm).free / 1e9 if shm.is_dir() else 0.0)
        while scale > 0.3 and need_gb > 0.6 * room:
            scale *= 0.75; need_gb *= 0.75
        if need_gb <= 0.6 * room:
            hd = shm / f"mdk_bench_human_{os.getpid()}"; hd.mkdir(exist_ok=True)
            try:
                lens = [max(1_000_000, int(m * 1_000_000 * scale)) for m in HUMAN_LADDER]
                sph = G.synth(hd / "human", ",".join(str(x) for x in lens), cov, S1_SEED + 3000, par=min(96, G.ncores))
                G.e2e("e2e_human", sph, cfg_l, "24 distinct contigs with hg38's length ladder (tools/mdk_synth -j: every contig its own bases and reads), 30x, in RAM-backed storage; "
                      + ("the whole ladder: 3.1 Gb" if scale > 0.99 else f"scaled to {sum(lens) / 1e9:.2f} Gb" + (": what this box's memory holds next to the oracle's in-memory copy" if scale < asked else " (--human-gb)")),
                      runs=2, cpu_runs=1, extras={"sample_bp": sum(lens), "contigs": len(lens), "storage": str(hd)})
            finally:
                shutil.rmtree(hd, ignore_errors=True)
        else:
            G.skipped.append({"leg": "e2e_human", "why": f"needs ~{need_gb:.0f} GB of memory, {room:.0f} GB available"})
    # ---- BASELINE configs[4]: 100x with --mergeContext and a bigWig mappability filter ----
    if args.cfg5_mb > 0 and G.fits("e2e_cfg5", 35 + 0.9 * args.cfg5_mb):
        per = max(1, args.cfg5_mb // 4) * 1_000_000
        sp5 = G.synth(G.data / f"cfg5_{args.cfg5_mb}Mb_100x", ",".join([str(per)] * 4), 100.0, S1_SEED + 5000, par=min(64, G.ncores), more=["--bw", "--bbm"])
        G.e2e("e2e_cfg5", sp5, cfg_l, f"BASELINE.json configs[4] at {4 * per // 1_000_000} Mb: 100x, --mergeContext, the mappability track read as bigWig (-M) by this build and as BBM (-B, the same values) by the oracle, which has no bigWig reader",
              opts=["--mergeContext", "-M", str(sp5) + ".bw"], oracle_opts=["--mergeContext", "-B", str(sp5) + ".bbm"], runs=3, extras={"sample_bp": 4 * per, "coverage": 100.0})
    if G.skipped:
        result["legs_skipped"] = G.skipped
    if G.slow_runs:
        result["slow_runs"] = G.slow_runs
    result["bench_seconds"] = round(time.time() - t_start, 1)


def ranks_on_sample(G, result, sp, n_ranks, one, ref_dir, devices=None):
    """`MethylDackel extract` as n_ranks processes (csrc/host/mdk_ranks.c) on a sample: chunks dealt k mod N, and claimed from rank 0's counter (MDK_CLAIM=1)"""
    out = {"ranks": n_ranks, "sample_bp": one.get("sample_bp"), "one_rank_seconds": one["seconds"]}
    for mode, env in (("dealt", {}), ("claimed", {"MDK_CLAIM": "1"})):
        t, ts, d, ok, _ = G.run_ours(sp, f"ranks{n_ranks}_{mode}", env, runs=2, ranks=n_ranks)
        same = G.same(d, ref_dir)
        out[mode] = {"seconds": t, "runs": ts, "ok": bool(ok), "identical_to_oracle": bool(same), "speedup_vs_one_rank": one["seconds"] / t, "value": one["calls"] / t}
    import torch
    shared = devices is None and torch.cuda.device_count() < n_ranks
    out["exchange"] = {"transport": "the ranks share one physical GPU: site buffers travel over the ranks' TCP connections (RCCL refuses two ranks on one device)" if shared
                       else "ncclSend/ncclRecv of the site buffers to rank 0 (csrc/mdk_comm.hip md_comm_result_send/recv), control over TCP"}
    result["e2e_ranks"] = out


def ranks_leg(args, result, data, work, world, mdk, dist):
    """--gpus N > 1: the thing that shards -- the command as N ranks, one per GPU, on the XL sample, next to the one-rank run"""
    G = Legs(args, result, data, work, [], time.time(), mdk)
    cov = args.coverage
    spl = G.synth(G.data / f"cpu_sample_{args.large_sample_length}_{cov}", str(args.large_sample_length), cov, S1_SEED + 1000)
    spx = G.data / f"xl_{args.large_sample_length}x{args.xl_copies}_{cov}"
    if not Path(str(spx) + ".bam").exists():
        subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(spl), str(spx), str(args.xl_copies)], check=True, capture_output=True, timeout=600)
    t_c, ts_c, d_c, ph = G.run_oracle(spx, "ranks_cpu", 32, 250_000, 1)
    t1, ts1, d1, ok1, _ = G.run_ours(spx, "ranks_one", {}, runs=2)
    calls = _calls_of(d_c)
    one = {"seconds": t1, "calls": calls, "sample_bp": args.large_sample_length * args.xl_copies}
    ranks_on_sample(G, result, spx, world, one, d_c)
    result["e2e_ranks"].update({"cpu_all_cores_seconds": t_c, "one_rank_identical_to_oracle": bool(ok1 and G.same(d1, d_c)),
                                "value_unit": "CpG calls/s of the whole job (the command's wall clock, teardown in place)"})



def main():
    t_start = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--resident", type=int, default=16, help="R: resident 1 Mb intervals per rank (R x ~56 MB must exceed the 256 MiB Infinity Cache)")
    ap.add_argument("--passes", type=int, default=96, help="P: a step presents the R resident intervals P times (P x R chunks)")
    ap.add_argument("--length", type=int, default=1_000_000, help="interval (chunk) length; S1 = 1 Mb")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--extra", default="", help="extra extract options, e.g. '--CHG --CHH' (not the headline config)")
    ap.add_argument("--synth-args", default="", help="extra mdk_synth options, e.g. '--clean' (not the headline config)")
    ap.add_argument("--cpu-sample-length", type=int, default=32_000_000, help="bp of the same synthetic workload the CPU oracle is timed on")
    ap.add_argument("--large-sample-length", type=int, default=128_000_000, help="bp of the second, larger end-to-end sample (0 = skip)")
    ap.add_argument("--xl-copies", type=int, default=4, help="the XL end-to-end sample = this many copies of the large sample as that many contigs (<= 1: skip)")
    ap.add_argument("--budget-s", type=float, default=420.0, help="seconds the whole run may take: end-to-end legs that no longer fit are left out (and named in legs_skipped)")
    ap.add_argument("--human-gb", type=float, default=0.0, help="Gb of the human-like 24-contig 30x sample (0: skip); scaled down to what the box's memory holds")
    ap.add_argument("--cfg5-mb", type=int, default=128, help="Mb of the 100x sample for BASELINE configs[4] (0: skip)")
    ap.add_argument("--ranks-leg", type=int, default=2, help="N = 1 runs: also run the command as this many ranks sharing the GPU (0/1: skip)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the rocprofv3 --pmc passes that measure the dominant family's HBM traffic for this line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-only", action="store_true", help="the step is the pileup alone over resident segments (round 2's loop; for profiling that kernel)")
    ap.add_argument("--no-exchange", action="store_true", help="N ranks without the gather of site buffers to rank 0 (what the run falls back to when no communicator can be made)")
    ap.add_argument("--devices", default="", help="comma list: physical device of each local rank (tests: two ranks on one GPU)")
    ap.add_argument("--data-dir", default="", help="keep the synthetic inputs here and reuse them on the next run (profiling passes)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"[bench] WORLD_SIZE={world} overrides --gpus {args.gpus}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    devmap = [int(x) for x in args.devices.split(",")] if args.devices else None
    dev_index = devmap[local_rank % len(devmap)] if devmap else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        # torch.distributed carries only bookkeeping (the exchange's bootstrap, barriers, the max over ranks) over gloo; the data path --
        # site buffers to rank 0 -- is libmdk_hip's own exchange (RCCL between devices), created below
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import methyldackel_amd as mdk
    if rank == 0:
        mdk.build()
    if world > 1:
        dist.barrier()
    L = mdk.lib_hip()

    R = max(2 * GROUP, (args.resident + GROUP - 1) // GROUP * GROUP)
    work = Path(tempfile.mkdtemp(prefix=f"mdk_bench_r{rank}_"))
    data = Path(args.data_dir) if args.data_dir else work
    data.mkdir(parents=True, exist_ok=True)
    prefix = data / f"S1_r{rank}_{R}x{args.length}_{args.coverage}"
    t0 = time.time()
    if not (Path(str(prefix) + ".json").exists() and Path(str(prefix) + ".bam").exists()):
        synth = subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(prefix), "-L", str(args.length * R), "-c", str(args.coverage),
                                "-s", str(S1_SEED + rank)] + args.synth_args.split(), capture_output=True, text=True, check=True)
        Path(str(prefix) + ".json").write_text(synth.stdout)
    synth_info = json.loads(Path(str(prefix) + ".json").read_text())
    log(f"[bench] rank {rank}: synthetic {R} x {args.length} bp at {args.coverage}x in {time.time() - t0:.1f} s")
    extra = args.extra.split()
    cmd = [str(prefix) + ".fa", str(prefix) + ".bam", "--chunkSize", str(args.length), "-@", str(min(32, os.cpu_count() or 1))] + extra + ["-o", str(work / "gpu")]

    # the R intervals: records uploaded as `extract` uploads them; they then stay resident in HBM, one per device slot
    t0 = time.time()
    plan = mdk.Plan(cmd)
    plan.set_prep(1)                         # as `MethylDackel extract` runs: the chunk's BAM records go to the device, which prepares them itself
    cfg = plan.dev_cfg()
    cfg.n_slots = R + 2                      # R resident intervals + two slots for the streamed figure
    dev = mdk.Device(cfg, device=dev_index)
    dev.set_prep(plan.prep_cfg())
    reads = segs = n_sites_sum = cpg_calls = all_calls = raw_bytes = raw_records = 0
    keep_batches = []                        # (host copies of two batches for the streamed figure)
    n_chunks = 0
    while n_chunks < R:
        chunk = plan.next_chunk()
        assert chunk is not None and not chunk.skipped, "the synthetic contig must give R full chunks"
        plan.ensure_reference(dev, chunk.tid)
        dev.upload_raw(n_chunks, chunk.raw)   # H2D of the records
        dev.launch(n_chunks)                  # preparation + pileup
        sites = dev.download(n_chunks)        # waits: the pipeline's host buffers may be recycled after this
        _, n_seg_dev, n_read_dev = dev.debug_segments(n_chunks)
        reads += n_read_dev; segs += n_seg_dev; n_sites_sum += sites.n_sites; raw_bytes += sum(chunk.raw.range[i].bytes for i in range(chunk.raw.n_ranges)); raw_records += chunk.raw.n_records
        if sites.n_sites:
            a = np.ctypeslib.as_array(C.cast(sites.site, C.POINTER(C.c_uint32)), shape=(int(sites.n_sites), 4))      # md_site = {pos, nmeth, nunmeth, meta}
            c = a[:, 1].astype(np.int64) + a[:, 2]
            all_calls += int(c.sum()); cpg_calls += int(c[((a[:, 3] >> 1) & 3) == 0].sum())
        if n_chunks < 2:
            b = chunk.raw
            cat = b"".join(C.string_at(b.range[i].ptr, b.range[i].bytes) for i in range(b.n_ranges))
            offs = np.asarray(mdk.raw_record_offsets(b), dtype=np.uint32).tobytes()
            keep_batches.append((b.tid, b.beg, b.end, b.n_records, b.woff, b.wlen, cat, offs))
        n_chunks += 1
    t_host = time.time() - t0
    log(f"[bench] rank {rank}: {R} intervals resident after {t_host:.1f} s")
    slots = list(range(R))
    slot_arr = (C.c_int * R)(*slots)

    comm = C.c_void_p(); exchange_off = None
    if world > 1:
        idbuf = torch.zeros(mdk.COMM_ID_BYTES, dtype=torch.uint8)
        have_id = torch.tensor([1], dtype=torch.int64)
        if rank == 0:
            raw = C.create_string_buffer(mdk.COMM_ID_BYTES)
            if L.md_comm_unique_id(raw) == 0:
                idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
            else:                                                     # librccl could not be loaded: the ranks run without the exchange (below)
                have_id[0] = 0; log(f"[bench] no RCCL id: {(L.md_dev_last_error() or b'?').decode(errors='replace')}")
        dist.broadcast(idbuf, src=0); dist.broadcast(have_id, src=0)
        # ranks that share a physical device cannot be RCCL peers: they exchange through an IPC mapping of rank 0's receive buffers
        phys = torch.tensor([dev_index], dtype=torch.int64); allphys = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allphys, phys)
        shared = len({int(x.item()) for x in allphys}) < world
        if args.no_exchange:
            cb = None; rc = -2
        elif shared:
            def oob_allgather(_ctx, send, recv, nbytes):
                t = torch.frombuffer(bytearray(C.string_at(send, nbytes)), dtype=torch.uint8).clone(); outs = [torch.zeros(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(outs, t)
                C.memmove(recv, b"".join(bytes(o.numpy().tobytes()) for o in outs), nbytes * world)
                return 0
            cb = mdk.md_comm_oob_fn(oob_allgather)
            rc = L.md_comm_open_rank_shared(dev.h, rank, world, cb, None, C.byref(comm))
        elif int(have_id.item()) == 0:
            cb = None; rc = -2
        else:
            cb = None
            rc = L.md_comm_open_rank(dev.h, rank, world, bytes(idbuf.numpy().tobytes()), C.byref(comm))
        # every rank must hold a communicator before anyone enters a collective on it: if one could not be made (no usable RCCL, a device
        # the library refuses), all ranks go on WITHOUT the exchange -- the interval-sharded work itself needs none -- and the line says so
        comm_err = "" if rc == 0 else "--no-exchange" if args.no_exchange else (L.md_dev_last_error() or b"?").decode(errors="replace")
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int64); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            log(f"[bench] rank {rank}: no exchange between the ranks ({comm_err or 'another rank could not create its communicator'})")
            if rc == 0:
                L.md_comm_close(comm)
            comm = C.c_void_p()
            exchange_off = comm_err or "a rank could not create its communicator"
        joined = torch.tensor([1], dtype=torch.int64); dist.all_reduce(joined, op=dist.ReduceOp.SUM)
        n_gpus = int(joined.item())                                   # ranks taking part
    else:
        n_gpus, shared = 1, False
    bench = C.c_void_p()
    use_comm = world > 1 and bool(comm.value)
    rc = L.md_bench_open(dev.h, comm if use_comm else None, slot_arr, R, GROUP, C.byref(bench))
    if world > 1:
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int64); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and use_comm:                          # (the size agreement over the communicator failed somewhere: once more without it)
            exchange_off = (L.md_dev_last_error() or b"?").decode(errors="replace") if rc else "md_bench_open failed on another rank"
            log(f"[bench] rank {rank}: no exchange between the ranks ({exchange_off})")
            if rc == 0:
                L.md_bench_close(bench)
            bench = C.c_void_p(); use_comm = False
            rc = L.md_bench_open(dev.h, None, slot_arr, R, GROUP, C.byref(bench))
    assert rc == 0, L.md_dev_last_error()
    assert L.md_bench_set_prep(bench, 0 if args.kernel_only else 1) == 0, L.md_dev_last_error()

    chunks_per_step = args.passes * R          # GROUP of them share one launch of each kernel
    res = mdk.md_bench_run_result()

    def run(k_steps):
        rc = L.md_bench_run(bench, k_steps * chunks_per_step // GROUP, C.byref(res))
        assert rc == 0, L.md_dev_last_error()

    def fence():
        if world > 1:
            dist.barrier()
        dev.sync()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if args.steps:
        rc = L.md_bench_verify(bench)
        assert rc == 0, L.md_dev_last_error()
    exchanges, bytes_per_exchange = int(res.exchanges), int(res.bytes_per_exchange)
    log(f"[bench] rank {rank}: timed loop {dt:.2f} s")
    L.md_bench_close(bench)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([cpg_calls, all_calls, int(n_sites_sum)], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_cpg_calls, total_calls, total_sites = (int(x) for x in tot.tolist())
    else:
        total_cpg_calls, total_calls, total_sites = cpg_calls, all_calls, int(n_sites_sum)

    # kernel-level timing with HIP events on the launch stream, inside the library, rotating over the R resident intervals
    br = dev.bench_rotate(slots, 8, 200, per_launch=GROUP)
    br_single = dev.bench_rotate(slots, 2 * R, max(200, 50 * R))
    prep_ms = C.c_float(0); prep1_ms = C.c_float(0)
    assert L.md_dev_bench_prep_rotate(dev.h, slot_arr, R, GROUP, 2 * (R // GROUP), 40 * (R // GROUP), C.byref(prep_ms)) == 0, L.md_dev_last_error()
    assert L.md_dev_bench_prep_rotate(dev.h, slot_arr, R, 1, R, 10 * R, C.byref(prep1_ms)) == 0, L.md_dev_last_error()
    pile_s, prep_s = br.ms_pileup / 1e3, prep_ms.value / 1e3
    # algorithmic bytes per LAUNCH (GROUP chunks).  Pileup: SURVEY.md 8d (reads' payload + reference + sites).  Preparation: its input -- every
    # byte of the chunk's records, once -- + the 32-byte segments it writes; its intermediates (48 B per record, the name table) are overhead,
    # not algorithmic bytes.  The numerator is the one rounds 3-5 used, so the fractions compare; since round 6 the scan fetches
    # only the lines its fields lie in (1.7 of a record's 2.2: `traffic` below is what the counters saw), and `step` prices preparation +
    # pileup with SURVEY.md 8d's own per-chunk figure.
    pile_bytes = int(br.algo_bytes)
    prep_bytes = int((raw_bytes / R + 32.0 * segs / R) * GROUP)
    fam = {
        "pileup": {"kernel": "k_pileup_multi<false,false>", "bound": "hbm", "kernel_ms": br.ms_pileup, "kernel_ms_per_chunk": br.ms_pileup / GROUP, "algo_bytes_per_launch": pile_bytes,
                   "achieved": pile_bytes / pile_s / 1e9 if pile_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s"},
        "preparation": {"kernel": "k_prep_zero + k_prep_scan + k_prep_segs", "bound": "hbm", "kernel_ms": prep_ms.value, "kernel_ms_per_chunk": prep_ms.value / GROUP, "algo_bytes_per_launch": prep_bytes,
                        "achieved": prep_bytes / prep_s / 1e9 if prep_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "one_chunk_per_launch_ms": prep1_ms.value},
    }
    for f in fam.values():
        f["frac"] = f["achieved"] / HBM_PEAK_GBS
    dominant = max(fam, key=lambda k: fam[k]["kernel_ms"])
    step_bytes = pile_bytes           # SURVEY.md 8d's per-unit figure is the pileup's: the step moves it once per chunk
    step_s = pile_s + prep_s
    br1 = dev.bench(0, 5, 200)                 # one interval relaunched on cache-resident data, for comparison with round 1

    # streamed: the same chunk with its H2D upload (registered staging) and the D2H of its sites, two slots, chunk k+1 uploaded
    # and launched while chunk k is downloaded
    streamed = None
    if world == 1 and len(keep_batches) == 2:
        pinned, batches, keep = [], [], []
        L.md_host_alloc.restype = C.c_void_p
        for (tid, beg, end, n_rec, woff, wlen, cat, offs) in keep_batches:
            pc = L.md_host_alloc(C.c_uint64(len(cat))); po = L.md_host_alloc(C.c_uint64(max(len(offs), 8 << 20)))
            C.memmove(pc, cat, len(cat)); C.memmove(po, offs, len(offs))
            pinned += [pc, po]
            rg = (mdk.md_raw_range * 1)(); rg[0].ptr = C.cast(pc, C.POINTER(C.c_uint8)); rg[0].bytes = len(cat)
            b = mdk.md_raw_batch(); b.tid = tid; b.beg = beg; b.end = end; b.n_ranges = 1; b.range = rg; b.n_records = n_rec
            b.rec_off = C.cast(po, C.POINTER(C.c_uint32)); b.woff = woff; b.wlen = wlen
            batches.append(b); keep.append(rg)
        n_stream = 200
        for timed in (False, True):
            nn = n_stream if timed else 10
            ts = time.perf_counter()
            dev.submit_raw(R, batches[0])
            for k in range(1, nn):
                dev.submit_raw(R + (k & 1), batches[k & 1])
                dev.download(R + ((k - 1) & 1))
            dev.download(R + ((nn - 1) & 1))
            t_stream = time.perf_counter() - ts
        h2d = (len(keep_batches[0][6]) + len(keep_batches[0][7]) + len(keep_batches[1][6]) + len(keep_batches[1][7])) / 2
        per_chunk_calls = cpg_calls / R
        streamed = {"ms_per_chunk": t_stream / n_stream * 1e3, "value": per_chunk_calls * n_stream / t_stream, "unit": "CpG calls/s", "h2d_bytes_per_chunk": int(h2d),
                    "h2d_GBps": h2d * n_stream / t_stream / 1e9,
                    "note": "per chunk: hipMemcpyAsync of the chunk's BAM records + record table from registered huge-page staging memory, the preparation kernels, k_pileup, "
                            "D2H of the site records; two slots (chunk k+1 is uploaded and launched while chunk k is downloaded), one chunk per launch"}
        for p in pinned:
            L.md_host_free(C.c_void_p(p))

    # the dense-context configuration (BASELINE.json configs[2]: --CHG --CHH with --OT/--OB trimming on the same reads) through its own
    # kernel (8 lanes of a wavefront per segment), same resident intervals, same rotation, same 8-chunk launches
    dense = None
    if world == 1 and not extra and not args.synth_args and args.length == 1_000_000:
        plan2 = mdk.Plan([str(prefix) + ".fa", str(prefix) + ".bam", "--chunkSize", str(args.length), "-@", str(min(32, os.cpu_count() or 1)), "--CHG", "--CHH",
                          "--OT", "6,146,6,146", "--OB", "6,146,6,146", "-o", str(work / "dense")])
        plan2.set_prep(1)
        cfg2 = plan2.dev_cfg(); cfg2.n_slots = R
        dev2 = mdk.Device(cfg2, device=dev_index); dev2.set_prep(plan2.prep_cfg())
        calls2 = 0
        for i in range(R):
            c2 = plan2.next_chunk(); plan2.ensure_reference(dev2, c2.tid); dev2.upload_raw(i, c2.raw); dev2.launch(i)
            st2 = dev2.download(i)
            if st2.n_sites:
                a2 = np.ctypeslib.as_array(C.cast(st2.site, C.POINTER(C.c_uint32)), shape=(int(st2.n_sites), 4))
                calls2 += int(a2[:, 1].sum(dtype=np.int64) + a2[:, 2].sum(dtype=np.int64))
        brd = dev2.bench_rotate(slots, 8, 100, per_launch=GROUP)
        dense = {"workload": "the same R resident intervals with --CHG --CHH --OT 6,146,6,146 --OB 6,146,6,146 (BASELINE.json configs[2])", "kernel": "k_pileup_multi<false,true> (8 lanes per segment)",
                 "tile": int(brd.tile), "kernel_ms": brd.ms_pileup, "kernel_ms_per_chunk": brd.ms_pileup / GROUP, "algo_bytes_per_launch": int(brd.algo_bytes),
                 "achieved": brd.algo_bytes / (brd.ms_pileup / 1e3) / 1e9 if brd.ms_pileup > 0 else 0.0, "unit": "GB/s",
                 "frac": brd.algo_bytes / (brd.ms_pileup / 1e3) / 1e9 / HBM_PEAK_GBS if brd.ms_pileup > 0 else 0.0,
                 "sites_per_interval": int(brd.n_sites) // GROUP, "calls_per_interval": calls2 // R,
                 "value": (calls2 / R) * GROUP / (brd.ms_pileup / 1e3) if brd.ms_pileup > 0 else 0.0, "value_unit": "cytosine calls/s (all contexts), pileup kernel only"}
        dev2.close(); plan2.close()

    # HBM traffic of the dominant family, per launch: measured for THIS line when rocprofv3 is on the box -- one pass per counter (FETCH_SIZE,
    # WRITE_SIZE: MI355X_MICROARCH.md's recipe) over tools/prep_bench.py, which launches the same kernels on the same 16 resident intervals, 8
    # chunks per launch; only the dispatches with each kernel's largest grid (the 8-chunk launches) are averaged.  Without rocprofv3 the
    # committed summary of a separate profiled run stands in, and the line says which.
    traffic, traffic_info = None, None
    fam_kernels = {"preparation": ["k_prep_zero", "k_prep_scan", "k_prep_segs"], "pileup": ["k_pileup_multi<false,false>"]}
    under_profiler = any(os.environ.get(k) for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH"))      # (this run is itself being profiled: no profiler inside a profiler)
    if rank == 0 and world == 1 and not args.no_live_traffic and not under_profiler and not extra and not args.synth_args and args.length == 1_000_000 and shutil.which("rocprofv3"):
        try:
            import csv, glob
            per = {}
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                pd = work / f"pmc_{ctr}"
                subprocess.run([shutil.which("rocprofv3"), "--kernel-trace", "--output-format", "csv", "--pmc", ctr, "-d", str(pd), "-o", "p", "--", sys.executable, str(REPO / "tools/prep_bench.py"), str(R)],
                               cwd="/tmp", env=dict(os.environ, PREP_BENCH_FAST="2", TMPDIR="/tmp"), capture_output=True, text=True, timeout=400)
                rows = {}
                for f in glob.glob(f"{pd}/**/*counter_collection.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r["Counter_Name"] != ctr:
                            continue
                        k = re.sub(r"\(.*\)\s*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].strip())).replace(", ", ",").replace(" ", "")
                        rows.setdefault(k, []).append((int(r["Grid_Size"]), float(r["Counter_Value"])))
                for k, v in rows.items():
                    gmax = max(g for g, _ in v); big = [x for g, x in v if g >= 0.98 * gmax]
                    per.setdefault(k, {})[ctr] = (sum(big) / len(big) * 1024.0, len(big))              # rocprofv3 reports KiB
            ks = fam_kernels[dominant]
            if all(k in per and "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k] for k in ks):
                fr = sum(per[k]["FETCH_SIZE"][0] for k in ks); wr = sum(per[k]["WRITE_SIZE"][0] for k in ks)
                traffic = 2 * fr + wr
                traffic_info = {"measured": "live: rocprofv3 --pmc passes run by this bench.py invocation (tools/prep_bench.py: the same kernels on the same resident intervals, 8 chunks per launch)",
                                "kernels": ks, "dispatches": {k: per[k]["FETCH_SIZE"][1] for k in ks}, "fetch_raw_counter_bytes": fr, "write_raw_counter_bytes": wr,
                                "per_kernel": {k: {"fetch_x2_plus_write": 2 * per[k]["FETCH_SIZE"][0] + per[k]["WRITE_SIZE"][0]} for k in ks},
                                "note": "per launch, summed over the family's kernels: FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests as 64) + WRITE_SIZE"}
        except Exception as ex:
            log(f"[bench] live traffic counters failed: {ex!r}")
    if traffic is None:
        try:
            # the round's final summary (rNNfin_*) if there is one, else the last one of the latest round by name
            cand = sorted((REPO / "profiles").glob("r[0-9][0-9]*_rocprofv3_pmc_summary.json"), key=lambda q: (q.name[:3], "fin" in q.name.split("_")[0], q.name))
            prof = json.load(open(cand[-1]))
            k = prof["families"].get(dominant)
            if k and not extra and not args.synth_args and args.length == 1_000_000 and k.get("chunks_per_launch") == GROUP:
                traffic = k["hbm_bytes_per_launch"]["fetch_x2_plus_write"]
                traffic_info = {"measured": "NOT this run: the committed summary of a separate profiled run", "file": cand[-1].name, "kernels": k["kernels"], "dispatches": k["dispatches"], "fetch_raw_counter_bytes": k["hbm_bytes_per_launch"]["fetch_raw"], "write_raw_counter_bytes": k["hbm_bytes_per_launch"]["write_raw"],
                                "note": "per launch, summed over the family's kernels: FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests as 64) + WRITE_SIZE"}
        except Exception:
            pass

    result = None
    if rank == 0:
        chunks = args.steps * chunks_per_step
        value = (total_cpg_calls / R) * chunks / dt if dt > 0 else 0.0        # total_cpg_calls = one pass over every rank's R intervals
        headline = not extra and not args.synth_args and args.length == 1_000_000
        result = {
            "metric": "CpG calls/sec, synthetic 1 Mb contig 30x paired-end WGBS BAM, CpG extract",
            "value": value, "unit": "CpG calls/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3 if args.steps else 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32", "data": "synthetic",
            "config": {"layout": "as `MethylDackel extract` has them after the upload: the chunks' BAM records resident in HBM; every step prepares them again (admission, strand, compaction, "
                                 "pairing, CIGAR -> segments) and piles them up" if not args.kernel_only else "--kernel-only: resident segments, the pileup alone",
                       "workload": ("S1: synthetic 30x PE 2x150 WGBS, CpG-only extract in 1 Mb chunks (BASELINE.json configs[1])" if headline
                                    else f"synthetic {args.length} bp chunks, {args.coverage}x, extract {' '.join(extra)}") +
                                   f"; per GPU {R} different resident 1 Mb intervals (~{raw_bytes / 1e6:.0f} MB of records, beyond the 256 MiB Infinity Cache)",
                       "step": f"one pass over a batch of {args.passes} x {R} = {chunks_per_step} chunks per GPU (the {R} resident intervals in rotation): preparation + pileup per chunk, {GROUP} chunks per launch of each kernel",
                       "chunks_per_step_per_gpu": chunks_per_step, "launches_per_step_per_gpu": chunks_per_step // GROUP, "ms_per_chunk": dt / chunks * 1e3 if chunks else 0.0,
                       "interval_bp": args.length, "coverage": args.coverage, "resident_intervals_per_gpu": R,
                       "records_per_interval": raw_records // R, "record_bytes_per_interval": raw_bytes // R,
                       "reads_admitted_per_interval": reads // R, "segments_per_interval": segs // R, "records_per_gpu": synth_info["records"],
                       "sites_per_interval": int(n_sites_sum) // R, "cpg_calls_per_interval": int(cpg_calls) // R,
                       "tile": int(br.tile), "tiles_per_launch": int(br.n_tiles), "lds_bytes_per_workgroup": int(br.lds_bytes),
                       "parallelism": f"interval-sharded x{n_gpus}" + (f" + gather of site buffers to rank 0 ({GROUP} chunks = one launch per exchange, {bytes_per_exchange} B per exchange and rank)" if world > 1 else ""),
                       "in_flight": "2 launches queued per GPU, in order on one stream (launch g is issued, then the chunks of launch g-1 are collected: their site counts read back with one copy); every launch is issued and collected inside the timed region"},
            "roofline": dict(fam[dominant], family=dominant, traffic=traffic, traffic_source=traffic_info, chunks_per_launch=GROUP,
                             measured=f"HIP events around launches of the family's kernels ({GROUP} resident 1 Mb chunks per launch) rotating over the {R} resident intervals on one stream",
                             kernels=fam,
                             step={"algo_bytes_per_chunk": step_bytes // GROUP, "device_ms_per_chunk": step_s / GROUP * 1e3, "achieved": step_bytes / step_s / 1e9 if step_s > 0 else 0.0, "unit": "GB/s",
                                   "frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS if step_s > 0 else 0.0,
                                   "note": "SURVEY.md 8d's algorithmic bytes of a chunk over the device time of its preparation + pileup (sum of the families' HIP-event times)"},
                             pileup_one_chunk_per_launch={"kernel_ms": br_single.ms_pileup, "achieved": br_single.algo_bytes / (br_single.ms_pileup / 1e3) / 1e9 if br_single.ms_pileup > 0 else 0.0},
                             pileup_cache_resident={"kernel_ms": br1.ms_pileup, "achieved": br1.algo_bytes / (br1.ms_pileup / 1e3) / 1e9 if br1.ms_pileup > 0 else 0.0,
                                                    "note": "interval 0 relaunched back to back: its records stay in the 256 MiB Infinity Cache (the round-1 measurement)"}),
            "host_upload_s": t_host,
        }
        if world > 1:
            result["exchange"] = {"exchanges": exchanges, "bytes_per_exchange_per_rank": bytes_per_exchange,
                                  "transport": f"NONE -- the ranks ran without the gather of site buffers to rank 0: {exchange_off}" if exchange_off else
                                               "device copies into an IPC mapping of rank 0's buffers (ranks sharing one physical GPU)" if shared else "ncclSend/ncclRecv group (libmdk_hip md_comm_gather)"}
        if streamed:
            result["streamed"] = streamed
        if dense:
            result["dense_contexts"] = dense
        log(f"[bench] device legs done in {time.time() - t_start:.0f} s")
        if not args.no_cpu_baseline and world == 1:
          try:
            e2e_legs(args, result, data, work, extra, headline, t_start, mdk)
          except Exception as ex:            # a leg that fails or hangs (timeout) must not take the measured line with it
            result["legs_error"] = repr(ex)[:500]
            log(f"[bench] a CPU/end-to-end leg failed: {ex!r}")
        if world > 1 and not args.no_cpu_baseline:
          try:
            ranks_leg(args, result, data, work, world, mdk, dist)
          except Exception as ex:
            result["e2e_ranks"] = {"error": repr(ex)[:300]}
        print(json.dumps(result), flush=True)
    if world > 1:
        L.md_comm_close(comm)
    dev.close()
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

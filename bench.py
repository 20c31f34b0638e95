#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X `MethylDackel extract` hot path.

Metric (BASELINE.json): CpG calls/s, synthetic 30x paired-end WGBS, CpG-only extract, in 1 Mb chunks (configs[1], "S1").

Workload.  Every rank holds R (default 16) DIFFERENT S1-sized intervals resident in HBM: the R chunks the reference's
schedule (1 Mb chunks, extract.c:325-350) cuts out of an R Mb synthetic contig, each about 52 MB of admitted reads, about
0.8 GB together -- beyond the 256 MiB Infinity Cache, so every launch streams its inputs from HBM.  One STEP is one pass of
the hot path over a batch of P x R chunks (default P = 384: 6144 chunks in 768 kernel launches), i.e. the R resident intervals presented P
times in rotation.  Inside a step every launch is issued and collected (site counts read back) with two launches queued
on one in-order stream (launch g is issued, then launch g-1 is collected); with N ranks the kernels write into send buffers and the results of 8
consecutive launches travel to rank 0 with one ncclSend/ncclRecv exchange (libmdk_hip's md_comm, RCCL over xGMI) while the
next launch is computed.  A kernel launch covers 8 resident chunks (md_dev_launch_group): one 1 Mb chunk is only 489
workgroups, fewer than two per CU.  The loop is libmdk_hip's md_bench_run (C); Python only brackets it.

Also on the JSON line:
  roofline     -- k_pileup: algorithmic bytes per launch (SURVEY.md 8d formula, averaged over the R intervals) / HIP-event
                  time per launch while rotating over the R intervals on one stream, vs 8 TB/s
  cpu_baseline -- the CPU oracle (`oracle/`, "port") end to end on a 32 Mb sample of the same workload with all host
                  cores (`-@ nproc`: chunk-parallel workers as the reference's, extract.c:1479-1486), and with one thread
  streamed     -- the same chunks with H2D upload + kernel + D2H of the sites per chunk (pinned staging, two slots)
  e2e_cli      -- `MethylDackel extract` of this build on the CPU sample's BAM, whole-process wall clock
"""
import argparse
import ctypes as C
import numpy as np
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
S1_SEED = 0x5EED0001
GROUP = 8                    # resident chunks per kernel launch (md_dev_launch_group) = chunks per exchange


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--resident", type=int, default=16, help="R: resident 1 Mb intervals per rank (R x ~52 MB must exceed the 256 MiB Infinity Cache)")
    ap.add_argument("--passes", type=int, default=384, help="P: a step presents the R resident intervals P times (P x R chunk launches)")
    ap.add_argument("--length", type=int, default=1_000_000, help="interval (chunk) length; S1 = 1 Mb")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--extra", default="", help="extra extract options, e.g. '--CHG --CHH' (not the headline config)")
    ap.add_argument("--synth-args", default="", help="extra mdk_synth options, e.g. '--clean' (not the headline config)")
    ap.add_argument("--cpu-sample-length", type=int, default=32_000_000, help="bp of the same synthetic workload the CPU oracle is timed on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--data-dir", default="", help="keep the synthetic inputs here and reuse them on the next run (profiling passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"[bench] WORLD_SIZE={world} overrides --gpus {args.gpus}")
    n_gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        # torch.distributed carries only bookkeeping (the RCCL id, barriers, the max over ranks) over gloo; the data path --
        # site buffers to rank 0 -- is libmdk_hip's own RCCL communicator, created below
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

    # host side: decode + admit + pack the R intervals once; they then stay resident in HBM, one per device slot
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
        dev.upload_raw(n_chunks, chunk.raw)   # H2D of the records + admission, strand, name pairing, CIGAR expansion on the device
        dev.launch(n_chunks)
        sites = dev.download(n_chunks)        # waits: the pipeline's host buffers may be recycled after this
        _, n_seg_dev, n_read_dev = dev.debug_segments(n_chunks)
        reads += n_read_dev; segs += n_seg_dev; n_sites_sum += sites.n_sites; raw_bytes += sum(chunk.raw.range[i].bytes for i in range(chunk.raw.n_ranges)); raw_records += chunk.raw.n_records
        for i in range(sites.n_sites):
            r = sites.site[i]
            c = r.nmeth + r.nunmeth
            all_calls += c
            if ((r.meta >> 1) & 3) == 0:
                cpg_calls += c
        if n_chunks < 2:
            b = chunk.raw
            cat = b"".join(C.string_at(b.range[i].ptr, b.range[i].bytes) for i in range(b.n_ranges))
            offs = C.string_at(b.rec_off, 4 * b.n_records)
            keep_batches.append((b.tid, b.beg, b.end, b.n_records, b.woff, b.wlen, cat, offs))
        n_chunks += 1
    t_host = time.time() - t0
    slots = list(range(R))
    slot_arr = (C.c_int * R)(*slots)

    comm = C.c_void_p()
    if world > 1:
        idbuf = torch.zeros(mdk.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            raw = C.create_string_buffer(mdk.COMM_ID_BYTES)
            rc = L.md_comm_unique_id(raw)
            assert rc == 0, L.md_dev_last_error()
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        dist.broadcast(idbuf, src=0)
        rc = L.md_comm_open_rank(dev.h, rank, world, bytes(idbuf.numpy().tobytes()), C.byref(comm))
        assert rc == 0, L.md_dev_last_error()
    bench = C.c_void_p()
    rc = L.md_bench_open(dev.h, comm if world > 1 else None, slot_arr, R, GROUP, C.byref(bench))
    assert rc == 0, L.md_dev_last_error()

    launches_per_step = args.passes * R        # chunk passes per step; GROUP of them share one kernel launch
    res = mdk.md_bench_run_result()

    def run(k_steps):
        rc = L.md_bench_run(bench, k_steps * launches_per_step // GROUP, C.byref(res))
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
    pile_s = br.ms_pileup / 1e3
    achieved = br.algo_bytes / pile_s / 1e9 if pile_s > 0 else 0.0
    br1 = dev.bench(0, 5, 200)                 # one interval relaunched on cache-resident data, for comparison with round 1

    # streamed: the same chunk with its H2D upload (pinned staging) and the D2H of its sites, two slots, chunk k+1 uploaded
    # and launched while chunk k is downloaded
    streamed = None
    if world == 1 and len(keep_batches) == 2:
        pinned, batches, keep = [], [], []
        L.md_host_alloc.restype = C.c_void_p
        for (tid, beg, end, n_rec, woff, wlen, cat, offs) in keep_batches:
            pc = L.md_host_alloc(C.c_uint64(len(cat))); po = L.md_host_alloc(C.c_uint64(len(offs)))
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
                    "note": "per chunk: hipMemcpyAsync of the chunk's BAM records + record table from pinned host memory, the preparation kernels, k_pileup, D2H of the site records; "
                            "two slots (chunk k+1 is uploaded and launched while chunk k is downloaded)"}
        for p in pinned:
            L.md_host_free(C.c_void_p(p))
    prep_ms = C.c_float(0)
    rc = L.md_dev_bench_prep(dev.h, 0, 3, 30, C.byref(prep_ms))
    assert rc == 0, L.md_dev_last_error()

    # the dense-context configuration (BASELINE.json configs[2]: --CHG --CHH on the same reads) through its own kernel
    # (8 lanes of a wavefront per segment), same resident intervals, same rotation, same 8-chunk launches
    dense = None
    if world == 1 and not extra and not args.synth_args and args.length == 1_000_000:
        plan2 = mdk.Plan([str(prefix) + ".fa", str(prefix) + ".bam", "--chunkSize", str(args.length), "-@", str(min(32, os.cpu_count() or 1)), "--CHG", "--CHH", "-o", str(work / "dense")])
        plan2.set_prep(1)
        cfg2 = plan2.dev_cfg(); cfg2.n_slots = R
        dev2 = mdk.Device(cfg2, device=dev_index); dev2.set_prep(plan2.prep_cfg())
        calls2 = 0
        for i in range(R):
            c2 = plan2.next_chunk(); plan2.ensure_reference(dev2, c2.tid); dev2.upload_raw(i, c2.raw); dev2.launch(i)
            st2 = dev2.download(i)
            if st2.n_sites:
                a2 = np.ctypeslib.as_array(C.cast(st2.site, C.POINTER(C.c_uint32)), shape=(int(st2.n_sites), 4))      # md_site = {pos, nmeth, nunmeth, meta}
                calls2 += int(a2[:, 1].sum(dtype=np.int64) + a2[:, 2].sum(dtype=np.int64))
        brd = dev2.bench_rotate(slots, 8, 100, per_launch=GROUP)
        dense = {"workload": "the same R resident intervals with --CHG --CHH (BASELINE.json configs[2])", "kernel": "k_pileup_multi<.., QW> (8 lanes per segment)",
                 "tile": int(brd.tile), "kernel_ms": brd.ms_pileup, "kernel_ms_per_chunk": brd.ms_pileup / GROUP, "algo_bytes_per_launch": int(brd.algo_bytes),
                 "achieved": brd.algo_bytes / (brd.ms_pileup / 1e3) / 1e9 if brd.ms_pileup > 0 else 0.0, "unit": "GB/s",
                 "frac": brd.algo_bytes / (brd.ms_pileup / 1e3) / 1e9 / HBM_PEAK_GBS if brd.ms_pileup > 0 else 0.0,
                 "sites_per_interval": int(brd.n_sites) // GROUP, "calls_per_interval": calls2 // R,
                 "value": (calls2 / R) * GROUP / (brd.ms_pileup / 1e3) if brd.ms_pileup > 0 else 0.0, "value_unit": "cytosine calls/s (all contexts), kernel only"}
        dev2.close(); plan2.close()

    # HBM traffic of the kernel cannot be sampled from inside this process; it is taken from the committed rocprofv3 PMC summary
    # of this same command (profiles/, produced by tools/gpu_round.sh + tools/summarize_prof.py) with the calibration measured by
    # tools/mdk_calib (byte gathers of a known line count), or left null
    traffic, traffic_note = None, None
    try:
        prof = json.load(open(sorted((REPO / "profiles").glob("r02*_rocprofv3_pmc_summary.json"))[-1]))      # the latest committed summary of this round
        hb = prof["hbm_traffic_bytes_per_launch"]
        if not extra and not args.synth_args and args.length == 1_000_000:
            traffic = hb["fetch_calibrated"] + hb["write_calibrated"]
            traffic_note = hb["note"]
    except Exception:
        pass

    result = None
    if rank == 0:
        launches = args.steps * launches_per_step
        value = (total_cpg_calls / R) * launches / dt if dt > 0 else 0.0        # total_cpg_calls = one pass over every rank's R intervals
        headline = not extra and not args.synth_args and args.length == 1_000_000
        result = {
            "metric": "CpG calls/sec, synthetic 1 Mb contig 30x paired-end WGBS BAM, CpG extract",
            "value": value, "unit": "CpG calls/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3 if args.steps else 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32", "data": "synthetic",
            "config": {"layout": "as `MethylDackel extract` leaves it: BAM records resident, segments built by the device preparation, sequence/quality bytes read in place",
                       "workload": ("S1: synthetic 30x PE 2x150 WGBS, CpG-only extract in 1 Mb chunks (BASELINE.json configs[1])" if headline
                                    else f"synthetic {args.length} bp chunks, {args.coverage}x, extract {' '.join(extra)}") +
                                   f"; per GPU {R} different resident 1 Mb intervals (~{br.algo_bytes * R / 1e6:.0f} MB algorithmic, beyond the 256 MiB Infinity Cache)",
                       "step": f"one pass over a batch of {args.passes} x {R} = {launches_per_step} chunks per GPU (the {R} resident intervals in rotation), {GROUP} chunks per kernel launch",
                       "chunks_per_step_per_gpu": launches_per_step, "kernel_launches_per_step_per_gpu": launches_per_step // GROUP, "ms_per_chunk": dt / launches * 1e3 if launches else 0.0,
                       "interval_bp": args.length, "coverage": args.coverage, "resident_intervals_per_gpu": R,
                       "reads_admitted_per_interval": reads // R, "segments_per_interval": segs // R, "records_per_gpu": synth_info["records"],
                       "sites_per_interval": int(n_sites_sum) // R, "cpg_calls_per_interval": int(cpg_calls) // R,
                       "tile": int(br.tile), "tiles_per_launch": int(br.n_tiles), "lds_bytes_per_workgroup": int(br.lds_bytes),
                       "parallelism": f"interval-sharded x{n_gpus}" + (f" + RCCL gather of site buffers to rank 0 ({GROUP} chunks = one launch per exchange, {bytes_per_exchange} B per exchange and rank)" if world > 1 else ""),
                       "in_flight": "2 kernel launches queued per GPU, in order on one stream (launch g is issued, then the chunks of launch g-1 are collected: their site counts read back with one copy); every launch is issued and collected inside the timed region"},
            "roofline": {"bound": "hbm", "kernel": "k_pileup", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "algo_bytes_per_launch": int(br.algo_bytes), "kernel_ms": br.ms_pileup,
                         "chunks_per_launch": GROUP, "kernel_ms_per_chunk": br.ms_pileup / GROUP,
                         "measured": f"HIP events around 200 launches of k_pileup_multi ({GROUP} resident 1 Mb chunks per launch, {int(br.n_tiles)} workgroups) rotating over the {R} resident intervals on one stream",
                         "one_chunk_per_launch": {"kernel_ms": br_single.ms_pileup, "achieved": br_single.algo_bytes / (br_single.ms_pileup / 1e3) / 1e9 if br_single.ms_pileup > 0 else 0.0,
                                                  "note": "k_pileup over ONE 1 Mb chunk (489 workgroups, fewer than two per CU), rotating over the resident intervals: what round 1 launched"},
                         "cache_resident_comparison": {"kernel_ms": br1.ms_pileup, "achieved": br1.algo_bytes / (br1.ms_pileup / 1e3) / 1e9 if br1.ms_pileup > 0 else 0.0,
                                                       "note": "interval 0 relaunched back to back: its ~52 MB stay in the 256 MiB Infinity Cache (the round-1 measurement)"}},
            "host_prep_s": t_host,
            "device_prep": {"ms_per_chunk": prep_ms.value, "records_per_chunk": raw_records // R, "record_bytes_per_chunk": raw_bytes // R,
                            "achieved_GBps": (raw_bytes / R) / (prep_ms.value / 1e3) / 1e9 if prep_ms.value > 0 else 0.0,
                            "note": "per chunk, before the pileup: k_rec_scan (fields, CIGAR length, NH/XG aux walk, admission, strand), k_compact + name table, k_pair (overlap pairing with "
                                    "buffer eviction), k_seg_count/k_seg_write (CIGAR -> segments, tile runs), two block scans; HIP events around 30 repetitions on resident records. "
                                    "Runs once per chunk in `extract`; the step of this benchmark is the pileup over the segments it leaves resident"},
        }
        if world > 1:
            result["exchange"] = {"exchanges": exchanges, "bytes_per_exchange_per_rank": bytes_per_exchange, "transport": "ncclSend/ncclRecv group (libmdk_hip md_comm_gather)"}
        if streamed:
            result["streamed"] = streamed
        if dense:
            result["dense_contexts"] = dense
        if not args.no_cpu_baseline and world == 1:
            # CPU baseline on a bounded sample of the same workload: the same generator and parameters at 32 Mb, end to end from
            # the BAM file; the product's CLI is timed on the same file.
            sp = data / f"cpu_sample_{args.cpu_sample_length}_{args.coverage}"
            if not Path(str(sp) + ".bam.bai").exists():
                subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(sp), "-L", str(args.cpu_sample_length), "-c", str(args.coverage), "-s", str(S1_SEED + 1000)] + args.synth_args.split(),
                               capture_output=True, text=True, check=True)
            oracle = REPO / "oracle/_build/mdk_oracle"
            ncores = os.cpu_count() or 1
            # all cores: the reference's chunk-parallel workers need enough chunks to go round, so the chunk size is chosen to
            # give every thread about four (outputs do not depend on --chunkSize)
            chunk_all = max(50_000, args.cpu_sample_length // (4 * ncores))
            timings = {}
            for name, thr, ck in (("single", 1, None), ("allcore", ncores, chunk_all)):
                d = work / f"co_{name}"; d.mkdir()
                opts = ["-@", str(thr)] + (["--chunkSize", str(ck)] if ck else [])
                t1 = time.perf_counter()
                subprocess.run([str(oracle), "extract", str(sp) + ".fa", str(sp) + ".bam"] + opts + extra + ["-o", "out"], check=True, capture_output=True, cwd=d)
                timings[name] = time.perf_counter() - t1
            same = all((work / "co_single" / f).read_bytes() == (work / "co_allcore" / f).read_bytes() for f in os.listdir(work / "co_single"))
            calls = 0
            for line in open(work / "co_single" / "out_CpG.bedGraph"):
                f = line.split("\t")
                if len(f) == 6:
                    calls += int(f[4]) + int(f[5])
            threads = str(min(64, ncores))
            e2e = {}
            for name, env in (("default", {}), ("detached", {"MDK_DETACH": "1"})):
                (work / f"cg_{name}").mkdir()
                best = None
                for _ in range(3):
                    time.sleep(0.6)          # let the previous process' GPU context finish tearing down: back-to-back commands otherwise wait for each other's teardown (up to 0.25 s at start-up and at exit)
                    t1 = time.perf_counter()
                    rg = mdk.run_cli([str(sp) + ".fa", str(sp) + ".bam", "-@", threads] + extra + ["-o", "out"], cwd=work / f"cg_{name}", env=env)
                    dtc = time.perf_counter() - t1
                    best = dtc if best is None else min(best, dtc)
                ident = rg.returncode == 0 and all((work / f"cg_{name}" / f).read_bytes() == (work / "co_single" / f).read_bytes() for f in os.listdir(work / "co_single"))
                e2e[name] = (best, bool(ident))
            t_cpu = timings["allcore"]
            result["cpu_baseline"] = {"value": calls / t_cpu, "unit": "CpG calls/s", "cores": ncores, "kind": "port",
                                      "sample": f"oracle/mdk_oracle extract -@ {ncores} --chunkSize {chunk_all} (C restatement of the reference with its chunk-parallel worker threads, extract.c:325-350,1479-1486; "
                                                f"end to end from the BAM file: inflate, pileup, text) on a {args.cpu_sample_length} bp / {args.coverage}x sample of the same synthetic workload: "
                                                f"{t_cpu:.2f} s, {calls} CpG calls; the reference binary itself cannot be built here (no htslib)",
                                      "seconds": t_cpu, "cpg_calls": calls, "identical_to_single_thread": bool(same),
                                      "single_thread": {"value": calls / timings["single"], "seconds": timings["single"], "cores": 1}}
            best, ident = e2e["default"]
            result["e2e_cli"] = {"seconds": best, "value": calls / best, "unit": "CpG calls/s", "threads": int(threads),
                                 "speedup_vs_cpu_baseline": t_cpu / best, "speedup_vs_single_thread": timings["single"] / best, "identical_to_oracle": ident,
                                 "detached_seconds": e2e["detached"][0], "detached_identical": e2e["detached"][1],
                                 "note": "`MethylDackel extract` of this build on the same file, wall-clock of the whole process, one process (start-up, HIP init, inflate, chunk preparation, H2D, kernels, "
                                         "D2H, text, teardown); detached_seconds = the opt-in MDK_DETACH=1 mode, where the parent returns when the outputs are closed and a child finishes the GPU teardown"}
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

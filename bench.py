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
    ap.add_argument("--xxl-copies", type=int, default=8, help="a third end-to-end sample = this many copies of the large sample (kept in /dev/shm when there is room; <= --xl-copies: skip)")
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
    # algorithmic bytes per LAUNCH (GROUP chunks).  Pileup: SURVEY.md 8d (reads' payload + reference + sites).  Preparation: every byte of the
    # chunk's records once (the fields it needs sit at both ends of a record: every 128-byte line is touched) + the 32-byte segments it
    # writes; its intermediates (64 B per admitted read, the name table) are overhead, not algorithmic bytes.
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
            oracle = REPO / "oracle/_build/mdk_oracle"
            ncores = os.cpu_count() or 1
            threads = str(min(64, ncores))

            def sample(length, tag):
                sp = data / f"cpu_sample_{length}_{args.coverage}"
                if not Path(str(sp) + ".bam.bai").exists():
                    subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(sp), "-L", str(length), "-c", str(args.coverage), "-s", str(S1_SEED + 1000)] + args.synth_args.split(),
                                   capture_output=True, text=True, check=True)
                return sp

            def run_oracle(sp, name, thr, ck, runs):
                d = work / f"co_{name}"; d.mkdir()
                opts = ["-@", str(thr)] + (["--chunkSize", str(ck)] if ck else [])
                r = median_run(lambda: subprocess.run([str(oracle), "extract", str(sp) + ".fa", str(sp) + ".bam"] + opts + extra + ["-o", "out"], check=True, capture_output=True, cwd=d, timeout=900), runs), d
                log(f"[bench] oracle {name}: {r[0][1]}")
                return r

            def wait_gone(marker, limit=8.0):
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

            def run_ours(sp, name, env, runs=3, gap=1.0):
                d = work / f"cg_{name}"; d.mkdir(); rcs = []; ts = []; inner = []; profs = []
                for _ in range(runs):
                    # (outside the clock) the previous command's process -- by default a child the command does not wait for -- is still taking its address
                    # space down for 0.2-0.4 s after the command has returned; a command started into that shares the driver's locks with it and is
                    # slower itself, mostly in the time until its device is usable (0.16 -> 0.25-0.45 s; 5 runs 0.3 s apart: 0.40, 0.49, 0.65, 0.75,
                    # 0.80 s; gpurun_out r04t/r04u).  Runs are measured in isolation: the next one starts a second after the previous one's last
                    # process has gone
                    if gap > 0:
                        wait_gone(str(sp) + ".bam"); time.sleep(gap)        # (the driver goes on releasing a process's GPU resources for a while after the process is gone; gap = 0: a queue of samples, back to back)
                    t1 = time.perf_counter()
                    r = mdk.run_cli([str(sp) + ".fa", str(sp) + ".bam", "-@", threads] + extra + ["-o", "out"], cwd=d, env=dict(env, MDK_HOST_PROFILE="1"), timeout=300)
                    ts.append(time.perf_counter() - t1); rcs.append(r.returncode)
                    m = re.search(r"total ([0-9.]+)s; chunks prepared", r.stderr)          # the command's own clock, entry of extract_main to outputs closed
                    inner.append(float(m.group(1)) if m else None)
                    profs.append([l[:400] for l in r.stderr.splitlines() if l.startswith("[mdk")])
                log(f"[bench] {name}: wall {ts} inside the process {inner}")
                # a run far off the others is kept with its own account of where the time went (MDK_HOST_PROFILE lines), not just as a number
                med = statistics.median(ts)
                for k, t in enumerate(ts):
                    if t > 1.6 * med and t - med > 0.3:
                        slow_runs.append({"leg": name, "run": k, "seconds": t, "median": med, "profile": profs[k]})
                return med, ts, d, all(r == 0 for r in rcs), inner

            slow_runs = []

            def calls_of(d):
                n = 0
                for line in open(d / "out_CpG.bedGraph"):
                    f = line.split("\t")
                    if len(f) == 6:
                        n += int(f[4]) + int(f[5])
                return n

            sp = sample(args.cpu_sample_length, "small")
            # the CPU baseline is the BEST the CPU path does over a small sweep of worker threads x chunk size (one run each), then 3 runs at that
            # setting; the reference's chunk-parallel workers need enough chunks to go round, and outputs do not depend on --chunkSize
            (t_single, ts_single), d_single = run_oracle(sp, "single", 1, None, 3)
            sweep = []
            for thr, ck in ((32, 250_000), (64, 50_000), (64, 250_000), (128, 1_000_000), (ncores, max(50_000, args.cpu_sample_length // (4 * ncores)))):
                if thr > ncores:
                    continue
                (t1, _), _ = run_oracle(sp, f"sweep_{thr}_{ck}", thr, ck, 1)
                sweep.append({"threads": thr, "chunk_size": ck, "seconds": t1})
            best = min(sweep, key=lambda q: q["seconds"])
            (t_all, ts_all), d_all = run_oracle(sp, "allcore", best["threads"], best["chunk_size"], 3)
            same = all((d_single / f).read_bytes() == (d_all / f).read_bytes() for f in os.listdir(d_single))
            calls = calls_of(d_single)
            # The wall clock on top is the command as it runs by default: the process that did the work is the process the caller waits for, its
            # teardown included -- the CPU baseline's protocol.  With MDK_DETACH=1 the command does its work in a child and returns when the child
            # reports its outputs closed (csrc/host/main.c); that figure is reported next to it as `detached`.
            t_g, ts_g, d_g, ok_g, in_g = run_ours(sp, "inplace", {})
            t_gd, ts_gd, _, ok_gd, _ = run_ours(sp, "detached", {"MDK_DETACH": "1"})
            ident = ok_g and ok_gd and all((d_g / f).read_bytes() == (d_single / f).read_bytes() for f in os.listdir(d_single))
            result["cpu_baseline"] = {"value": calls / t_all, "unit": "CpG calls/s", "cores": best["threads"], "kind": "port",
                                      "sample": f"oracle/mdk_oracle extract -@ {best['threads']} --chunkSize {best['chunk_size']} -- the fastest of a sweep over worker threads x chunk size on this box's {ncores} hardware threads "
                                                f"(C restatement of the reference with its chunk-parallel worker threads, extract.c:325-350,1479-1486; end to end from the BAM file: inflate with CRC32 check, pileup, text) "
                                                f"on a {args.cpu_sample_length} bp / {args.coverage}x sample of the same synthetic workload; 3 runs, median "
                                                f"{t_all:.2f} s, {calls} CpG calls; the reference binary itself cannot be built here (no htslib)",
                                      "seconds": t_all, "runs": ts_all, "cpg_calls": calls, "identical_to_single_thread": bool(same), "host_threads": ncores, "sweep": sweep,
                                      "single_thread": {"value": calls / t_single, "seconds": t_single, "runs": ts_single, "cores": 1}}
            result["e2e_cli"] = {"seconds": t_g, "runs": ts_g, "value": calls / t_g, "unit": "CpG calls/s", "threads": int(threads), "protocol": "3 runs, median, whole-process wall clock with the teardown in place (the CPU baseline's protocol)",
                                 "speedup_vs_cpu_baseline": t_all / t_g, "speedup_vs_single_thread": t_single / t_g, "identical_to_oracle": bool(ident),
                                 "inside_process_runs": in_g, "bam_bytes": os.path.getsize(str(sp) + ".bam"),
                                 "detached": {"seconds": t_gd, "runs": ts_gd, "speedup_vs_cpu_baseline": t_all / t_gd,
                                              "note": "MDK_DETACH=1 (opt-in): the command's work is done by a child, the command returns when the child reports its outputs closed and the child's address-space teardown goes on behind the caller"},
                                 "note": "`MethylDackel extract` of this build on the same file, one process (start-up, HIP init, inflate on the host's threads and -- once the device is up -- on the device, "
                                         "chunk preparation, H2D, kernels, D2H, text, teardown).  `seconds` is the caller's wall clock around a process that tears its own address space down (the default); "
                                         "inside_process_runs = the command's own clock from entry to outputs closed"}
            # the device inflate on the record: the 32 Mb sample's BGZF members through md_piece_* (tools/piece_bench: whole file in 64 MB pieces, three in
            # flight, and the kernels alone on the largest resident piece, HIP events)
            try:
                pb = subprocess.run([str(REPO / "tools/_build/piece_bench"), str(sp) + ".bam", "64", "3", "0"], capture_output=True, text=True, timeout=300)
                pj = json.loads(pb.stdout)
                kv = pj["kernel_only"]["v0"]
                inf_bytes = kv["comp_bytes"] + kv["out_bytes"]
                result["inflate"] = {"workload": f"the {args.cpu_sample_length} bp sample's BAM: {pj['members']} BGZF members, {pj['file_MB']:.0f} MB -> {pj['inflated_MB']:.0f} MB",
                                     "pipelined": {"GBps_compressed": pj["pass1"]["GBps_compressed"], "GBps_inflated": pj["pass1"]["GBps_inflated"], "seconds": pj["pass1"]["seconds"],
                                                   "note": "whole file through md_piece_submit / md_piece_wait: staging copy, H2D of the compressed bytes, k_inflate, k_crc32, k_walk, digests back; 64 MB pieces, 3 in flight"},
                                     "GBps_compressed": kv["GBps_compressed"], "GBps_inflated": kv["GBps_inflated"],
                                     "roofline": {"kernel": "k_inflate", "bound": "hbm", "kernel_ms": kv["inflate_ms"], "algo_bytes_per_launch": inf_bytes, "achieved": inf_bytes / (kv["inflate_ms"] * 1e6), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": inf_bytes / (kv["inflate_ms"] * 1e6) / HBM_PEAK_GBS, "members_per_launch": pj["kernel_only"]["piece_members"],
                                                  "note": "algorithmic bytes = compressed bytes read + inflated bytes written, one 64 MB piece per launch; the kernel is bound by one lane's dependent symbol decode per member, not by HBM (DESIGN.md 4)"},
                                     "crc32_ms": kv["crc32_ms"], "walk_ms": kv["walk_ms"], "crc32_GBps": kv["out_bytes"] / (kv["crc32_ms"] * 1e6) if kv["crc32_ms"] > 0 else None}
                # a 64 MB piece is ~3,400 members = wavefronts, about half of what the device holds at once; the command keeps 8 pieces (of 96 MB since the end of round 5) in flight (8 device teams).
                # What the kernel does with the device full: the whole file as one launch
                pw = subprocess.run([str(REPO / "tools/_build/piece_bench"), str(sp) + ".bam", "1024", "1", "0"], capture_output=True, text=True, timeout=300)
                kw = json.loads(pw.stdout)["kernel_only"]["v0"]
                result["inflate"]["device_full"] = {"members_per_launch": json.loads(pw.stdout)["kernel_only"]["piece_members"], "kernel_ms": kw["inflate_ms"], "GBps_compressed": kw["GBps_compressed"], "GBps_inflated": kw["GBps_inflated"],
                                                    "crc32_ms": kw["crc32_ms"], "walk_ms": kw["walk_ms"], "identical_to_zlib": kw.get("identical_to_zlib"),
                                                    "note": "k_inflate over all members of the file in one launch (what the command's eight pieces in flight present to the device); HIP events"}
            except Exception as ex:
                result["inflate"] = {"error": repr(ex)[:300]}
            if args.large_sample_length and headline:
                spl = sample(args.large_sample_length, "large")
                alt = (32, 250_000) if best["threads"] != 32 else (64, 250_000)
                (t_alt, _), _ = run_oracle(spl, "large_alt", alt[0], alt[1], 1)
                (t_la, ts_la), d_la = run_oracle(spl, "large_allcore", best["threads"], best["chunk_size"], 2)
                cfg_l = {"threads": best["threads"], "chunk_size": best["chunk_size"]}
                if t_alt < t_la:
                    (t_la, ts_la), d_la = run_oracle(spl, "large_alt3", alt[0], alt[1], 2); ts_la = ts_la + [t_alt]; t_la = statistics.median(ts_la); cfg_l = {"threads": alt[0], "chunk_size": alt[1]}
                t_lg, ts_lg, d_lg, ok_lg, in_lg = run_ours(spl, "large_inplace", {}, runs=5)
                t_ld, ts_ld, _, ok_ld, _ = run_ours(spl, "large_detached", {"MDK_DETACH": "1"}, runs=3)
                t_lq, ts_lq, _, ok_lq, _ = run_ours(spl, "large_queue", {}, runs=4, gap=0.0)
                ident_l = ok_lg and ok_ld and ok_lq and all((d_lg / f).read_bytes() == (d_la / f).read_bytes() for f in os.listdir(d_la))
                calls_l = calls_of(d_la)
                bam_l = os.path.getsize(str(spl) + ".bam")
                result["e2e_large"] = {"sample_bp": args.large_sample_length, "bam_bytes": bam_l, "cpg_calls": calls_l,
                                       "cpu_all_cores_seconds": t_la, "cpu_runs": ts_la, "cpu_setting": cfg_l, "seconds": t_lg, "runs": ts_lg, "value": calls_l / t_lg, "unit": "CpG calls/s",
                                       "speedup_vs_cpu_all_cores": t_la / t_lg, "identical_to_oracle": bool(ident_l),
                                       "inside_process_runs": in_lg, "bam_GBps": bam_l / t_lg / 1e9,
                                       "detached": {"seconds": t_ld, "runs": ts_ld, "speedup_vs_cpu_all_cores": t_la / t_ld,
                                                    "note": "MDK_DETACH=1 (opt-in: work in a child, return at outputs closed, teardown behind the caller): csrc/host/main.c detach_teardown, as the mold linker does"},
                                       "queue": {"seconds_per_sample": t_lq, "runs": ts_lq, "speedup_vs_cpu_all_cores": t_la / t_lq,
                                                 "note": "four runs back to back with no pause between them, teardown in place: what a queue of samples gets per sample"},
                                       "protocol": "CPU: the sweep's best setting and one alternative, the faster of them, median; this build: 5 runs, median; the caller's wall clock around the command with the teardown in place (the default), each run started one second after the previous command's last process has gone"}
                if args.xl_copies > 1:
                    # a sample large enough that start-up and exit are a small part of the run: K copies of the large sample as K contigs (tools/mdk_replicate)
                    spx = data / f"xl_{args.large_sample_length}x{args.xl_copies}_{args.coverage}"
                    if not Path(str(spx) + ".bam").exists():
                        t1 = time.time()
                        subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(spl), str(spx), str(args.xl_copies)], check=True, capture_output=True, timeout=600)
                        log(f"[bench] xl sample written in {time.time() - t1:.1f} s")
                    (t_xa, ts_xa), d_xa = run_oracle(spx, "xl_allcore", cfg_l["threads"], cfg_l["chunk_size"], 1)
                    t_xg, ts_xg, d_xg, ok_xg, in_xg = run_ours(spx, "xl_inplace", {}, runs=3)
                    t_xd, ts_xd, _, ok_xd, _ = run_ours(spx, "xl_detached", {"MDK_DETACH": "1"}, runs=2)
                    ident_x = ok_xg and ok_xd and all((d_xg / f).read_bytes() == (d_xa / f).read_bytes() for f in os.listdir(d_xa))
                    calls_x = calls_of(d_xa); bam_x = os.path.getsize(str(spx) + ".bam")
                    result["e2e_xl"] = {"sample_bp": args.large_sample_length * args.xl_copies, "contigs": args.xl_copies, "bam_bytes": bam_x, "cpg_calls": calls_x,
                                        "cpu_all_cores_seconds": t_xa, "cpu_runs": ts_xa, "cpu_setting": cfg_l, "seconds": t_xg, "runs": ts_xg, "value": calls_x / t_xg, "unit": "CpG calls/s",
                                        "speedup_vs_cpu_all_cores": t_xa / t_xg, "identical_to_oracle": bool(ident_x), "inside_process_runs": in_xg,
                                        "detached": {"seconds": t_xd, "runs": ts_xd, "speedup_vs_cpu_all_cores": t_xa / t_xd},
                                        "bam_GBps": bam_x / t_xg / 1e9, "bam_GBps_inside_process": [bam_x / q / 1e9 if q else None for q in in_xg],
                                        "protocol": "CPU: one run at the large sample's setting; this build: 3 runs, median; whole-process wall clock, teardown in place"}
                if args.xxl_copies > max(1, args.xl_copies):
                    # ... and one where start-up and teardown are a small part of this build's run too: in RAM-backed storage when the box has room for it
                    shm = Path("/dev/shm")
                    need = bam_l * (args.xxl_copies + 1)
                    xdir = data
                    try:
                        if shm.is_dir() and shutil.disk_usage(shm).free > 3 * need:
                            xdir = shm / f"mdk_bench_xxl_{os.getpid()}"; xdir.mkdir(exist_ok=True)
                    except OSError:
                        pass
                    if shutil.disk_usage(xdir).free > 2 * need:
                        spy = xdir / f"xxl_{args.large_sample_length}x{args.xxl_copies}_{args.coverage}"
                        try:
                            t1 = time.time()
                            subprocess.run([str(REPO / "tools/_build/mdk_replicate"), str(spl), str(spy), str(args.xxl_copies)], check=True, capture_output=True, timeout=900)
                            log(f"[bench] xxl sample written in {time.time() - t1:.1f} s under {xdir}")
                            (t_ya, ts_ya), d_ya = run_oracle(spy, "xxl_allcore", cfg_l["threads"], cfg_l["chunk_size"], 1)
                            t_yg, ts_yg, d_yg, ok_yg, in_yg = run_ours(spy, "xxl_inplace", {}, runs=2)
                            t_yd, ts_yd, _, ok_yd, _ = run_ours(spy, "xxl_detached", {"MDK_DETACH": "1"}, runs=1)
                            ident_y = ok_yg and ok_yd and all((d_yg / f).read_bytes() == (d_ya / f).read_bytes() for f in os.listdir(d_ya))
                            calls_y = calls_of(d_ya); bam_y = os.path.getsize(str(spy) + ".bam")
                            result["e2e_xxl"] = {"sample_bp": args.large_sample_length * args.xxl_copies, "contigs": args.xxl_copies, "bam_bytes": bam_y, "cpg_calls": calls_y, "storage": str(xdir),
                                                 "cpu_all_cores_seconds": t_ya, "cpu_runs": ts_ya, "cpu_setting": cfg_l, "seconds": t_yg, "runs": ts_yg, "value": calls_y / t_yg, "unit": "CpG calls/s",
                                                 "speedup_vs_cpu_all_cores": t_ya / t_yg, "identical_to_oracle": bool(ident_y), "inside_process_runs": in_yg,
                                                 "detached": {"seconds": t_yd, "runs": ts_yd, "speedup_vs_cpu_all_cores": t_ya / t_yd},
                                                 "bam_GBps": bam_y / t_yg / 1e9,
                                                 "protocol": "CPU: one run at the large sample's setting; this build: 2 runs, median; whole-process wall clock, teardown in place"}
                        finally:
                            if xdir != data:
                                shutil.rmtree(xdir, ignore_errors=True)
            if slow_runs:
                result["slow_runs"] = slow_runs
          except Exception as ex:            # a leg that fails or hangs (timeout) must not take the measured line with it
            result["legs_error"] = repr(ex)[:500]
            log(f"[bench] a CPU/end-to-end leg failed: {ex!r}")
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

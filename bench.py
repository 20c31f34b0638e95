#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X `MethylDackel extract` hot path.

Metric (BASELINE.json): CpG calls/s on the synthetic 1 Mb contig, 30x paired-end WGBS BAM (configs[1], "S1"),
CpG-only extract.  One "step" = one pass of the device path (pileup + scan + gather kernels) over the admitted reads
of one S1 interval, inputs already resident in HBM.  With N ranks every rank owns its own S1 interval (weak scaling:
independent intervals, as the path shards by contig/interval) and, after each step, the per-interval site buffers are
gathered to rank 0 over RCCL -- the one real exchange step of the path.

Also reported on the same JSON line:
  roofline     -- pileup kernel: algorithmic bytes (SURVEY.md 8d formula) / HIP-event kernel time vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (`oracle/`, single thread, "port") timed on the same S1 BAM on this box's host
"""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
S1_SEED = 0x5EED0001


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--length", type=int, default=1_000_000, help="interval length per rank (S1 = 1 Mb)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--extra", default="", help="extra extract options, e.g. '--CHG --CHH' (not the headline config)")
    ap.add_argument("--synth-args", default="", help="extra mdk_synth options, e.g. '--clean' (not the headline config)")
    ap.add_argument("--cpu-sample-length", type=int, default=32_000_000, help="bp of the same synthetic workload the CPU oracle is timed on (about 10 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    # MDK_BENCH_BACKEND=gloo is a test mode for boxes with fewer GPUs than ranks (ranks then share devices and the exchange is
    # staged through host memory); the real multi-GPU run uses "nccl", i.e. RCCL over xGMI
    backend = os.environ.get("MDK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    cdev = "cuda" if backend == "nccl" else "cpu"          # where the tensors of the small bookkeeping collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import methyldackel_amd as mdk
    if rank == 0:
        mdk.build()
    if world > 1:
        dist.barrier()

    work = Path(tempfile.mkdtemp(prefix=f"mdk_bench_r{rank}_"))
    prefix = work / "S1"
    synth = subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(prefix), "-L", str(args.length), "-c", str(args.coverage),
                            "-s", str(S1_SEED + rank)] + args.synth_args.split(), capture_output=True, text=True, check=True)
    synth_info = json.loads(synth.stdout)
    extra = args.extra.split()
    cmd = [str(prefix) + ".fa", str(prefix) + ".bam", "--chunkSize", str(args.length)] + extra + ["-o", str(work / "gpu")]

    # host side: decode + admit + pack the interval once; it then stays resident in HBM
    t0 = time.time()
    plan = mdk.Plan(cmd)
    cfg = plan.dev_cfg()
    chunk = plan.next_chunk()
    t_host = time.time() - t0
    assert chunk is not None and not chunk.skipped
    dev = mdk.Device(cfg, device=dev_index)
    plan.ensure_reference(dev, chunk.tid)
    dev.upload(0, chunk.batch)
    dev.launch(0)
    sites = dev.download(0)
    n_sites = sites.n_sites
    cpg_calls = 0
    all_calls = 0
    for i in range(n_sites):
        r = sites.site[i]
        c = r.nmeth + r.nunmeth
        all_calls += c
        if ((r.meta >> 1) & 3) == 0:
            cpg_calls += c
    variant = cfg.minOppositeDepth > 0
    L = mdk.lib_hip()

    # the kernel writes its result straight into torch tensors (md_dev_bind_output), which is what travels over RCCL.
    # Two slots, as in extract_main: chunk k is launched while chunk k-1 is collected (two chunks in flight).
    w0 = dev.wait(0)
    n_tiles = w0.n_tiles
    cap = int(w0.n_slots) + 1024             # slots = kept context positions of the interval: fixed by the reference, not by the reads
    dev.upload(1, chunk.batch)
    t_site = [torch.zeros((cap, 4), dtype=torch.int32, device="cuda") for _ in range(2)]
    t_var = [torch.zeros((cap, 2), dtype=torch.int32, device="cuda") if variant else None for _ in range(2)]
    t_seg = [torch.zeros((n_tiles + 1, 2), dtype=torch.int32, device="cuda") for _ in range(2)]
    for sl in range(2):
        dev.bind_output(sl, C.c_void_p(t_site[sl].data_ptr()), C.c_void_p(t_var[sl].data_ptr()) if variant else None, C.c_void_p(t_seg[sl].data_ptr()), cap, n_tiles + 1)

    # N > 1: the exchange step of the sharded path -- per-interval site buffers travel to rank 0 (RCCL gather over xGMI).
    # The kernels write straight into the send buffer (no staging copy), and the results of GROUP consecutive steps
    # travel together: fewer, larger collectives, the next group being computed while the previous one is on the links.
    GROUP = 8
    if world > 1:
        shape = torch.tensor([cap, n_tiles + 1], dtype=torch.int64, device=cdev)
        shapes = [torch.zeros(2, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(shapes, shape)
        gcap = int(max(int(x[0].item()) for x in shapes)); gtiles = int(max(int(x[1].item()) for x in shapes))
        E = gcap * 4 + gtiles * 2                                  # int32 words of one step: sites, then tile segments
        sendbuf = [torch.zeros((GROUP * E,), dtype=torch.int32, device="cuda") for _ in range(2)]
        if backend == "nccl":
            recvbuf = [[torch.empty_like(sendbuf[0]) for _ in range(world)] if rank == 0 else None for _ in range(2)]
        else:
            recvbuf = [[torch.empty((GROUP * E,), dtype=torch.int32) for _ in range(world)] if rank == 0 else None for _ in range(2)]
        pending = [None, None]
        t_dummy_var = t_var

    def bind_for(k):
        """step k writes into step-slot k % GROUP of send buffer (k // GROUP) & 1"""
        if world == 1:
            return
        x, e = (k // GROUP) & 1, k % GROUP
        if e == 0 and pending[x] is not None:                      # this buffer is about to be overwritten: its gather must be over
            pending[x].wait(); pending[x] = None
        base = sendbuf[x].data_ptr() + 4 * e * E
        sl = k & 1
        dev.bind_output(sl, C.c_void_p(base), C.c_void_p(t_dummy_var[sl].data_ptr()) if variant else None, C.c_void_p(base + 16 * gcap), gcap, gtiles)

    def exchange(k, last):
        """after step k has been collected: send its group when the group is complete (or the run ends)"""
        if world == 1 or not (k % GROUP == GROUP - 1 or last):
            return
        x = (k // GROUP) & 1
        if backend == "nccl":
            pending[x] = dist.gather(sendbuf[x], recvbuf[x], dst=0, async_op=True)
        else:
            dist.gather(sendbuf[x].cpu(), recvbuf[x], dst=0)

    def run(k_steps):
        """k_steps passes over the batch: every pass is launched, collected and (N > 1) exchanged inside the call"""
        n = 0
        for k in range(k_steps):
            bind_for(k)
            dev.launch(k & 1)
            if k:
                n = dev.wait((k - 1) & 1).n_slots
                exchange(k - 1, False)
        if k_steps:
            n = dev.wait((k_steps - 1) & 1).n_slots
            exchange(k_steps - 1, True)
        if world > 1:
            for x in range(2):
                if pending[x] is not None:
                    pending[x].wait(); pending[x] = None
        return n

    def fence():
        if world > 1:
            dist.barrier()
        dev.sync()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    n_last = run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    assert n_last >= n_sites
    # the bound buffers hold the same sites as the library's own download (segment order -> ascending)
    last = (args.steps - 1) & 1 if args.steps else 0
    if world > 1 and args.steps:
        kl = args.steps - 1
        flat = sendbuf[(kl // GROUP) & 1][(kl % GROUP) * E:(kl % GROUP + 1) * E].cpu().numpy().view("uint32")
        chk = flat[: gcap * 4].reshape(gcap, 4); segs = flat[gcap * 4:].reshape(gtiles, 2)
    else:
        chk = t_site[last].cpu().numpy().view("uint32"); segs = t_seg[last].cpu().numpy().view("uint32")
    got = []
    for t in range(n_tiles):
        o, c = int(segs[t, 0]), int(segs[t, 1])
        got.extend(int(x) for x in chk[o:o + c, 0])
    assert got == [sites.site[i].pos for i in range(n_sites)], "bound-output sites differ from md_dev_download"
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([cpg_calls, all_calls, int(n_sites)], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_cpg_calls, total_calls, total_sites = (int(x) for x in tot.tolist())
        if rank == 0 and args.steps:            # rank 0 really received every rank's interval: the tile segments of the last step hold sites
            kl = args.steps - 1
            for r in recvbuf[(kl // GROUP) & 1]:
                one = r[(kl % GROUP) * E:(kl % GROUP + 1) * E]
                assert int(one[gcap * 4:].view(-1, 2)[:, 1].sum().item()) > 0
    else:
        total_cpg_calls, total_calls, total_sites = cpg_calls, all_calls, int(n_sites)

    # kernel-level timing with HIP events on the launch stream, inside the library
    br = dev.bench(0, 5, 50)
    pile_s = br.ms_pileup / 1e3
    achieved = br.algo_bytes / pile_s / 1e9 if pile_s > 0 else 0.0

    # HBM traffic of the kernel cannot be sampled from inside this process; it is taken from the committed rocprofv3 PMC summary
    # of this same command (profiles/, produced by tools/gpu_round.sh + tools/summarize_prof.py), or left null
    traffic, traffic_note = None, None
    try:
        prof = json.load(open(REPO / "profiles" / "r01_rocprofv3_pmc_summary.json"))
        hb = prof["hbm_traffic_bytes_per_launch"]
        if not extra and not args.synth_args and args.length == 1_000_000:
            traffic = hb["fetch_x2_gfx950_correction"] + hb["write_raw"]
            traffic_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per k_pileup dispatch (profiles/r01_rocprofv3_pmc_summary.json): FETCH_SIZE KiB x 2 (gfx950 under-count, "
                            "MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB; raw FETCH_SIZE is %.0f bytes" % hb["fetch_raw"])
    except Exception:
        pass

    result = None
    if rank == 0:
        value = total_cpg_calls * args.steps / dt
        result = {
            "metric": "CpG calls/sec, synthetic 1 Mb contig 30x paired-end WGBS BAM, CpG extract",
            "value": value, "unit": "CpG calls/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": "S1: synthetic 1 Mb contig, 30x PE 2x150 WGBS BAM, CpG-only extract (BASELINE.json configs[1])" if not extra and not args.synth_args and args.length == 1_000_000
                       else f"synthetic {args.length} bp, {args.coverage}x, extract {' '.join(extra)}",
                       "interval_bp": args.length, "coverage": args.coverage, "reads_admitted_per_gpu": int(chunk.batch.n_reads), "segments_per_gpu": int(chunk.batch.n_segs),
                       "records_per_gpu": synth_info["records"], "sites_per_gpu": int(n_sites), "cpg_calls_per_gpu": int(cpg_calls),
                       "tile": int(br.tile), "tiles": int(br.n_tiles), "lds_bytes_per_workgroup": int(br.lds_bytes), "parallelism": f"interval-sharded x{n_gpus}" + (" + RCCL gather of site buffers" if world > 1 else ""),
                       "in_flight": "2 chunks per GPU (step k is launched while step k-1 is collected, as extract_main does); every step is launched and collected inside the timed region"},
            "roofline": {"bound": "hbm", "kernel": "k_pileup", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "algo_bytes_per_launch": int(br.algo_bytes), "kernel_ms": br.ms_pileup, "all_kernels_ms": br.ms_total},
            "host_prep_s": t_host,
        }
        if not args.no_cpu_baseline and world == 1:
            # CPU baseline on a bounded sample of the same workload: the same generator and parameters at 32 Mb (about 10 s of
            # single-thread CPU time for the oracle), end to end from the BAM file; the product's CLI is timed on the same file.
            sp = work / "cpu_sample"
            subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(sp), "-L", str(args.cpu_sample_length), "-c", str(args.coverage), "-s", str(S1_SEED + 1000)] + args.synth_args.split(),
                           capture_output=True, text=True, check=True)
            oracle = REPO / "oracle/_build/mdk_oracle"
            (work / "co").mkdir(); (work / "cg").mkdir()
            t1 = time.perf_counter()
            subprocess.run([str(oracle), "extract", str(sp) + ".fa", str(sp) + ".bam"] + extra + ["-o", "out"], check=True, capture_output=True, cwd=work / "co")
            t_cpu = time.perf_counter() - t1
            calls = 0
            for line in open(work / "co" / "out_CpG.bedGraph"):
                f = line.split("\t")
                if len(f) == 6:
                    calls += int(f[4]) + int(f[5])
            threads = str(min(64, os.cpu_count() or 1))
            best = None
            for _ in range(2):
                t1 = time.perf_counter()
                rg = mdk.run_cli([str(sp) + ".fa", str(sp) + ".bam", "-@", threads] + extra + ["-o", "out"], cwd=work / "cg")
                dtc = time.perf_counter() - t1
                best = dtc if best is None else min(best, dtc)
            ident = rg.returncode == 0 and all((work / "cg" / f).read_bytes() == (work / "co" / f).read_bytes() for f in os.listdir(work / "co"))
            result["cpu_baseline"] = {"value": calls / t_cpu, "unit": "CpG calls/s", "cores": 1, "kind": "port",
                                      "sample": f"oracle/mdk_oracle extract (single-thread C restatement of the reference, end to end from the BAM file: inflate, pileup, text) on a "
                                                f"{args.cpu_sample_length} bp / {args.coverage}x sample of the same synthetic workload: {t_cpu:.2f} s, {calls} CpG calls; "
                                                f"the reference binary itself cannot be built here (no htslib)",
                                      "seconds": t_cpu, "cpg_calls": calls}
            result["e2e_cli"] = {"seconds": best, "value": calls / best, "unit": "CpG calls/s", "threads": int(threads), "speedup_vs_cpu_baseline": t_cpu / best,
                                 "identical_to_oracle": bool(ident),
                                 "note": "`MethylDackel extract` of this build on the same file, wall-clock of the whole process (start-up, HIP init ~0.4 s, inflate, pack, H2D, kernels, D2H, text)"}
        print(json.dumps(result), flush=True)
    dev.close()
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

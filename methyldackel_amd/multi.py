"""Interval-sharded `extract` over several GPUs of one node: one process per GPU (torchrun), RCCL for the exchange.

The path shards naturally: per-position counts depend only on the reads overlapping the position, and
the reference already processes the genome as independent chunks (extract.c:325-350).  Chunk k of the reference's
schedule is owned by rank k % world.  Every rank walks the same schedule (mdk_plan_set_shard) but admits, packs and
counts only its own chunks; after each round of `world` chunks the per-chunk site buffers are gathered to rank 0 --
the one real exchange step -- and rank 0 replays the chunks in index order through the host emitters, so the output
files are byte-identical to a single-GPU run.

The counting step is injectable (`count_fn`) so that the sharding / gather / ordered-emit logic can be exercised on a
CPU-only box with the gloo backend (tests/test_sharding_gloo.py); the product always uses the GPU (`device_count_fn`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

import methyldackel_amd as mdk


def device_count_fn(dev: "mdk.Device"):
    """chunk -> (sites ndarray [n,4] uint32, var ndarray [n,2] uint32 or None), computed on this rank's GPU"""
    def fn(plan, chunk):
        plan.ensure_reference(dev, chunk.tid)
        dev.submit(0, chunk.batch)
        s = dev.download(0)
        n = s.n_sites
        sites = np.ctypeslib.as_array(C.cast(s.site, C.POINTER(C.c_uint32)), shape=(n, 4)).copy() if n else np.zeros((0, 4), np.uint32)
        var = None
        if s.var:
            var = np.ctypeslib.as_array(C.cast(s.var, C.POINTER(C.c_uint32)), shape=(n, 2)).copy() if n else np.zeros((0, 2), np.uint32)
        return sites, var
    return fn


def _as_md_sites(sites: np.ndarray, var):
    s = mdk.md_sites()
    s.n_sites = int(sites.shape[0])
    sites = np.ascontiguousarray(sites, dtype=np.uint32)
    s.site = sites.ctypes.data_as(C.POINTER(mdk.md_site))
    keep = [sites]
    if var is not None:
        var = np.ascontiguousarray(var, dtype=np.uint32)
        s.var = var.ctypes.data_as(C.POINTER(mdk.md_site_var))
        keep.append(var)
    return s, keep


def extract_sharded(args, count_fn_factory, device=None):
    """Run one `extract` command line sharded over the initialised process group.  `count_fn_factory(plan)` returns the
    per-chunk counting function for this rank.  Returns the number of chunks this rank counted."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev_t = device if device is not None else torch.device("cpu")
    if rank != 0:
        os.environ["MDK_NO_OUTPUT"] = "1"
    try:
        plan = mdk.Plan(args)
    finally:
        os.environ.pop("MDK_NO_OUTPUT", None)
    plan.set_shard(rank, world)
    count_fn = count_fn_factory(plan)
    variant = plan.dev_cfg().minOppositeDepth > 0
    width = 6 if variant else 4
    mine = 0
    done = False
    while not done:
        # one round: `world` consecutive chunks, one per rank
        meta, owned = [], None
        for _ in range(world):
            c = plan.next_chunk()
            if c is None:
                done = True
                break
            meta.append(c)
            if not (c.skipped & mdk.CHUNK_FOREIGN) and not (c.skipped & mdk.CHUNK_EMPTY):
                # the exchange below carries ONE buffer per rank and round, addressed by `index % world`
                assert owned is None and c.index % world == rank, "a round must hold exactly one chunk per owner"
                sites, var = count_fn(plan, c)
                owned = np.concatenate([sites, var], axis=1) if variant else sites
                mine += 1
        # exchange: sizes first, then padded buffers to rank 0 (disjoint intervals: a gather, not a reduction)
        n_own = 0 if owned is None else owned.shape[0]
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev_t) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([n_own], dtype=torch.int64, device=dev_t))
        cap = max(int(x.item()) for x in sizes)
        send = torch.zeros((max(cap, 1), width), dtype=torch.int32, device=dev_t)
        if n_own:
            send[:n_own] = torch.from_numpy(owned.view(np.int32)).to(dev_t)
        recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, recv, dst=0)
        if rank == 0:
            for c in meta:
                if c.skipped & mdk.CHUNK_EMPTY:
                    plan.emit(c, mdk.md_sites())
                    continue
                owner = c.index % world
                n = int(sizes[owner].item())
                arr = recv[owner][:n].cpu().numpy().view(np.uint32)
                s, keep = _as_md_sites(arr[:, :4], arr[:, 4:6] if variant else None)
                plan.emit(c, s)
    if rank == 0:
        plan.finish()
    plan.close()
    return mine


def main(argv=None):
    """torchrun entry point: python -m torch.distributed.run --nproc-per-node N -m methyldackel_amd.multi [extract options] ref.fa aln.bam"""
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "extract":
        argv = argv[1:]
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("methyldackel_amd.multi needs GPUs (there is no CPU path)")
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    devs = {}

    def factory(plan):
        devs["d"] = mdk.Device(plan.dev_cfg(), device=local)
        return device_count_fn(devs["d"])

    extract_sharded(argv, factory, device=torch.device("cuda", local))
    devs["d"].close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The N>1 path on CPU: world_size-2 gloo run of the interval-sharded driver (methyldackel_amd/multi.py).
What runs here is the product's host logic -- schedule agreement across ranks, shard ownership, the gather of per-chunk
site buffers, ordered emission on rank 0 -- with the GPU counting step replaced by the oracle's per-column counters
(test infrastructure).  The outputs must be byte-identical to the oracle's single-process run."""
import filecmp
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import read_dump, run_oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, args, dump_path, ret):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import methyldackel_amd as mdk
    from methyldackel_amd import multi
    dump = read_dump(dump_path)
    variant = "--minOppositeDepth" in args

    def factory(plan):
        def count(plan, c):     # oracle counters restricted to this chunk: stands in for the device
            rows = sorted((pos, v) for (tid, pos), v in dump.items() if tid == c.tid and c.beg <= pos < c.end)
            sites = np.array([[pos, v[2], v[3], (v[0] << 1) | v[1]] for pos, v in rows], dtype=np.uint32).reshape(-1, 4)
            var = np.array([[v[4], v[5]] for pos, v in rows], dtype=np.uint32).reshape(-1, 2) if variant else None
            return sites, var
        return count

    n = multi.extract_sharded(args, factory)
    ret[rank] = n
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("extra", [["--chunkSize", "4000"], ["--CHG", "--mergeContext", "--chunkSize", "2500"],
                                   ["--minOppositeDepth", "2", "--maxVariantFrac", "0.4", "--chunkSize", "7001", "--CHH"],
                                   ["-l", "BED", "--keepStrand", "--chunkSize", "3000", "--cytosine_report"]],
                         ids=["cpg", "merge", "variant", "bed"])
def test_world2_gloo_matches_single_process(tmp_path, small_synth, extra):
    if "BED" in extra:      # -l: chunks no region touches are passed over by every rank (nothing packed, nothing emitted)
        from bedgen import random_bed
        extra = [str(random_bed(tmp_path / "r.bed", [("chrS1", 40000), ("chrS2", 20000)], n=20, seed=41)) if x == "BED" else x for x in extra]
    base = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam")] + extra
    od = tmp_path / "oracle"; od.mkdir()
    dump = tmp_path / "dump.tsv"
    r = run_oracle(base + ["-o", "out"], cwd=od, dump=dump)
    assert r.returncode == 0
    gd = tmp_path / "sharded"; gd.mkdir()
    cwd = os.getcwd()
    os.chdir(gd)
    try:
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), base + ["-o", "out"], str(dump), ret), nprocs=2, join=True)
    finally:
        os.chdir(cwd)
    assert ret[0] > 0 and ret[1] > 0 and (abs(ret[0] - ret[1]) <= 1 or "-l" in extra), "both ranks must have counted about half of the chunks"
    seen = 0
    for f in os.listdir(od):
        if f.startswith("out"):
            seen += 1
            assert filecmp.cmp(od / f, gd / f, shallow=False), f
    assert seen > 0

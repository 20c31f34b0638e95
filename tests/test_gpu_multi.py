"""GPU: the command as one process per GPU (csrc/host/mdk_ranks.c) and the exchange layer (csrc/mdk_comm.hip).

`MethylDackel extract` started N times with MDK_RANK/MDK_WORLD (or by torchrun) shards the chunk schedule -- chunk k on rank
k mod N -- and rank 0 collects every chunk and writes.  On this box all ranks sit on the same physical GPU, which RCCL
refuses; rank 0 sees the equal PCI bus ids during the bootstrap and the site buffers travel over the ranks' TCP connections
instead of ncclSend/ncclRecv.  Everything else -- the schedule, two slots per rank, the ring on rank 0, ordered emission -- is
the multi-GPU code path.  RCCL itself is exercised with the one communicator a single GPU allows (world size 1) and, when the
box has two GPUs, by tests/test_gpu_bench.py."""
import ctypes as C

import pytest

import methyldackel_amd as mdk
from conftest import synth
from test_gpu_parity import compare_cli

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("extra", [["--chunkSize", "3000"], ["--CHG", "--CHH", "--mergeContext", "--chunkSize", "5000", "--minOppositeDepth", "2", "--maxVariantFrac", "0.4"]], ids=["cpg", "allctx_merge_variant"])
def test_sharded_command_byte_exact(tmp_path, small_synth, n, extra):
    compare_cli(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam")] + extra, ranks=n, env={"MDK_DEVICE": "0"})


def test_sharded_command_with_host_fallback_and_empty_chunks(tmp_path):
    """a name with 40 records (the device hands the chunk back) and contigs without reads, across 2 ranks"""
    from bamwriter import record, write_bam, write_fasta
    ref = ("ACGTCGCGTTCGAACGCGTA" * 60)[:1100]
    recs = [record(1, 10 + 10 * k, 99 if k % 2 == 0 else 147, "60M", ref[10 + 10 * k:70 + 10 * k], 35, qname="same", mpos=20 + 10 * k) for k in range(40)]
    recs += [record(1, 600 + 3 * k, 0, "50M", ref[600 + 3 * k:650 + 3 * k], 30, qname=f"u{k}") for k in range(60)]
    write_bam(tmp_path / "m.bam", [("empty1", 500), ("c1", len(ref)), ("empty2", 300)], recs)
    write_fasta(tmp_path / "m.fa", [("empty1", "ACGT" * 125), ("c1", ref), ("empty2", "CG" * 150)])
    compare_cli(tmp_path, [str(tmp_path / "m.fa"), str(tmp_path / "m.bam"), "-F", "0", "-q", "0", "--keepDupes", "--chunkSize", "400"], ranks=2, env={"MDK_DEVICE": "0"})


def test_unindexed_bam_and_region(tmp_path, small_synth):
    """without a .bai every rank reads the stream through and keeps its own chunks; with -r the schedule starts inside a contig"""
    import shutil
    shutil.copy(small_synth / "pe.fa", tmp_path / "u.fa"); shutil.copy(small_synth / "pe.bam", tmp_path / "u.bam")
    compare_cli(tmp_path, [str(tmp_path / "u.fa"), str(tmp_path / "u.bam"), "--chunkSize", "2500"], ranks=3, env={"MDK_DEVICE": "0"})
    compare_cli(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--chunkSize", "2500", "-r", "chrS1:5000-30000"], ranks=2, env={"MDK_DEVICE": "0"})


def test_a_rank_that_cannot_start_ends_the_others(tmp_path, small_synth):
    """rank 1 is given a device that does not exist: it leaves with the no-device code and rank 0, waiting for it, gives up too"""
    r = mdk.run_ranks([str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-o", "x"], 2, cwd=tmp_path, devices=[0, 99], timeout=120)
    assert r.rank_returncodes[1] != 0 and r.rank_returncodes[0] != 0, r.rank_stderr
    assert "cannot open MI355X device 99" in r.rank_stderr[1]


def test_rccl_single_rank_communicator(tmp_path, small_synth):
    """librccl is loaded on demand, an id is made, a world-1 communicator initialised on the device handle; the exchange of a
    one-rank world has no peer to talk to and completes"""
    L = mdk.lib_hip()
    plan = mdk.Plan([str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--chunkSize", "20000", "-o", str(tmp_path / "x")])
    dev = mdk.Device(plan.dev_cfg())
    idb = C.create_string_buffer(mdk.COMM_ID_BYTES)
    assert L.md_comm_unique_id(idb) == 0, L.md_dev_last_error()
    assert any(idb.raw)
    comm = C.c_void_p()
    assert L.md_comm_open_rank(dev.h, 0, 1, idb.raw, C.byref(comm)) == 0, L.md_dev_last_error()
    assert L.md_comm_world(comm) == 1
    c = plan.next_chunk(); plan.ensure_reference(dev, c.tid)
    dev.submit(0, c.batch)
    dv = dev.wait(0)
    snd = (C.c_void_p * 1)(dv.d_site); sb = (C.c_uint64 * 1)(dv.n_slots * 16)
    rcv = (C.c_void_p * 1)(None); rb = (C.c_uint64 * 1)(0)
    assert L.md_comm_gather(comm, snd, sb, rcv, rb) == 0, L.md_dev_last_error()
    assert L.md_comm_wait(comm) == 0
    # the benchmark loop over this communicator: the ranks agree on the size of an exchanged region with an ncclAllReduce
    # (max of their site / tile counts) before any send or receive is posted; with one rank the agreement is its own size
    c2 = plan.next_chunk(); dev.submit(1, c2.batch); dev.wait(1)
    slots = (C.c_int * 2)(0, 1); bench = C.c_void_p()
    assert L.md_bench_open(dev.h, comm, slots, 2, 1, C.byref(bench)) == 0, L.md_dev_last_error()
    region = L.md_bench_region_bytes(bench)
    assert region >= 16 * max(dv.n_slots, 1) and region % 256 == 0
    res = mdk.md_bench_run_result()
    assert L.md_bench_run(bench, 6, C.byref(res)) == 0, L.md_dev_last_error()
    assert res.launches == 6 and L.md_bench_verify(bench) == 0, L.md_dev_last_error()
    L.md_bench_close(bench)
    L.md_comm_close(comm)
    dev.close(); plan.close()

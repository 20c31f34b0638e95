"""GPU parity for `perRead`: `MethylDackel perRead` (k_perread, one lane per read) against the oracle's restatement of
perRead.c -- output text byte-identical; and the kernel at the C-ABI against the slow Python walk."""
import ctypes as C

import pytest

import methyldackel_amd as mdk
from batch_eval import eval_perread
from conftest import GOLDEN, synth
from test_host_logic import read_fasta
from test_perread import FIX, SYN, oracle_perread

pytestmark = pytest.mark.gpu


def compare(tmp_path, args, to_file=True):
    od, gd = tmp_path / "oracle", tmp_path / "gpu"
    od.mkdir(exist_ok=True), gd.mkdir(exist_ok=True)
    tail = ["-o", "out.txt"] if to_file else []
    ro = oracle_perread(list(args) + tail, cwd=od)
    rg = mdk.run_cli(list(args) + tail, cwd=gd, command="perRead")
    assert rg.returncode == ro.returncode, (rg.returncode, ro.returncode, rg.stderr[-2000:])
    if to_file:
        assert (gd / "out.txt").read_bytes() == (od / "out.txt").read_bytes()
    assert rg.stdout == ro.stdout
    return ro


@pytest.mark.parametrize("args", FIX, ids=[" ".join(a[1:]).replace(str(GOLDEN) + "/", "") for a in FIX])
def test_cli_fixtures(tmp_path, args):
    compare(tmp_path, args)


@pytest.mark.parametrize("which,extra", SYN, ids=[f"{w}:{' '.join(e)}" for w, e in SYN])
def test_cli_synthetic(tmp_path, small_synth, which, extra):
    compare(tmp_path, [str(small_synth / f"{which}.fa"), str(small_synth / f"{which}.bam")] + extra)


def test_cli_to_stdout_with_threads_and_bed(tmp_path, small_synth):
    bed = tmp_path / "b.bed"
    bed.write_text("chrS1\t5000\t5100\nchrS2\t100\t200\n")
    ro = compare(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed), "--chunkSize", "2000", "-@", "4", "-p", "10"], to_file=False)
    assert len(ro.stdout.splitlines()) > 20


def test_cli_contig_missing_from_fasta(tmp_path, small_synth):
    fa = tmp_path / "one.fa"
    txt = (small_synth / "pe.fa").read_text()
    fa.write_text(txt[: txt.index(">", 1)])
    compare(tmp_path, [str(fa), str(small_synth / "pe.bam")])


def test_abi_counts_equal_python_walk(tmp_path, small_synth):
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-p", "25", "--chunkSize", "9000", "-o", str(tmp_path / "x")]
    plan = mdk.Plan(args, command="perRead")
    cfg = plan.dev_cfg()
    dev = mdk.Device(cfg)
    n = 0
    while (c := plan.next_chunk()) is not None:
        if c.skipped or not c.pr.n_reads:
            continue
        plan.ensure_reference(dev, c.tid)
        got = dev.perread(0, c.pr)
        assert got == eval_perread(c.pr, read_fasta(args)[plan.target_name(c.tid)], cfg.minPhred)
        n += len(got)
    assert n > 1000
    dev.close(); plan.close()


def test_s1_perread(tmp_path):
    synth(tmp_path / "S1", "-L", "1000000", "-c", "30", "-s", "0x5EED0001")
    compare(tmp_path, [str(tmp_path / "S1.fa"), str(tmp_path / "S1.bam"), "-@", "8"])
    compare(tmp_path, [str(tmp_path / "S1.fa"), str(tmp_path / "S1.bam"), "-@", "8", "-p", "20", "-q", "0"])


@pytest.mark.parametrize("extra", [["-p", "25", "--chunkSize", "9000"], ["-q", "0", "-F", "16", "-R", "1", "--chunkSize", "3000"], ["-p", "1", "-q", "40"]], ids=["p25", "flags", "p1_q40"])
def test_abi_device_selected_reads_equal_host_selected(tmp_path, small_synth, extra):
    """md_dev_perread_submit_raw (the device selects the reads that start in the chunk and pass -R/-F/-q, and walks them where they
    lie in the BAM records) against md_dev_perread_submit of the host-selected reads: same reads in the same order (checked by
    position), same counts"""
    import struct
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam")] + extra + ["-o", str(tmp_path / "x")]
    ph, pd = mdk.Plan(args, command="perRead"), mdk.Plan(args, command="perRead")
    pd.set_prep(1)
    dev = mdk.Device(ph.dev_cfg()); dev.set_prep(pd.prep_cfg())
    assert pd.prep_cfg().perread == 1
    n = 0
    while True:
        ch, cd = ph.next_chunk(), pd.next_chunk()
        assert (ch is None) == (cd is None)
        if ch is None:
            break
        if ch.skipped:
            continue
        ph.ensure_reference(dev, ch.tid)
        want = dev.perread(0, ch.pr) if ch.pr.n_reads else []
        kept, got = dev.perread_raw(1, cd.raw)
        assert got == want, ch.index
        cat = b"".join(C.string_at(cd.raw.range[i].ptr, cd.raw.range[i].bytes) for i in range(cd.raw.n_ranges))
        pos = [struct.unpack_from("<i", cat, cd.raw.rec_off[k] + 8)[0] for k in kept]
        assert pos == [ch.pr.read[i].pos for i in range(ch.pr.n_reads)], ch.index
        n += len(got)
    assert n > 500
    dev.close(); ph.close(); pd.close()


def test_cli_host_prep_mode(tmp_path, small_synth):
    import os
    od, gd = tmp_path / "oracle", tmp_path / "gpu"
    od.mkdir(), gd.mkdir()
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--chunkSize", "6000", "-o", "out.txt"]
    ro = oracle_perread(args, cwd=od)
    rg = mdk.run_cli(args, cwd=gd, command="perRead", env={"MDK_HOST_PREP": "1"})
    assert rg.returncode == ro.returncode == 0
    assert (gd / "out.txt").read_bytes() == (od / "out.txt").read_bytes()

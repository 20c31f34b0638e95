"""GPU parity for `perRead`: `MethylDackel perRead` (k_perread, one lane per read) against the oracle's restatement of
perRead.c -- output text byte-identical; and the kernel at the C-ABI against the slow Python walk."""
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

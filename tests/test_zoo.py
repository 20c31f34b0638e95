"""The hand-made corner-case BAM (tests/zoo.py) through `perRead`, `mbias` and `extract`: host side on CPU, the commands on
the GPU.  These are the places where the oracle had to give the reference's out-of-bounds reads a defined meaning."""
import pytest

import methyldackel_amd as mdk
import zoo
from test_mbias import check as mbias_check
from test_perread import check as perread_check
from test_host_logic import check as extract_check

PR = [[], ["-p", "10"], ["-p", "30", "--chunkSize", "97"], ["-q", "0", "-F", "4"], ["-p", "38"]]
MB = [[], ["--CHG", "--CHH"], ["--CHH", "--chunkSize", "53", "--nOT", "1,2,3,4", "--nOB", "0,0,1,0"], ["-p", "30", "--keepSingleton", "--keepDiscordant", "-F", "0"]]
EX = [["-q", "0"], ["--CHG", "--CHH", "--chunkSize", "211", "-p", "1"], ["--keepSingleton", "--keepDiscordant", "--ignoreFlags", "0", "--minOppositeDepth", "1", "--maxVariantFrac", "0.2"]]


@pytest.mark.parametrize("extra", PR, ids=[" ".join(e) or "defaults" for e in PR])
def test_perread_host(tmp_path, extra):
    fa, bam = zoo.build(tmp_path)
    text = perread_check(tmp_path, [fa, bam] + extra)
    assert "nocigar\tz1\t800\t0.0\t0" in text and "noseq\tz1\t700\t0.0\t0" in text


@pytest.mark.parametrize("extra", MB, ids=[" ".join(e) or "defaults" for e in MB])
def test_mbias_host(tmp_path, extra):
    fa, bam = zoo.build(tmp_path)
    n, want = mbias_check(tmp_path, [fa, bam] + extra)
    assert max(q for (_, _, q) in want) >= 550          # the 600-base read: rows beyond what a workgroup keeps in LDS


@pytest.mark.parametrize("extra", EX, ids=[" ".join(e) for e in EX])
def test_extract_host(tmp_path, extra):
    fa, bam = zoo.build(tmp_path)
    extract_check(tmp_path, [fa, bam] + extra, variant="--minOppositeDepth" in extra)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", PR, ids=[" ".join(e) or "defaults" for e in PR])
def test_perread_gpu(tmp_path, extra):
    from test_gpu_perread import compare
    fa, bam = zoo.build(tmp_path)
    compare(tmp_path, [fa, bam] + extra)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", MB, ids=[" ".join(e) or "defaults" for e in MB])
def test_mbias_gpu(tmp_path, extra):
    from test_gpu_mbias import compare_mbias
    fa, bam = zoo.build(tmp_path)
    compare_mbias(tmp_path, [fa, bam] + extra)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", EX, ids=[" ".join(e) for e in EX])
def test_extract_gpu(tmp_path, extra):
    from test_gpu_parity import compare_cli
    fa, bam = zoo.build(tmp_path)
    compare_cli(tmp_path, [fa, bam] + extra)


@pytest.mark.gpu
def test_mbias_undeterminable_strand_aborts(tmp_path):
    """a paired read with neither 0x40 nor 0x80 reaching a G position: updateMetrics aborts (common.c:122-125) in mbias too"""
    from bamwriter import record, write_bam, write_fasta
    from test_mbias import oracle_mbias
    ref = "ACGTCGCGCGTTTTCGCGCGAAAACGCG" * 4
    write_fasta(tmp_path / "e.fa", [("c", ref)])
    write_bam(tmp_path / "e.bam", [("c", len(ref))], [record(0, 2, 1 | 2, "40M", ref[2:42], 40, qname="s0", mpos=2)])
    ro = oracle_mbias([tmp_path / "e.fa", tmp_path / "e.bam", "--noSVG"], cwd=tmp_path)
    rg = mdk.run_cli([tmp_path / "e.fa", tmp_path / "e.bam", "--noSVG"], cwd=tmp_path, command="mbias")
    assert ro.returncode == -6 and rg.returncode == -6, (ro.returncode, rg.returncode)
    assert "Can't determine the strand of a read!" in ro.stderr and "Can't determine the strand of a read!" in rg.stderr

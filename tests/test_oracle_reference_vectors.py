"""Pins the CPU oracle against every expectation the reference's own test-suite holds for `extract`
(reference tests/test.py:17-145: 15 CLI runs on the fixture BAMs, asserting output line counts).  The fixture
BAM/FASTA files under tests/golden/ are the reference's data files, byte for byte."""
import pytest

from conftest import GOLDEN, run_oracle

# (reference test.py lines, args, {suffix: expected line count}, stdout must contain)
CASES = [
    ("t1:17-22", ["ct100.fa", "ct_aln.bam", "-q", "2"], {"_CpG.bedGraph": 1}),
    ("t2:24-31", ["cg100.fa", "cg_aln.bam", "-q", "2"], {"_CpG.bedGraph": ">1"}),
    ("t3:33-39", ["cg100.fa", "cg_aln.bam", "-q", "10"], {"_CpG.bedGraph": 1}),
    ("t4:41-56", ["--methylKit", "--CHH", "--CHG", "cg100.fa", "cg_aln.bam", "-q", "2"], {"_CpG.methylKit": ">1", "_CHG.methylKit": 1, "_CHH.methylKit": 2}),
    ("t5:58-64", ["--minDepth", "2", "cg100.fa", "cg_aln.bam", "-q", "2"], {"_CpG.bedGraph": 1}),
    ("t6:66-72", ["--ignoreFlags", "0xD00", "cg100.fa", "cg_aln.bam", "-q", "2"], {"_CpG.bedGraph": 49}),
    ("t7:74-80", ["--requireFlags", "0xD00", "cg100.fa", "cg_aln.bam", "-q", "2"], {"_CpG.bedGraph": 49}),
    ("t9:90-96", ["-p", "1", "-q", "0", "--minOppositeDepth", "3", "--maxVariantFrac", "0.25", "cg100.fa", "cg_with_variants.bam"], {"_CpG.bedGraph": 48}),
    ("t10:98-105", ["chgchh.fa", "chgchh_aln.bam"], {"_CpG.bedGraph": 2}),
    ("t11:107-113", ["-q", "5", "chgchh.fa", "chgchh_aln.bam"], {"_CpG.bedGraph": 3}),
    ("t12:115-121", ["-q", "5", "--minConversionEfficiency", "0.9", "chgchh.fa", "chgchh_aln.bam"], {"_CpG.bedGraph": 2}),
    ("t13:123-129", ["-q", "5", "--minConversionEfficiency", "1.0", "chgchh.fa", "chgchh_aln.bam"], {"_CpG.bedGraph": 1}),
    ("t14:131-137", ["-q", "1", "cg100.fa", "NH.bam"], {"_CpG.bedGraph": 1}),
    ("t15:139-145", ["--ignoreNH", "-q", "1", "cg100.fa", "NH.bam"], {"_CpG.bedGraph": 49}),
]
REF_FILES = {"ct100.fa", "cg100.fa", "chgchh.fa", "ct_aln.bam", "cg_aln.bam", "cg_with_variants.bam", "chgchh_aln.bam", "NH.bam"}


def resolve(args):
    return [str(GOLDEN / a) if a in REF_FILES else a for a in args]


def count_lines(p):
    return sum(1 for _ in open(p))


@pytest.mark.parametrize("name,args,expect", CASES, ids=[c[0] for c in CASES])
def test_reference_expectation(tmp_path, name, args, expect):
    r = run_oracle(resolve(args) + ["-o", tmp_path / "t"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    for suffix, want in expect.items():
        n = count_lines(str(tmp_path / "t") + suffix)
        if want == ">1":
            assert n > 1
        else:
            assert n == want, f"{name}{suffix}: {n} lines, reference asserts {want}"


def test_t2_content_known_answer(tmp_path):
    """Known answer derived by hand from the fixture: every even position 0..96 except 16
    (read 1 has a T there, read 2 a C, equal quals -> both zeroed by the overlap rule, overlaps.c:97-100)."""
    run_oracle(resolve(["cg100.fa", "cg_aln.bam", "-q", "2"]) + ["-o", tmp_path / "t"], cwd=tmp_path)
    lines = open(tmp_path / "t_CpG.bedGraph").read().splitlines()
    assert lines[0] == f'track type="bedGraph" description="{tmp_path / "t"} CpG methylation levels"'
    want = [f"chrCG\t{p}\t{p + 1}\t100\t1\t0" for p in range(0, 98, 2) if p != 16]
    assert lines[1:] == want


def test_t9_variant_line(tmp_path):
    r = run_oracle(resolve(["-p", "1", "-q", "0", "--minOppositeDepth", "3", "--maxVariantFrac", "0.25", "cg100.fa", "cg_with_variants.bam"]) + ["-o", tmp_path / "t"], cwd=tmp_path)
    assert r.stdout == "1 positions were excluded due to likely being variants.\n"


@pytest.mark.xfail(strict=True, reason="reference tests/test.py:82-88 asserts 12 lines for --nOT 50,50,40,40; executing "
                   "common.c:174-208 + overlaps.c:54-119 by hand (and this oracle) gives 11 (10 calls: C at 40..58). "
                   "Unresolved without a reference binary; documented in DESIGN.md")
def test_t8_reference_expectation_unresolved(tmp_path):
    run_oracle(resolve(["--nOT", "50,50,40,40", "cg100.fa", "cg_aln.bam", "-q", "2"]) + ["-o", tmp_path / "t"], cwd=tmp_path)
    assert count_lines(tmp_path / "t_CpG.bedGraph") == 12


def test_t8_by_the_code(tmp_path):
    run_oracle(resolve(["--nOT", "50,50,40,40", "cg100.fa", "cg_aln.bam", "-q", "2"]) + ["-o", tmp_path / "t"], cwd=tmp_path)
    lines = open(tmp_path / "t_CpG.bedGraph").read().splitlines()[1:]
    assert lines == [f"chrCG\t{p}\t{p + 1}\t100\t1\t0" for p in range(40, 60, 2)]

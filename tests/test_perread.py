"""CPU-only: the `perRead` command (perRead.c).  No expectation in the reference's own tests: parity rests on the oracle's
restatement ("parity unpinned" by the reference).  The perRead plan's chunks (schedule without adjustBounds, reads that
start in the chunk, -F/-R/-q) are evaluated by tests/batch_eval.py and written through mdk_plan_emit_perread; the text
must equal the oracle's, byte for byte."""
import subprocess
import sys

import pytest

import methyldackel_amd as mdk
from batch_eval import eval_perread
from bedgen import random_bed
from conftest import GOLDEN, ORACLE, REPO
from test_host_logic import read_fasta


def oracle_perread(args, cwd):
    return subprocess.run([str(ORACLE), "perRead"] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True)


DRIVER = """
import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)
import methyldackel_amd as mdk
from batch_eval import eval_perread
from test_host_logic import read_fasta
args = sys.argv[1:]
plan = mdk.Plan(args, command="perRead")
cfg = plan.dev_cfg()
while (c := plan.next_chunk()) is not None:
    if c.skipped & mdk.CHUNK_NOREF:
        plan.emit_perread(c, None)
    elif c.skipped:
        plan.emit_perread(c, [])
    else:
        ref = read_fasta(args)[plan.target_name(c.tid)]
        plan.emit_perread(c, eval_perread(c.pr, ref, cfg.minPhred))
plan.close()
""" % (str(REPO), str(REPO / "tests"))


def host_perread(args, cwd):
    return subprocess.run([sys.executable, "-c", DRIVER] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True)


def check(tmp_path, args):
    o = oracle_perread(list(args) + ["-o", tmp_path / "o.txt"], cwd=tmp_path)
    assert o.returncode == 0, o.stderr
    g = host_perread(list(args) + ["-o", tmp_path / "g.txt"], cwd=tmp_path)
    assert g.returncode == 0, g.stderr
    want, got = (tmp_path / "o.txt").read_text(), (tmp_path / "g.txt").read_text()
    assert got == want
    return want


def G(*n):
    return [str(GOLDEN / x) for x in n]


FIX = [G("cg100.fa", "cg_aln.bam") + ["-q", "2"], G("cg100.fa", "cg_aln.bam") + ["-q", "2", "-p", "30"], G("ct100.fa", "ct_aln.bam") + ["-q", "0"],
       G("chgchh.fa", "chgchh_aln.bam") + ["-q", "5", "-F", "256"], G("cg100.fa", "NH.bam") + ["-q", "1", "-R", "1"], G("cg100.fa", "cg_with_variants.bam") + ["-q", "0", "-p", "1"]]


@pytest.mark.parametrize("args", FIX, ids=[" ".join(a[1:]).replace(str(GOLDEN) + "/", "") for a in FIX])
def test_fixtures(tmp_path, args):
    assert check(tmp_path, args)


SYN = [
    ("pe", []),
    ("pe", ["-p", "20"]),                                       # many bases below -p: the walk's skip-and-evaluate-the-next behaviour
    ("pe", ["-p", "38", "--chunkSize", "700"]),
    ("pe", ["-q", "0", "-F", "3840", "-R", "3", "-r", "chrS1:3000-20000"]),
    ("bis", ["-p", "13", "--chunkSize", "4000"]),
    ("se", ["-p", "24"]),
]


@pytest.mark.parametrize("which,extra", SYN, ids=[f"{w}:{' '.join(e)}" for w, e in SYN])
def test_synthetic(tmp_path, small_synth, which, extra):
    text = check(tmp_path, [small_synth / f"{which}.fa", small_synth / f"{which}.bam"] + extra)
    assert len(text.splitlines()) > 100


def test_stdout_and_bed_chunk_skipping(tmp_path, small_synth):
    """-l only passes over whole chunks (perRead.c:150-166): reads of a kept chunk are listed wherever they lie"""
    bed = tmp_path / "b.bed"
    bed.write_text("chrS1\t5000\t5100\nchrS2\t100\t200\n")
    args = [small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed, "--chunkSize", "2000"]
    o = oracle_perread(args, cwd=tmp_path)
    g = host_perread(args, cwd=tmp_path)
    assert o.returncode == 0 and g.returncode == 0, g.stderr
    assert g.stdout == o.stdout and 20 < len(o.stdout.splitlines()) < 2000


def test_contig_missing_from_fasta_lists_reads_with_zero_calls(tmp_path, small_synth):
    fa = tmp_path / "one.fa"
    txt = (small_synth / "pe.fa").read_text()
    fa.write_text(txt[: txt.index(">", 1)])                      # chrS1 only
    want = check(tmp_path, [fa, small_synth / "pe.bam"])
    assert any(l.split("\t")[1] == "chrS2" and l.endswith("\t0.0\t0") for l in want.splitlines())


BAD = [([], 0), ([GOLDEN / "cg100.fa"], 255), ([GOLDEN / "nope.fa", GOLDEN / "cg_aln.bam"], 254), ([GOLDEN / "cg100.fa", GOLDEN / "nope.bam"], 252),
       ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "--chunkSize", "0"], 1), ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "--ignoreNH"], 1),
       ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "-r", "nope"], 250), ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "-o", "/nonexistent/dir/x"], 2)]


@pytest.mark.parametrize("args,rc", BAD, ids=[" ".join(str(a).replace(str(GOLDEN) + "/", "") for a in b[0]) or "no arguments" for b in BAD])
def test_option_errors_match_the_oracle(tmp_path, args, rc):
    o = oracle_perread(args, cwd=tmp_path)
    assert o.returncode == rc
    code = ("import sys; sys.path.insert(0, %r); import methyldackel_amd as mdk\n"
            "try:\n    mdk.Plan(sys.argv[1:], command='perRead'); print('opened')\nexcept mdk.MdkError as e:\n    print(e)\n" % str(REPO))
    g = subprocess.run([sys.executable, "-c", code] + [str(a) for a in args], cwd=tmp_path, capture_output=True, text=True)
    want = rc if rc < 128 else rc - 256
    assert f"returned {want}" in g.stdout
    def message(s):      # what precedes the usage text (the usage texts differ by design)
        out = []
        for l in s.splitlines():
            if l.startswith("Usage:"):
                break
            if l.strip():
                out.append(l)
        return out
    assert message(g.stderr) == message(o.stderr)

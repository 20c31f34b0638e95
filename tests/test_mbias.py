"""CPU-only: the `mbias` command (MBias.c + svg.c).  The reference's tests hold no expectation for mbias, so parity here
is pinned on the oracle's restatement alone ("parity unpinned" by the reference).  Checked:
  * host side: the batches an mbias plan hands out, evaluated by tests/batch_eval.py, reproduce the oracle's --txt table
    (admission, unpaired segments, window-relative contexts at chunk edges, BED, --nOT trimming);
  * output side: mdk_mbias_report over the oracle's histogram writes byte-identical SVGs, table and suggestion line."""
import filecmp
import subprocess
import sys

import numpy as np
import pytest

import methyldackel_amd as mdk
from batch_eval import eval_mbias
from bedgen import random_bed
from conftest import GOLDEN, ORACLE, REPO
from test_host_logic import read_fasta

STRANDS = {"OT": 1, "OB": 2, "CTOT": 3, "CTOB": 4}


def oracle_mbias(args, cwd):
    return subprocess.run([str(ORACLE), "mbias"] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True)


def parse_txt(text):
    rows = {}
    lines = text.splitlines()
    assert lines[0] == "Strand\tRead\tPosition\tnMethylated\tnUnmethylated"
    for l in lines[1:]:
        s, r, q, m, u = l.split("\t")
        rows[(STRANDS[s], int(r), int(q) - 1)] = [int(m), int(u)]
    return rows


def host_hist(args):
    plan = mdk.Plan(args, command="mbias")
    cfg = plan.dev_cfg()
    hist, nchunks = {}, 0
    while (c := plan.next_chunk()) is not None:
        nchunks += 1
        if c.skipped:
            continue
        ref = read_fasta(args)[plan.target_name(c.tid)]
        eval_mbias(c.batch, ref, cfg, plan.regions(c.tid), hist)
    plan.close()
    return hist, nchunks


def check(tmp_path, args):
    o = oracle_mbias(list(args) + ["--noSVG"], cwd=tmp_path)
    assert o.returncode == 0, o.stderr
    want = parse_txt(o.stdout)
    got, n = host_hist(list(args) + ["--noSVG"])
    assert got == want
    return n, want


def G(*n):
    return [str(GOLDEN / x) for x in n]


FIX = [
    G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    G("cg100.fa", "cg_aln.bam") + ["-q", "2", "--CHG", "--CHH", "--noCpG"],
    G("cg100.fa", "cg_aln.bam") + ["-q", "2", "--nOT", "10,10,20,20", "--nOB", "5,0,0,7"],
    G("chgchh.fa", "chgchh_aln.bam") + ["-q", "5", "--CHG", "--CHH"],
    G("chgchh.fa", "chgchh_aln.bam") + ["-q", "5", "--CHH", "--minConversionEfficiency", "0.9"],
    G("ct100.fa", "ct_aln.bam") + ["-q", "2", "--CHH"],
    G("cg100.fa", "NH.bam") + ["-q", "1", "--ignoreNH"],
]


@pytest.mark.parametrize("args", FIX, ids=[" ".join(a[1:]).replace(str(GOLDEN) + "/", "") for a in FIX])
def test_fixture_histograms(tmp_path, args):
    check(tmp_path, args)


SYN = [
    ("pe", []),
    ("pe", ["--CHG", "--CHH", "--chunkSize", "997"]),                      # many chunk edges: window-relative contexts
    ("pe", ["--CHG", "--noCpG", "--chunkSize", "333", "-r", "chrS1:2000-9000"]),
    ("pe", ["--keepDupes", "--keepSingleton", "--keepDiscordant", "-F", "0", "-q", "0", "-p", "1"]),
    ("pe", ["--nOT", "3,4,5,6", "--nOB", "7,8,9,10", "--requireFlags", "2"]),
    ("bis", ["--CHH", "--nCTOT", "5,5,5,5", "--nCTOB", "2,0,0,9", "--chunkSize", "5000"]),
    ("se", ["--CHG"]),
]


@pytest.mark.parametrize("which,extra", SYN, ids=[f"{w}:{' '.join(e)}" for w, e in SYN])
def test_synthetic_histograms(tmp_path, small_synth, which, extra):
    n, want = check(tmp_path, [small_synth / f"{which}.fa", small_synth / f"{which}.bam"] + extra)
    assert want and n >= 1


def test_histogram_with_bed(tmp_path, small_synth):
    bed = random_bed(tmp_path / "r.bed", [("chrS1", 40000), ("chrS2", 20000)], n=60, seed=51)
    check(tmp_path, [small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed, "--keepStrand", "--CHG", "--chunkSize", "2500"])


def _report(hist_rows, opref, svg, txt, which, cwd):
    """run mdk_mbias_report in a subprocess (it prints through C stdio)"""
    n = max((q for (_, _, q) in hist_rows), default=-1) + 1
    a = np.zeros((max(n, 1), 4, 2, 2), dtype=np.uint32)
    for (s, r, q), (m, u) in hist_rows.items():
        a[q, s - 1, r - 1] = (m, u)
    np.save(cwd / "hist.npy", a[:n] if n else a[:0])
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import methyldackel_amd as mdk\n"
            "rc = mdk.mbias_report(np.load('hist.npy'), %r, %d, %d, %d); sys.exit(rc & 255)\n" % (str(REPO), opref, svg, txt, which))
    return subprocess.run([sys.executable, "-c", code], cwd=cwd, capture_output=True, text=True)


REPORTS = [
    ("pe", [], 1), ("pe", ["--CHG", "--CHH"], 7), ("bis", ["--CHH", "--noCpG"], 4), ("se", ["--CHG", "-p", "20"], 3),
    ("pe", ["--noCpG", "--CHG", "-r", "chrS2:1-3000"], 2),
]


@pytest.mark.parametrize("which,extra,mask", REPORTS, ids=[f"{w}:{' '.join(e)}" for w, e, _ in REPORTS])
def test_report_is_byte_identical(tmp_path, small_synth, which, extra, mask):
    od, gd = tmp_path / "o", tmp_path / "g"
    od.mkdir(), gd.mkdir()
    o = oracle_mbias([small_synth / f"{which}.fa", small_synth / f"{which}.bam", "out", "--txt"] + extra, cwd=od)
    assert o.returncode == 0
    g = _report(parse_txt(o.stdout), "out", 1, 1, mask, gd)
    assert g.returncode == 0, g.stderr
    assert g.stdout == o.stdout
    assert g.stderr == o.stderr and "Suggested inclusion options:" in g.stderr
    svgs = sorted(f.name for f in od.iterdir() if f.suffix == ".svg")
    assert svgs and svgs == sorted(f.name for f in gd.iterdir() if f.suffix == ".svg")
    for f in svgs:
        assert filecmp.cmp(od / f, gd / f, shallow=False), f


def test_report_on_skewed_profiles(tmp_path):
    """hand-made histograms that trigger every branch of the bound suggestion: biased 5' end, biased 3' end, a read
    with no calls, a strand with calls only on read 2, power-of-two lengths"""
    rng = np.random.default_rng(5)
    for case, L in enumerate((32, 64, 100, 128, 151)):
        rows = {}
        for q in range(L):
            n = int(rng.integers(200, 400))
            f1 = 0.75 - (0.5 if q < 6 else 0) + (0.2 if q > L - 5 else 0)
            f2 = 0.75 + (0.2 if q < 3 else 0) - (0.6 if q > L - 9 else 0)
            rows[(1, 1, q)] = [int(n * f1), n - int(n * f1)]
            rows[(1, 2, q)] = [int(n * f2), n - int(n * f2)]
            if q % 3:
                rows[(2, 2, q)] = [int(n * 0.1), n - int(n * 0.1)]
            if q > 10:
                rows[(4, 1, q)] = [n, 0]
        d = tmp_path / f"c{case}"; d.mkdir()
        # the oracle's report, fed the same numbers through a tiny driver: build a --txt-like table and let both sides plot it
        want = subprocess.run([str(ORACLE), "mbias-report", "out", "7"], cwd=d, input=_table(rows), capture_output=True, text=True)
        assert want.returncode == 0, want.stderr
        (d / "g").mkdir()
        g = _report(rows, "out", 1, 1, 7, d / "g")
        assert g.returncode == 0
        assert g.stdout == want.stdout and g.stderr == want.stderr
        assert "--OT" in g.stderr and g.stderr.split("--OT ")[1].split()[0] != "0,0,0,0"
        for f in ("out_OT.svg", "out_OB.svg", "out_CTOB.svg"):
            assert filecmp.cmp(d / f, d / "g" / f, shallow=False), (case, f)
        assert not (d / "out_CTOT.svg").exists() and not (d / "g" / "out_CTOT.svg").exists()


def _table(rows):
    inv = {v: k for k, v in STRANDS.items()}
    out = ["Strand\tRead\tPosition\tnMethylated\tnUnmethylated"]
    for (s, r, q) in sorted(rows, key=lambda k: (k[0], k[2], k[1])):
        out.append(f"{inv[s]}\t{r}\t{q + 1}\t{rows[(s, r, q)][0]}\t{rows[(s, r, q)][1]}")
    return "\n".join(out) + "\n"


BAD = [
    (["--noSVG"], -1 & 255), ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam"], -1 & 255), ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "p", "--noCpG"], -1 & 255),
    ([GOLDEN / "cg100.fa", GOLDEN / "nope.bam", "p"], -4 & 255), ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "p", "--chunkSize", "0"], 1),
    ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "p", "-r", "chrNope"], -6 & 255), ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "p", "--bogus"], 1),
    ([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "p", "-R", "2"], 1),        # the short form is missing from mbias' option string (MBias.c:353)
]


@pytest.mark.parametrize("args,rc", BAD, ids=[" ".join(str(a).replace(str(GOLDEN) + "/", "") for a in b[0]) for b in BAD])
def test_option_errors_match_the_oracle(tmp_path, args, rc):
    o = oracle_mbias(args, cwd=tmp_path)
    assert o.returncode == rc
    code = ("import sys; sys.path.insert(0, %r); import methyldackel_amd as mdk\n"
            "try:\n    mdk.Plan(sys.argv[1:], command='mbias')\nexcept mdk.MdkError as e:\n    print(e)\n" % str(REPO))
    g = subprocess.run([sys.executable, "-c", code] + [str(a) for a in args], cwd=tmp_path, capture_output=True, text=True)
    want = rc if rc < 128 else rc - 256
    assert f"returned {want}" in g.stdout
    first = lambda s: [l for l in s.splitlines() if l.strip()][:1]       # usage texts differ by design; the message does not
    assert first(g.stderr) == first(o.stderr)


@pytest.mark.parametrize("extra", [[], ["--CHG", "--CHH"], ["--noCpG", "--CHH", "-q", "20", "-p", "12"]], ids=["cpg", "allctx", "chh_q20_p12"])
def test_mbias_totals_equal_extract_totals_single_end(tmp_path, extra):
    """Cross-check that does not rest on the mbias restatement alone: without mate-overlap handling (single-end reads, so
    extract's overlap rule never fires and mbias has none, MBias.c:158-161) every methylation call `extract` counts at a
    position is the same (read, base) event `mbias` files under the base's position in the read.  So the column sums of the
    mbias table must equal the count sums of extract's bedGraphs, per context set, for the same filters (one chunk, so
    mbias' window-relative contexts at chunk edges cannot differ)."""
    from conftest import run_oracle, synth
    synth(tmp_path / "se", "-L", "60000", "-c", "18", "-s", "31", "--single", "--extras")
    args = [tmp_path / "se.fa", tmp_path / "se.bam"] + extra
    o = oracle_mbias(args + ["--noSVG", "--txt", "o"], cwd=tmp_path)      # mbias --txt prints the table
    assert o.returncode == 0, o.stderr
    table = parse_txt(o.stdout)
    (tmp_path / "e").mkdir()
    e = run_oracle(args + ["-o", "x"], cwd=tmp_path / "e")
    assert e.returncode == 0, e.stderr
    meth = unmeth = 0
    for f in (tmp_path / "e").iterdir():
        for line in open(f):
            t = line.split("\t")
            if len(t) == 6:
                meth += int(t[4]); unmeth += int(t[5])
    assert (sum(v[0] for v in table.values()), sum(v[1] for v in table.values())) == (meth, unmeth)
    assert meth + unmeth > 5000

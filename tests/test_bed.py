"""CPU-only: -l FILE / --keepStrand (bed.c).  The oracle restates the reference's cursor walk over the sorted region list
literally; the product folds it into per-contig runs (mdk_plan_regions) that chunks, reads and positions are tested
against.  Everything here compares the two: parsing (including the reference's quirks), run construction, the
chunk/read prefilters, and -- through tests/batch_eval.py -- the per-position strand rule."""
import os
import subprocess
import sys

import pytest

import methyldackel_amd as mdk
from bedgen import random_bed
from conftest import REPO, run_oracle
from test_host_logic import check

PE = [("chrS1", 40000), ("chrS2", 20000)]

CASES = [
    # sample, contigs, bed kwargs, extra args
    ("pe", PE, dict(n=40, seed=1), []),
    ("pe", PE, dict(n=40, seed=1), ["--keepStrand"]),
    ("pe", PE, dict(n=150, seed=2, crlf=True), ["--keepStrand", "--CHG", "--CHH"]),
    ("pe", PE, dict(n=60, seed=3, gz=True), ["--keepStrand", "--chunkSize", "777"]),
    ("pe", PE, dict(n=25, seed=4, dense=True), ["--keepStrand", "--chunkSize", "1500", "--CHH", "--noCpG"]),
    ("pe", PE, dict(n=6, seed=5), ["--keepStrand", "-r", "chrS1:5000-30000"]),
    ("pe", PE, dict(n=80, seed=6), ["--keepStrand", "--OT", "5,90,5,90", "--nOB", "3,3,3,3", "-p", "15"]),
    ("bis", [("chrS1", 30000)], dict(n=50, seed=7), ["--keepStrand", "--CHG"]),
    ("se", [("chrS1", 20000)], dict(n=50, seed=8), ["--keepStrand"]),
]


@pytest.mark.parametrize("sample,contigs,bk,extra", CASES, ids=[f"{c[0]}:n{c[2]['n']}s{c[2]['seed']}:{' '.join(c[3])}" for c in CASES])
def test_bed_batches_match_oracle(tmp_path, small_synth, sample, contigs, bk, extra):
    bed = random_bed(tmp_path / ("r.bed.gz" if bk.get("gz") else "r.bed"), contigs, **bk)
    chunks = check(tmp_path, [small_synth / f"{sample}.fa", small_synth / f"{sample}.bam", "-l", bed] + extra)
    assert chunks


def test_bed_variant_counters(tmp_path, small_synth):
    bed = random_bed(tmp_path / "r.bed", PE, n=60, seed=9)
    check(tmp_path, [small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed, "--keepStrand", "--minOppositeDepth", "2"], variant=True)


def test_runs_are_the_union_of_the_regions(tmp_path, small_synth):
    bed = random_bed(tmp_path / "r.bed", PE, n=120, seed=10)
    plan = mdk.Plan([small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed, "-o", tmp_path / "x"])
    names = {n: i for i, (n, _) in enumerate(PE)}
    cover = [bytearray(L + 2) for _, L in PE]
    for line in open(bed):
        f = line.split()
        if not f or f[0] not in names:
            continue
        t, s, e = names[f[0]], max(int(f[1]), 0), min(int(f[2]), PE[names[f[0]]][1] + 1)
        cover[t][s:e] = b"\1" * (e - s)
    for t in range(2):
        runs = plan.regions(t)
        got = bytearray(PE[t][1] + 2)
        last = 0
        for s, e, st in runs:
            assert last <= s < e and st in (0, 1, 2)
            got[s:e] = b"\1" * (e - s)
            last = e
        assert got == cover[t]
    plan.close()
    # no -l: no restriction
    plan = mdk.Plan([small_synth / "pe.fa", small_synth / "pe.bam", "-o", tmp_path / "y"])
    assert plan.regions(0) is None
    plan.close()


def test_skipped_chunks_and_index_seek(tmp_path, small_synth, monkeypatch):
    """chunks no region touches are flagged CHUNK_BED (and, with a .bai, not read at all); what is packed for the
    others does not depend on the index"""
    bed = tmp_path / "two.bed"
    bed.write_text("chrS1\t2000\t2300\nchrS1\t31000\t31050\tx\t0\t-\nchrS2\t19990\t25000\n")
    args = [small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed, "--chunkSize", "1000", "--keepStrand", "-o", tmp_path / "o"]

    def walk():
        plan = mdk.Plan(args)
        out = []
        while (c := plan.next_chunk()) is not None:
            out.append((c.index, c.tid, c.beg, c.end, c.skipped, c.batch.n_reads if not c.skipped else 0,
                        bytes(C_blob(c)) if not c.skipped else b""))
        plan.close()
        return out

    import ctypes as C

    def C_blob(c):
        return C.string_at(c.batch.blob, c.batch.blob_bytes)

    a = walk()
    monkeypatch.setenv("MDK_NO_INDEX", "1")
    b = walk()
    assert a == b
    live = [x for x in a if not x[4]]
    assert all(x[4] == mdk.CHUNK_BED for x in a if x[4])
    assert 3 <= len(live) <= 6 and len(a) > 50
    assert all(x[5] > 0 for x in live)


BAD = [
    ("unknown contig", "chrS1\t10\t20\nchrNope\t5\t9\n", 1),
    ("start >= end", "chrS1\t30\t30\n", 1),
    ("no columns", "chrS1\n", 1),
    ("no end", "chrS1\t10\n", 1),
    ("text start", "chrS1\tabc\t10\n", 1),
    ("start -1", "chrS1\t-1\t10\n", 1),
    ("doubled separator", "chrS1\t\t10\t20\n", 1),
]


@pytest.mark.parametrize("name,text,rc", BAD, ids=[b[0] for b in BAD])
def test_bad_bed_is_rejected_like_the_reference(tmp_path, small_synth, name, text, rc):
    bed = tmp_path / "bad.bed"
    bed.write_text(text)
    args = [small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed]
    o = run_oracle(args + ["-o", tmp_path / "o"], cwd=tmp_path)
    assert o.returncode == rc
    code = ("import sys; sys.path.insert(0, %r); import methyldackel_amd as mdk\n"
            "try:\n    mdk.Plan(sys.argv[1:])\nexcept mdk.MdkError as e:\n    print(e)\n" % str(REPO))
    g = subprocess.run([sys.executable, "-c", code] + [str(a) for a in args] + ["-o", str(tmp_path / "g")], capture_output=True, text=True)
    assert f"mdk_plan_open returned {rc}" in g.stdout
    assert g.stderr.replace(str(tmp_path / "g"), "X") == o.stderr.replace(str(tmp_path / "o"), "X")


QUIRKS = [
    ("an empty line ends the file", "chrS1\t100\t900\n\nchrS1\t5000\t9000\n"),
    ("track and browser lines, comments", "#c\ntrack x\nbrowser y\nchrS1\t100\t900\n"),
    ("strand is the first character of column 6", "chrS1\t100\t900\tn\t0\t-x\nchrS1\t2000\t2900\tn\t0\tplus\nchrS1 3000 3900 n 0 +\n"),
    ("fewer columns leave the strand open", "chrS1\t100\t900\tn\nchrS1\t2000\t2900\tn\t0\nchrS1\t3000\t3900\tn\t0\t\n"),
    ("blanks before the end column shift the strand column", "chrS1\t100\t \t900\t+\t-\t+\n"),
    ("numbers with trailing text, signs", "chrS1\t+100x\t900y\tn\t0\t-\n"),
    ("clamping", "chrS1\t-5\t50\nchrS2\t19000\t999999\n"),
    ("nested regions: the first in sorted order governs", "chrS1\t100\t5000\tn\t0\t+\nchrS1\t200\t300\tn\t0\t-\nchrS1\t4000\t6000\tn\t0\t-\n"),
    ("last line without newline", "chrS1\t100\t900\tn\t0\t-"),
]


@pytest.mark.parametrize("name,text", QUIRKS, ids=[q[0] for q in QUIRKS])
def test_bed_parsing_quirks(tmp_path, small_synth, name, text):
    bed = tmp_path / "q.bed"
    bed.write_bytes(text.encode())
    check(tmp_path, [small_synth / "pe.fa", small_synth / "pe.bam", "-l", bed, "--keepStrand"])

"""GPU parity on hand-crafted adversarial BAMs (tests/bamwriter.py): cases the synthetic generator does not produce.
Every case runs the same command line through the oracle and the product and compares files, stdout and return code."""
import random

import pytest

from bamwriter import aux_Z, aux_i, record, write_bam, write_fasta
from test_gpu_parity import compare_cli

pytestmark = pytest.mark.gpu


def ref_with_cpgs(n, seed=1):
    r = random.Random(seed)
    s = []
    while len(s) < n:
        s += r.choice(["CG", "CG", "A", "T", "C", "G", "CAG", "CTG", "ACGT", "N", "c", "g", "cg"])
    return "".join(s)[:n]


def bs_read(ref, pos, length, strand_odd, rng, meth=0.6):
    out = []
    for i in range(length):
        b = ref[pos + i].upper()
        if strand_odd and b == "C" and rng.random() > meth:
            b = "T"
        if not strand_odd and b == "G" and rng.random() > meth:
            b = "A"
        if b not in "ACGT":
            b = "N"
        out.append(b)
    return "".join(out)


def test_quality_boost_wraps_above_213(tmp_path):
    """(uint8_t)(q + 0.2*q) overflows for q >= 214 (overlaps.c:103-106): identical bases, equal and unequal huge quals"""
    rng = random.Random(3)
    ref = ref_with_cpgs(400, 2)
    recs = []
    for k, (q1, q2) in enumerate([(214, 214), (255, 254), (213, 213), (230, 10), (5, 250), (214, 0), (127, 128)]):
        pos = 10 + 7 * k
        s = bs_read(ref, pos, 100, True, rng)
        recs.append((pos, record(0, pos, 99, "100M", s, q1, qname=f"p{k}", mpos=pos + 20)))
        s2 = bs_read(ref, pos + 20, 100, True, rng)
        recs.append((pos + 20, record(0, pos + 20, 147, "100M", s2, q2, qname=f"p{k}", mpos=pos)))
    recs.sort(key=lambda x: x[0])
    write_bam(tmp_path / "a.bam", [("c1", 400)], [r for _, r in recs])
    write_fasta(tmp_path / "a.fa", [("c1", ref)])
    for extra in ([], ["-p", "200"], ["--CHG", "--CHH", "-p", "1", "--minOppositeDepth", "1", "--maxVariantFrac", "0.1"]):
        compare_cli(tmp_path, [str(tmp_path / "a.fa"), str(tmp_path / "a.bam")] + extra)


def test_three_records_per_qname_across_chunks_and_zero_length_alignments(tmp_path):
    """pairing is a toggle among ADMITTED reads of one chunk, with htslib's buffer eviction (overlaps.c:121-147): primary +
    supplementary + mate, an all-soft-clip read and an unpaired read sharing the qname, chunk boundaries in between"""
    rng = random.Random(5)
    ref = ref_with_cpgs(1600, 4)
    R = []

    def add(pos, flag, cig, qname, l=None, q=30, mpos=0):
        c = cig
        import re
        ql = sum(int(n) for n, op in re.findall(r"(\d+)([MIS=X])", c))
        rl = sum(int(n) for n, op in re.findall(r"(\d+)([MDN=X])", c))
        s = bs_read(ref, pos, max(rl, 1), flag & 0x40 and not flag & 0x10 or (flag & 0x80 and flag & 0x10), rng) if rl else ""
        s = (s + "A" * ql)[:ql]
        R.append((pos, len(R), record(0, pos, flag, c, s, q, qname=qname, mpos=mpos)))

    add(100, 99, "80M", "x"); add(130, 2147 & 0xFFFF, "60M", "x"); add(150, 147, "80M", "x")      # primary, supplementary(0x800), mate
    add(300, 99, "50M", "y"); add(310, 99 | 0x100, "40M", "y"); add(320, 147, "50M", "y"); add(330, 147 | 0x800, "30M", "y")
    add(495, 99, "30M", "z"); add(505, 147, "30M", "z")                                             # straddles chunkSize 500 boundary
    add(600, 99, "40M", "w"); add(610, 73, "20S", "w"); add(615, 147, "40M", "w")                  # zero-length alignment in between
    add(700, 99, "40M", "v"); add(705, 0, "40M", "v"); add(710, 147, "40M", "v")                    # unpaired record with the same name
    add(800, 99, "10M500N10M", "u"); add(805, 147, "20M", "u"); add(1000, 99, "30M", "t"); add(1310 - 300, 147, "30M", "t")
    R.sort(key=lambda x: (x[0], x[1]))
    write_bam(tmp_path / "b.bam", [("c1", 1600)], [r for _, _, r in R])
    write_fasta(tmp_path / "b.fa", [("c1", ref)])
    for extra in (["-F", "0", "--keepSingleton", "--keepDiscordant", "--CHG", "--CHH", "-p", "1"],
                  ["-F", "0", "--keepSingleton", "--keepDiscordant", "--chunkSize", "500", "--CHH"],
                  ["-F", "0", "--keepSingleton", "--keepDiscordant", "--chunkSize", "37", "--mergeContext", "--CHG"],
                  []):
        compare_cli(tmp_path, [str(tmp_path / "b.fa"), str(tmp_path / "b.bam")] + extra)


def test_cigar_zoo_and_long_runs(tmp_path):
    """leading insertion, hard+soft clips, padding, =/X, deletion at the start of the overlap, a 70,000-base run (longer than a
    16-bit segment length), reads hanging over the contig end"""
    rng = random.Random(7)
    L = 80000
    ref = ref_with_cpgs(L, 6)
    R = []

    def add(pos, flag, cig, qname, mpos=0, q=35):
        import re
        ql = sum(int(n) for n, op in re.findall(r"(\d+)([MIS=X])", cig))
        seq, p = [], pos
        for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cig):
            n = int(n)
            if op in "M=X":
                seq.append(bs_read(ref, p, n, bool(flag & 0x40) != bool(flag & 0x10), rng)); p += n
            elif op in "IS":
                seq.append("".join(rng.choice("ACGT") for _ in range(n)))
            elif op in "DN":
                p += n
        s = "".join(seq)
        assert len(s) == ql
        R.append((pos, len(R), record(0, pos, flag, cig, s, [rng.choice([2, 12, 23, 37, 41]) for _ in range(ql)], qname=qname, mpos=mpos)))

    add(5, 99, "3I40M2D30M5S", "a", 30); add(30, 147, "4H6S20M3I20M1P10M", "a", 5)
    add(100, 83, "10=5X40M", "b", 120); add(120, 163, "30M10D30M", "b", 100)
    add(200, 99, "70000M", "long", 300); add(300, 147, "69000M", "long", 200)
    add(L - 60, 99, "60M", "end", L - 40); add(L - 40, 147, "40M", "end", L - 60)
    R.sort(key=lambda x: (x[0], x[1]))
    write_bam(tmp_path / "c.bam", [("c1", L)], [r for _, _, r in R])
    write_fasta(tmp_path / "c.fa", [("c1", ref)])
    for extra in (["--CHG", "--CHH"], ["--chunkSize", "1000", "--mergeContext", "--CHG"], ["--OT", "10,60000,10,0", "--nOB", "3,3,3,3", "--CHH", "--minOppositeDepth", "1"]):
        compare_cli(tmp_path, [str(tmp_path / "c.fa"), str(tmp_path / "c.bam")] + extra)


def test_tags_flags_and_tiny_contigs(tmp_path):
    """XG of integer type (must be ignored), XG:Z with other payloads, NH of every integer type, mates of opposite strand parity
    (no overlap resolution), contigs of 1..3 bases, a contig without reads, lower-case / N reference"""
    rng = random.Random(9)
    refs = [("one", "C"), ("two", "CG"), ("three", "cgN"), ("empty", "ACGTACGTCGCG"), ("main", ref_with_cpgs(600, 8))]
    ref = refs[4][1]
    R = []
    R.append(record(1, 0, 0, "2M", "CG", 40, qname="t2"))
    R.append(record(2, 0, 16, "3M", "CGA", 40, qname="t3"))
    k = 0
    for aux in (aux_i("XG", 67, "C"), aux_i("XG", 71, "C"), aux_i("XG", 2, "i"), aux_Z("XG", "CT"), aux_Z("XG", "GA"), aux_Z("XG", "AB"), aux_Z("XG", ""),
                aux_i("NH", 1, "c"), aux_i("NH", 2, "C"), aux_i("NH", -3, "s"), aux_i("NH", 70000, "I"), aux_i("NH", 4000000000, "I"), aux_Z("NH", "5"), b""):
        pos = 20 + 11 * k
        R.append(record(4, pos, 99, "80M", bs_read(ref, pos, 80, True, rng), 33, qname=f"q{k}", mpos=pos + 30, aux=aux))
        k += 1
    k = 0
    mates = []
    for aux in (aux_Z("XG", "CT"), aux_Z("XG", "GA"), b""):
        pos = 60 + 13 * k
        mates.append((pos, record(4, pos, 147 if k != 1 else 163, "80M", bs_read(ref, pos, 80, k != 1, rng), 29, qname=f"q{k + 3}", mpos=pos - 30, aux=aux)))
        k += 1
    main = sorted([(r[4 + 4:4 + 8], r) for r in R[2:]], key=lambda x: int.from_bytes(x[0], "little", signed=True))
    allrec = R[:2] + [r for _, r in sorted([(int.from_bytes(r[8:12], "little", signed=True), r) for r in R[2:]] + mates, key=lambda x: x[0])]
    write_bam(tmp_path / "d.bam", [(n, len(s)) for n, s in refs], allrec)
    write_fasta(tmp_path / "d.fa", refs)
    for extra in (["--CHG", "--CHH", "-q", "0"], ["--ignoreNH", "--CHH"], ["--keepDiscordant", "--keepSingleton", "--CHG", "--mergeContext"], ["-r", "main:100-300", "--CHH"], ["-r", "three"]):
        compare_cli(tmp_path, [str(tmp_path / "d.fa"), str(tmp_path / "d.bam")] + extra)


def test_undeterminable_strand_aborts_like_the_reference(tmp_path):
    """a paired read with neither 0x40 nor 0x80 has strand 0; the reference aborts in updateMetrics (common.c:122-125)"""
    ref = "ACGTCGCGCGTTTTCGCGCGAAAACGCG" * 4
    write_fasta(tmp_path / "e.fa", [("c", ref)])
    write_bam(tmp_path / "e.bam", [("c", len(ref))], [record(0, 2, 1 | 2, "40M", ref[2:42].replace("c", "C"), 40, qname="s0", mpos=2)])
    from conftest import run_oracle
    import methyldackel_amd as mdk
    ro = run_oracle([str(tmp_path / "e.fa"), str(tmp_path / "e.bam"), "-o", str(tmp_path / "o")], cwd=tmp_path)
    rg = mdk.run_cli([str(tmp_path / "e.fa"), str(tmp_path / "e.bam"), "-o", str(tmp_path / "g")], cwd=tmp_path)
    assert ro.returncode == -6 and rg.returncode == -6, (ro.returncode, rg.returncode)
    assert "Can't determine the strand of a read!" in ro.stderr and "Can't determine the strand of a read!" in rg.stderr


def test_many_cigar_operations_far_skips_and_a_crowd_of_pairs(tmp_path):
    """what k_prep_segs keeps per workgroup of 256 records: more segments than its LDS stage holds (reads with ten gapless runs each, and so
    many of them that the segment array has to grow and the preparation is run again), reads whose N operations reach across dozens of
    tiles, pairs whose mates sit in different workgroups of the scan (a block of unrelated reads in between) next to pairs that sit side
    by side, and a name with a read in each of three workgroups"""
    rng = random.Random(11)
    L = 260000
    ref = ref_with_cpgs(L, 9)
    R = []

    def add(pos, flag, cig, qname, mpos=0, q=35):
        import re
        seq, p = [], pos
        for n, op in re.findall(r"(\d+)([MIDNS])", cig):
            n = int(n)
            if op == "M":
                seq.append(bs_read(ref, p, n, bool(flag & 0x40) != bool(flag & 0x10), rng)); p += n
            elif op in "IS":
                seq.append("".join(rng.choice("ACGT") for _ in range(n)))
            else:
                p += n
        s = "".join(seq)
        R.append((pos, len(R), record(0, pos, flag, cig, s, [rng.choice([12, 23, 37, 41]) for _ in range(len(s))], qname=qname, mpos=mpos)))

    many = "8M1D" * 9 + "8M"                                  # ten runs of 8, 89 reference bases
    for k in range(1200):                                       # 600 pairs of ten-run reads, mates 40 apart: ~16 segments and more per pair
        pos = 1000 + 15 * k
        add(pos, 99, many, f"m{k}", pos + 40); add(pos + 40, 147, many, f"m{k}", pos)
    for k in range(40):                                         # skips of 1 kb .. 120 kb: up to ~58 tiles of 2048
        pos = 30000 + 50 * k
        add(pos, 99, f"30M{1000 + 3000 * k}N30M", f"n{k}", pos + 10); add(pos + 10, 147, "50M", f"n{k}", pos)
    for k in range(300):                                        # mates 700 records apart (other workgroups of the scan), names shared by three reads for every tenth
        pos = 150000 + 20 * k
        add(pos, 99, "60M", f"f{k}", pos + 14000)
        add(pos + 14000, 147, "60M", f"f{k}", pos)
        if k % 10 == 0:
            add(pos + 7000, 99 | 0x800, "40M", f"f{k}", pos)
    for k in range(700):                                        # the crowd in between, singles
        add(150010 + 20 * k, 0, "70M", f"s{k}")
    R.sort(key=lambda x: (x[0], x[1]))
    write_bam(tmp_path / "m.bam", [("c1", L)], [r for _, _, r in R])
    write_fasta(tmp_path / "m.fa", [("c1", ref)])
    for extra in (["--CHG", "--CHH", "-q", "0"], ["-F", "0", "--keepSingleton", "--keepDiscordant", "--chunkSize", "100000", "-q", "0"], ["--chunkSize", "20000", "--mergeContext", "-q", "0"]):
        compare_cli(tmp_path, [str(tmp_path / "m.fa"), str(tmp_path / "m.bam")] + extra)

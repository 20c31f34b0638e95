"""TEST INFRASTRUCTURE: a hand-made BAM for the corners of `mbias` and `perRead` that the synthetic generator never
produces: bases below -p at every kind of CIGAR boundary and at the very end of reads of even and odd length, a read
without sequence, a mapped read without CIGAR, an unmapped read placed at its mate, absent qualities, hard clips, pads,
reference skips, a 600-base read, reads hanging over the contig end."""
import random
import re

from bamwriter import record, write_bam, write_fasta

L = 3000


def make_ref(seed=11):
    r = random.Random(seed)
    s = []
    while len(s) < L:
        s += r.choice(["CG", "CG", "CG", "A", "T", "C", "G", "CAG", "CTG", "ACGT", "N", "c", "g", "cg"])
    return "".join(s)[:L]


def build(tmp_path, name="zoo"):
    rng = random.Random(12)
    ref = make_ref()
    R = []

    def bs(pos, n, odd):
        out = []
        for i in range(n):
            b = ref[pos + i].upper() if pos + i < L else "A"
            if odd and b == "C" and rng.random() < 0.4:
                b = "T"
            if not odd and b == "G" and rng.random() < 0.4:
                b = "A"
            out.append(b if b in "ACGT" else "N")
        return "".join(out)

    def add(pos, flag, cig, qname, quals=None, mpos=0, seq=None):
        ql = sum(int(n) for n, op in re.findall(r"(\d+)([MIS=X])", cig))
        odd = (not (flag & 1) and not (flag & 0x10)) or bool(flag & 0x40) != bool(flag & 0x10) if flag & 1 else not (flag & 0x10)
        if seq is None:
            parts, p = [], pos
            for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cig):
                n = int(n)
                if op in "M=X":
                    parts.append(bs(p, n, odd)); p += n
                elif op in "IS":
                    parts.append("".join(rng.choice("ACGT") for _ in range(n)))
                elif op in "DN":
                    p += n
            seq = "".join(parts)
        if quals is None:
            quals = [rng.choice([2, 12, 23, 37, 41]) for _ in range(len(seq))]
        elif callable(quals):
            quals = [quals(i, len(seq)) for i in range(len(seq))]
        assert len(seq) == ql or cig == "" or ql == 0 or len(seq) == 0
        R.append((pos, len(R), record(0, pos, flag, cig, seq, quals, qname=qname, mpos=mpos)))

    low_last = lambda i, n: 2 if i == n - 1 else 37           # last base below -p
    q37_first = lambda i, n: 37 if i == 0 else (2 if i == n - 1 else 30)   # and a first quality whose high nibble reads as 'C'
    q69_first = lambda i, n: 69 if i == 0 else (2 if i == n - 1 else 30)   # ... as 'G'
    for k in range(40):                                        # even and odd lengths, both strands, all over the contig
        pos = 20 + 61 * k
        n = 50 + (k % 7)
        add(pos, 0 if k % 2 else 16, f"{n}M", f"end{k}", [low_last, q37_first, q69_first][k % 3])
    lowrun = lambda i, n: 2 if i % 10 in (3, 4, 5) else 35     # runs of low bases: only every other one is skipped
    add(100, 99, "30M2I30M", "ins", lambda i, n: 2 if i in (29, 30, 61) else 35, mpos=140)
    add(140, 147, "30M3D30M", "del", lambda i, n: 2 if i in (29, 59) else 35, mpos=100)
    add(300, 0, "5S40M6S", "clip", lambda i, n: 2 if i in (4, 5, 44, 45, 50) else 35)
    add(400, 16, "4H20M1P20M10N20M3H", "zoo", lowrun)
    add(500, 0, "60M", "lowrun", lowrun)
    add(600, 0, "60M", "noqual", 255)
    add(700, 0, "10M", "noseq", seq="", quals=[])
    add(800, 0, "", "nocigar", seq="ACGTACGTAC", quals=30)
    add(900, 73, "50M", "mate_of_unmapped", mpos=900)
    add(900, 133, "", "mate_of_unmapped", seq="ACGTACGTACGT", quals=30, mpos=900)
    add(1000, 99, "600M", "long", mpos=1300)
    add(1300, 147, "580M", "long", mpos=1000)
    add(L - 30, 0, "30M", "tail")
    add(L - 20, 16, "40M", "overhang")
    for k in range(120):                                       # background coverage so that mbias has a profile
        pos = rng.randrange(0, L - 120)
        fl = rng.choice([0, 16])
        add(pos, fl, "100M", f"bg{k}")
    R.sort(key=lambda x: (x[0], x[1]))
    write_bam(tmp_path / f"{name}.bam", [("z1", L)], [r for _, _, r in R])
    write_fasta(tmp_path / f"{name}.fa", [("z1", ref)])
    return str(tmp_path / f"{name}.fa"), str(tmp_path / f"{name}.bam")

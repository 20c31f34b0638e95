"""CPU: csrc/host/mdk_fasta.c -- the several-thread loader of the reference genome gives what the one-thread loader gives (names cut at the first
blank, per contig the printable characters of its lines), on files with everything a FASTA may hold: text in front of the first header, empty
names and contigs, blank lines, CR LF, blanks and control bytes inside lines, '>' inside a line, no newline at the end."""
import os, random, subprocess
from pathlib import Path
import pytest
REPO = Path(__file__).resolve().parent.parent
PROBE = REPO / "tools" / "_build" / "fasta_probe"


@pytest.fixture(scope="module")
def probe():
    if not PROBE.exists():
        subprocess.run(["make", "-C", str(REPO), "tools/_build/fasta_probe"], check=True, capture_output=True)
    return PROBE


def load(probe, fn, nth):
    e = dict(os.environ); e["MDK_FASTA_THREADS"] = str(nth)
    r = subprocess.run([str(probe), str(fn)], env=e, capture_output=True, text=True, errors="replace")
    assert r.returncode == 0, r.stderr
    return r.stdout


def make(rnd):
    parts = []
    if rnd.random() < 0.5: parts.append("junk before\nACGT\n")
    for _ in range(rnd.randint(0, 6)):
        parts.append(">" + "".join(rnd.choice("abcXYZ_1.") for _ in range(rnd.randint(0, 8))) + rnd.choice(["", " desc here", "\tx", "\r"]) + "\n")
        for _ in range(rnd.randint(0, 30)):
            ln = "".join(rnd.choice("ACGTNacgtn") for _ in range(rnd.randint(0, 70)))
            if rnd.random() < 0.1: ln = ln[:len(ln) // 2] + rnd.choice([" ", "\t", "\x00", "\x7f", "\x80", ">"]) + ln[len(ln) // 2:]
            parts.append(ln + rnd.choice(["\n", "\r\n", "\n", "\n\n"]))
    s = "".join(parts)
    if rnd.random() < 0.3: s = s.rstrip("\n")
    return s.encode("latin1")


def test_threads_equal_one_thread(probe, tmp_path):
    rnd = random.Random(5); fn = tmp_path / "t.fa"
    for k in range(120):
        fn.write_bytes(make(rnd))
        ref = load(probe, fn, 1)
        for nth in (2, 3, 7, 64):
            assert load(probe, fn, nth) == ref, (k, nth)


def test_a_known_file(probe, tmp_path):
    fn = tmp_path / "k.fa"
    fn.write_bytes(b"ignored\n>chr1 first contig\nACGT\nac gt\r\n\n>chr2\n>chr3\tx\nNNNN>N\nA")
    want = None
    for nth in (1, 2, 5):
        got = [l.split("\t")[:2] for l in load(probe, fn, nth).splitlines()]
        assert got == [["chr1", "8"], ["chr2", "0"], ["chr3", "7"]]
        h = load(probe, fn, nth); want = want or h; assert h == want

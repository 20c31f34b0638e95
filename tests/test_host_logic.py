"""CPU-only checks of the product's HOST side (csrc/host): chunk schedule, admission, strand, qname pairing and
SoA packing.  The packed batches are evaluated by the slow test-only evaluator (tests/batch_eval.py) and must equal
the oracle's per-column counters exactly.  No device work happens here."""
import ctypes as C

import pytest

import methyldackel_amd as mdk
from batch_eval import eval_batch
from conftest import GOLDEN, read_dump, run_oracle, synth


def host_counts(args):
    plan = mdk.Plan(args)
    cfg = plan.dev_cfg()
    out = {}
    chunks = []
    refs = {}
    while True:
        c = plan.next_chunk()
        if c is None:
            break
        chunks.append((c.index, c.tid, c.beg, c.end, c.batch.n_reads))
        if c.skipped:
            continue
        name = plan.target_name(c.tid)
        if name not in refs:
            refs[name] = read_fasta(args)[name]
        for p, v in eval_batch(c.batch, refs[name], cfg, plan.regions(c.tid)).items():
            assert (c.tid, p) not in out
            out[(c.tid, p)] = v
    plan.close()
    return out, chunks


_fa_cache = {}


def read_fasta(args):
    fa = [a for a in map(str, args) if a.endswith(".fa")][0]
    if fa not in _fa_cache:
        d, name = {}, None
        for line in open(fa, "rb"):
            if line.startswith(b">"):
                name = line[1:].split()[0].decode()
                d[name] = bytearray()
            else:
                d[name] += line.strip()
        _fa_cache[fa] = {k: bytes(v) for k, v in d.items()}
    return _fa_cache[fa]


def check(tmp_path, args, variant=False):
    dump = tmp_path / "dump.tsv"
    r = run_oracle(list(args) + ["-o", tmp_path / "o"], cwd=tmp_path, dump=dump)
    assert r.returncode == 0, r.stderr
    want = read_dump(dump)
    if not variant:
        want = {k: v[:4] + (0, 0) for k, v in want.items() if v[2] + v[3] > 0}
    got, chunks = host_counts(list(args) + ["-o", tmp_path / "g"])
    assert got == want
    return chunks


FIX = [
    ["cg100.fa", "cg_aln.bam", "-q", "2"],
    ["cg100.fa", "cg_aln.bam", "-q", "2", "--CHG", "--CHH"],
    ["cg100.fa", "cg_aln.bam", "-q", "2", "--ignoreFlags", "0xD00"],
    ["cg100.fa", "cg_aln.bam", "-q", "2", "--nOT", "50,50,40,40"],
    ["cg100.fa", "cg_aln.bam", "-q", "2", "--OT", "10,90,20,80"],
    ["cg100.fa", "NH.bam", "-q", "1"],
    ["cg100.fa", "NH.bam", "-q", "1", "--ignoreNH"],
    ["chgchh.fa", "chgchh_aln.bam", "-q", "5", "--CHG", "--CHH"],
    ["chgchh.fa", "chgchh_aln.bam", "-q", "5", "--minConversionEfficiency", "0.9"],
    ["ct100.fa", "ct_aln.bam", "-q", "2", "--CHH"],
]


@pytest.mark.parametrize("args", FIX, ids=[" ".join(a[1:]) for a in FIX])
def test_fixture_batches(tmp_path, args):
    args = [str(GOLDEN / a) if (a.endswith(".fa") or a.endswith(".bam")) else a for a in args]
    check(tmp_path, args)


def test_fixture_variant_counters(tmp_path):
    args = [str(GOLDEN / "cg100.fa"), str(GOLDEN / "cg_with_variants.bam"), "-p", "1", "-q", "0", "--minOppositeDepth", "3", "--maxVariantFrac", "0.25"]
    check(tmp_path, args, variant=True)


SYN = [
    ("pe", []),
    ("pe", ["--CHG", "--CHH", "--chunkSize", "7000"]),
    ("pe", ["--chunkSize", "100", "-r", "chrS1:5000-9000"]),
    ("pe", ["-F", "0", "--keepDupes", "--keepSingleton", "--keepDiscordant", "--ignoreNH", "-q", "0", "--chunkSize", "3001"]),
    ("pe", ["--OT", "6,146,6,146", "--OB", "6,146,6,146", "--nOT", "2,3,4,5", "--CHH"]),
    ("pe", ["-B", "BBM", "--chunkSize", "9999"]),
    ("pe", ["-B", "BBM", "-b", "140", "-t", "0.6"]),
    ("bis", ["--CHG"]),
    ("bis", ["--CTOT", "5,100,5,100", "--nCTOB", "3,3,3,3", "--CHH", "--chunkSize", "5000"]),
    ("se", ["--CHG", "--CHH"]),
]


@pytest.mark.parametrize("which,extra", SYN, ids=[f"{w}:{' '.join(e)}" for w, e in SYN])
def test_synthetic_batches(tmp_path, small_synth, which, extra):
    extra = [str(small_synth / "pe.bbm") if e == "BBM" else e for e in extra]
    args = [str(small_synth / f"{which}.fa"), str(small_synth / f"{which}.bam")] + extra
    variant = False
    check(tmp_path, args, variant)


def test_synthetic_variant_mode(tmp_path, small_synth):
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--minOppositeDepth", "2", "--maxVariantFrac", "0.5", "--CHG", "--chunkSize", "8000"]
    check(tmp_path, args, variant=True)


def test_chunk_schedule_matches_reference_rules(tmp_path, small_synth):
    """chunk ends never split a CpG/CHG (adjustBounds) and chunks tile each contig exactly"""
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--chunkSize", "1000", "-o", str(tmp_path / "x")]
    _, chunks = host_counts(args)
    ref = read_fasta(args)
    by_tid = {}
    for idx, tid, beg, end, n in chunks:
        by_tid.setdefault(tid, []).append((beg, end))
    assert [c[0] for c in chunks] == list(range(len(chunks)))
    names = ["chrS1", "chrS2"]
    for tid, iv in by_tid.items():
        seq = ref[names[tid]].upper()
        assert iv[0][0] == 0 and iv[-1][1] == len(seq)
        for (b0, e0), (b1, e1) in zip(iv, iv[1:]):
            assert e0 == b1
            assert not (seq[e0 - 1:e0] == b"C" and seq[e0:e0 + 1] == b"G"), "CpG split across chunks"
            assert not (seq[e0 - 2:e0 - 1] == b"C" and seq[e0:e0 + 1] == b"G"), "CHG split across chunks"


import test_gpu_edge_cases as edge


@pytest.mark.parametrize("fn", [edge.test_quality_boost_wraps_above_213, edge.test_three_records_per_qname_across_chunks_and_zero_length_alignments,
                                edge.test_cigar_zoo_and_long_runs, edge.test_tags_flags_and_tiny_contigs], ids=lambda f: f.__name__[5:])
def test_edge_cases_host_side(tmp_path, monkeypatch, fn):
    """the adversarial BAMs of tests/test_gpu_edge_cases.py through the host side + slow evaluator, on CPU"""
    n = [0]

    def run(tmp, args, env=None):
        d = tmp / f"case{n[0]}"
        d.mkdir()
        n[0] += 1
        check(d, list(args), variant="--minOppositeDepth" in args)
        return d, d

    monkeypatch.setattr(edge, "compare_cli", run)
    fn(tmp_path)
    assert n[0] >= 3


def test_bigwig_mappability_equals_bbm(tmp_path, small_synth):
    """-M <bigWig> (own reader: B+ chromosome tree, R-tree, zlib blocks, NaN where uncovered) admits exactly the reads that
    -B <BBM of the same values> admits, which the oracle also agrees with"""
    fa, bam = str(small_synth / "pe.fa"), str(small_synth / "pe.bam")
    for extra in ([], ["-t", "0.6", "-b", "120"], ["-b", "0"], ["-b", "200"]):
        a, ca = host_counts([fa, bam, "-M", str(small_synth / "pe.bw"), "--chunkSize", "9000", "-o", str(tmp_path / "a")] + extra)
        b, cb = host_counts([fa, bam, "-B", str(small_synth / "pe.bbm"), "--chunkSize", "9000", "-o", str(tmp_path / "b")] + extra)
        assert a == b and ca == cb
    plain, _ = host_counts([fa, bam, "-o", str(tmp_path / "c")])
    assert plain != a


def test_bigwig_to_bbm_conversion_without_bam(tmp_path, small_synth):
    """`extract -M x.bw -N out` / `-O` only convert (extract.c:983-994): the BBM written is byte-identical to the generator's
    encoding of the same track, and no device is needed for it"""
    import shutil
    shutil.copy(small_synth / "pe.bw", tmp_path / "m.bw")
    r = mdk.run_cli(["-M", str(tmp_path / "m.bw"), "-N", str(tmp_path / "named")], cwd=tmp_path)
    assert r.returncode == 0 and "writing .bbm file to" in r.stderr
    assert (tmp_path / "named.bbm").read_bytes() == (small_synth / "pe.bbm").read_bytes()
    r = mdk.run_cli(["-M", str(tmp_path / "m.bw"), "-O"], cwd=tmp_path)
    assert r.returncode == 0 and (tmp_path / "m.bbm").read_bytes() == (small_synth / "pe.bbm").read_bytes()
    r = mdk.run_cli(["-O"], cwd=tmp_path)
    assert r.returncode == 255 and "You must specify a bigWig file" in r.stderr


def test_index_seek_equals_streaming(tmp_path, small_synth, monkeypatch):
    """with a .bai the reader seeks (once for -r, per own chunk when sharded); without it it reads the file through.
    Both must pack exactly the same batches."""
    fa, bam = str(small_synth / "pe.fa"), str(small_synth / "pe.bam")
    assert (small_synth / "pe.bam.bai").exists()

    def batches(extra, shard=None):
        plan = mdk.Plan([fa, bam, "-o", str(tmp_path / "x")] + extra)
        if shard:
            plan.set_shard(*shard)
        out = []
        while (c := plan.next_chunk()) is not None:
            segs = bytes(C.string_at(c.batch.seg, c.batch.n_segs * 32)) if c.batch.n_segs else b""
            out.append((c.index, c.tid, c.beg, c.end, c.skipped, c.batch.n_reads, segs, bytes(C.string_at(c.batch.blob, c.batch.blob_bytes))))
        plan.close()
        return out

    cases = [(["-r", "chrS1:12000-31000", "--chunkSize", "3000"], None), (["-r", "chrS2:19990", "--chunkSize", "700"], None),
             (["-r", "chrS1:39990-40000"], None), (["--chunkSize", "5000"], (1, 3)), (["--chunkSize", "5000", "-r", "chrS2:500-15000"], (0, 2))]
    total = 0
    for extra, shard in cases:
        with_index = batches(extra, shard)
        monkeypatch.setenv("MDK_NO_INDEX", "1")
        without = batches(extra, shard)
        monkeypatch.delenv("MDK_NO_INDEX")
        assert with_index == without, (extra, shard)
        total += sum(b[5] for b in with_index)
    assert total > 1000


@pytest.mark.timeout(180)
def test_chunk_larger_than_the_slab_pool(tmp_path, monkeypatch):
    """one chunk spanning more inflate slabs than the reader's look-ahead cap (huge --chunkSize, here a cap of 2 slabs):
    the slabs a chunk holds stay pinned until it is complete, so the inflater must be allowed past the cap"""
    monkeypatch.setenv("MDK_SLAB_CAP", "2")
    synth(tmp_path / "wide", "-L", "1500000", "-c", "40", "-s", "5")
    plan = mdk.Plan([str(tmp_path / "wide.fa"), str(tmp_path / "wide.bam"), "--chunkSize", "1500000", "-o", str(tmp_path / "x")])
    c = plan.next_chunk()
    assert c is not None and c.batch.n_reads > 300000
    assert plan.next_chunk() is None
    plan.close()


def test_records_split_across_bgzf_members_take_the_walking_path(tmp_path):
    """htslib never splits a record across BGZF members and the scanner then reads the inflate threads' record tables;
    a file that does split them (--split-records) must be scanned by walking and give the same batches"""
    import subprocess, sys
    synth(tmp_path / "a", "-L", "60000", "-c", "25", "-s", "21")
    synth(tmp_path / "b", "-L", "60000", "-c", "25", "-s", "21", "--split-records")
    assert (tmp_path / "a.bam").read_bytes() != (tmp_path / "b.bam").read_bytes()
    ca = check(tmp_path, [tmp_path / "a.fa", tmp_path / "a.bam", "--CHG", "--chunkSize", "7000"])
    cb = check(tmp_path, [tmp_path / "b.fa", tmp_path / "b.bam", "--CHG", "--chunkSize", "7000"])
    assert ca == cb
    code = ("import sys; sys.path.insert(0, %r); import methyldackel_amd as mdk\n"
            "p = mdk.Plan(sys.argv[1:])\nwhile p.next_chunk() is not None: pass\np.finish(); p.close()\n" % str(mdk.REPO))
    def walked(prefix):
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / f"{prefix}.fa"), str(tmp_path / f"{prefix}.bam"), "-o", str(tmp_path / "x")],
                           capture_output=True, text=True, env=dict(__import__("os").environ, MDK_HOST_PROFILE="1"))
        line = [l for l in r.stderr.splitlines() if "by walking" in l][0]
        return int(line.split("tables ")[1].split(",")[0]), int(line.split("by walking ")[1].split(";")[0])
    fast_a, slow_a = walked("a")
    fast_b, slow_b = walked("b")
    assert slow_a == 0 and fast_a > 5000
    assert slow_b > 0.9 * (fast_b + slow_b)

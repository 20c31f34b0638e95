"""GPU: chunk preparation on the device (csrc/mdk_prep.hip) against the host's (csrc/host/mdk_pipeline.c), chunk by chunk.

Both turn the same candidate records into md_seg arrays; the device emits them in read order, the host in order of
reference start, and payload offsets differ by construction (bytes into the uploaded records vs 4-byte units into a packed
blob), so the comparison is on everything else as a multiset.  Per-position results from the device-prepared slot must equal
the host-prepared one exactly.  The command line tests of the other files run the device preparation too (it is the
default of `extract`); here MDK_HOST_PREP=1 is exercised as well."""
import ctypes as C
import os
from collections import Counter

import pytest

import methyldackel_amd as mdk
from bamwriter import record, write_bam, write_fasta
from conftest import GOLDEN, synth
from test_gpu_parity import compare_cli

pytestmark = pytest.mark.gpu


def seg_key(g):
    k = (g.rpos, g.len, g.q0, g.l_qseq, g.sf)
    return k + ((g.msf, g.m_q0, g.m_l_qseq) if g.sf & 32 else ())


def sites_list(s):
    out = []
    for i in range(s.n_sites):
        r = s.site[i]
        out.append((r.pos, r.nmeth, r.nunmeth, r.meta) + ((s.var[i].noff, s.var[i].nvar) if s.var else ()))
    return out


def both_ways(args):
    """every chunk through host preparation and through device preparation on the same device handle"""
    ph, pd = mdk.Plan(args), mdk.Plan(args)
    pd.set_prep(1)
    cfg = ph.dev_cfg()
    dev = mdk.Device(cfg)
    dev.set_prep(pd.prep_cfg())
    n_chunks = n_reads = n_segs = 0
    while True:
        ch, cd = ph.next_chunk(), pd.next_chunk()
        assert (ch is None) == (cd is None)
        if ch is None:
            break
        assert (ch.index, ch.tid, ch.beg, ch.end, ch.skipped) == (cd.index, cd.tid, cd.beg, cd.end, cd.skipped)
        if ch.skipped:
            continue
        ph.ensure_reference(dev, ch.tid); pd.ensure_reference(dev, cd.tid)
        dev.submit(0, ch.batch)
        want = sites_list(dev.download(0))
        dev.submit_raw(1, cd.raw)
        got = sites_list(dev.download(1))
        segs, n, nr = dev.debug_segments(1)
        assert nr == ch.batch.n_reads and n == ch.batch.n_segs, (ch.index, nr, ch.batch.n_reads, n, ch.batch.n_segs)
        assert Counter(seg_key(segs[i]) for i in range(n)) == Counter(seg_key(ch.batch.seg[i]) for i in range(ch.batch.n_segs)), ch.index
        assert got == want, ch.index
        n_chunks += 1; n_reads += nr; n_segs += n
    dev.close(); ph.close(); pd.close()
    return n_chunks, n_reads, n_segs


CMDS = [
    ("pe", []),
    ("pe", ["--CHG", "--CHH", "--chunkSize", "2500"]),
    ("pe", ["-q", "0", "-F", "0", "--keepDupes", "--keepSingleton", "--keepDiscordant", "--ignoreNH", "--chunkSize", "9000"]),
    ("pe", ["-R", "2", "--minOppositeDepth", "2", "--maxVariantFrac", "0.3", "--chunkSize", "5000"]),
    ("pe", ["--minConversionEfficiency", "0.7", "--CHH", "--chunkSize", "6000"]),
    ("pe", ["-B", "BBM", "--minMappableBases", "30", "--chunkSize", "7000"]),
    ("pe", ["-M", "BW", "--mappabilityThreshold", "0.5", "--chunkSize", "8000"]),
    ("pe", ["--OT", "3,140,5,130", "--nOB", "2,3,4,5", "--chunkSize", "10000"]),
    ("bis", ["--CHG", "--chunkSize", "4000"]),
    ("se", ["--CHH", "--chunkSize", "3000"]),
]


@pytest.mark.parametrize("which,extra", CMDS, ids=[f"{w}:{' '.join(e)}" for w, e in CMDS])
def test_device_prep_equals_host_prep(tmp_path, small_synth, which, extra):
    extra = [str(small_synth / "pe.bbm") if e == "BBM" else str(small_synth / "pe.bw") if e == "BW" else e for e in extra]
    n_chunks, n_reads, n_segs = both_ways([str(small_synth / f"{which}.fa"), str(small_synth / f"{which}.bam")] + extra + ["-o", str(tmp_path / "x")])
    assert n_chunks >= 1 and n_reads > 500 and n_segs >= n_reads


@pytest.mark.parametrize("wide", [None, "0", "1"], ids=["adaptive", "narrow", "wide"])
def test_long_read_names_and_long_aux_fields(tmp_path, wide):
    """k_prep_scan gathers a window of each record into LDS (GatherView): a sequencer's read names (38 letters) do not fit the default window --
    the lanes read such records from HBM, and after the first chunk's counters the wide-window kernel takes over -- and a bisulfite aligner's aux
    fields (XM:Z, 150 letters) are walked in HBM.  The three arrangements (chosen from the counters; MDK_SCAN_WIDE=0; MDK_SCAN_WIDE=1), chunk by
    chunk against the host preparation, then the command against the oracle."""
    synth(tmp_path / "il", "-L", "60000", "-c", "25", "-s", "21", "--illumina", "--extras")
    old = os.environ.get("MDK_SCAN_WIDE")
    try:
        if wide is None: os.environ.pop("MDK_SCAN_WIDE", None)
        else: os.environ["MDK_SCAN_WIDE"] = wide
        n_chunks, n_reads, n_segs = both_ways([str(tmp_path / "il.fa"), str(tmp_path / "il.bam"), "--chunkSize", "7000", "--CHG", "-o", str(tmp_path / "x")])
        assert n_chunks >= 8 and n_reads > 5000
    finally:
        if old is None: os.environ.pop("MDK_SCAN_WIDE", None)
        else: os.environ["MDK_SCAN_WIDE"] = old
    env = {"MDK_HOST_PROFILE": "1"}
    if wide is not None: env["MDK_SCAN_WIDE"] = wide
    compare_cli(tmp_path, [str(tmp_path / "il.fa"), str(tmp_path / "il.bam"), "--chunkSize", "9000", "--CHH"], env=env)
    assert ("wide windows from here on" in (tmp_path / "gpu_stderr.txt").read_text()) == (wide is None)


def test_device_prep_on_reference_fixtures(tmp_path):
    for fa, bam, extra in (("cg100.fa", "cg_aln.bam", ["-q", "2"]), ("cg100.fa", "cg_aln.bam", ["--ignoreFlags", "0xD00", "-q", "2"]), ("cg100.fa", "cg_with_variants.bam", ["-p", "1", "-q", "0", "--minOppositeDepth", "3"]),
                           ("chgchh.fa", "chgchh_aln.bam", ["-q", "5", "--minConversionEfficiency", "0.9"]), ("cg100.fa", "NH.bam", ["-q", "1"]), ("cg100.fa", "cg_aln.bam", ["-q", "2", "--chunkSize", "7"])):
        both_ways([str(GOLDEN / fa), str(GOLDEN / bam)] + extra + ["-o", str(tmp_path / "x")])


def test_device_prep_bed(tmp_path, small_synth):
    from bedgen import random_bed
    bed = tmp_path / "r.bed"
    names = [l[1:].split()[0] for l in open(small_synth / "pe.fa") if l.startswith(">")]      # contig names come from the generator's FASTA
    random_bed(bed, list(zip(names, [40000, 20000])), 60, seed=3)
    both_ways([str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed), "--keepStrand", "--chunkSize", "5000", "-o", str(tmp_path / "x")])


def many_records_one_name(tmp_path, n, step=10):
    ref = ("ACGTCGCGTTCGAACGCGTA" * 40)[:800]
    recs = []
    for k in range(n):
        pos = 10 + step * k
        fl = 99 if k % 2 == 0 else 147
        recs.append(record(0, pos, fl, "60M", ref[pos:pos + 60].replace("C", "T") if k % 3 == 0 else ref[pos:pos + 60], 35, qname="same", mpos=pos + 7))
    write_bam(tmp_path / "m.bam", [("c1", len(ref))], recs)
    write_fasta(tmp_path / "m.fa", [("c1", ref)])
    return str(tmp_path / "m.fa"), str(tmp_path / "m.bam")


@pytest.mark.parametrize("n,step,on_device", [(3, 10, True), (16, 10, True), (17, 10, False), (40, 10, False), (14, 5, False), (14, 9, True)])
def test_names_with_many_records(tmp_path, n, step, on_device):
    """up to 16 records of one name, at most 8 of them in the pileup buffer at once (60-base reads every `step` bases), stay on
    the device; beyond either limit it hands the chunk back (MDK_ERR_PREP_HOST) and the command prepares it on the host:
    byte-identical to the oracle either way"""
    fa, bam = many_records_one_name(tmp_path, n, step)
    compare_cli(tmp_path, [fa, bam, "-F", "0", "-q", "0", "--keepDupes"])
    # and through the API: the device's answer for the big groups is the documented error code
    plan = mdk.Plan([fa, bam, "-F", "0", "-q", "0", "--keepDupes", "-o", str(tmp_path / "x")]); plan.set_prep(1)
    dev = mdk.Device(plan.dev_cfg()); dev.set_prep(plan.prep_cfg())
    c = plan.next_chunk(); plan.ensure_reference(dev, c.tid)
    dev.submit_raw(0, c.raw)
    s = mdk.md_sites()
    rc = dev.L.md_dev_download(dev.h, 0, C.byref(s))
    assert rc == (0 if on_device else -7)
    if rc:
        plan.host_prepare(c)
        dev.submit(0, c.batch)
        assert dev.download(0).n_sites > 0
    dev.close(); plan.close()


@pytest.mark.parametrize("which,extra", [("pe", ["--CHG", "--chunkSize", "2500", "--mergeContext"]), ("bis", ["--CHH"]), ("pe", ["--minConversionEfficiency", "0.7"])])
def test_host_prep_mode_still_byte_exact(tmp_path, small_synth, which, extra):
    compare_cli(tmp_path, [str(small_synth / f"{which}.fa"), str(small_synth / f"{which}.bam")] + extra, env={"MDK_HOST_PREP": "1"})


def test_segment_array_growth(tmp_path):
    """reads full of deletions: far more segments than the 2-per-record the first launch reserves, so the preparation runs
    twice (MDK_ERR_PREP_REDO inside the library)"""
    import random
    rng = random.Random(5)
    ref = "".join(rng.choice("ACGT") for _ in range(3000))
    recs = []
    for k in range(300):
        pos = rng.randrange(0, 2000)
        cig = "".join(f"8M1D" for _ in range(9)) + "8M"
        seq = "".join(ref[pos + 9 * j:pos + 9 * j + 8] for j in range(10))
        recs.append((pos, record(0, pos, 0 if k % 2 else 16, cig, seq, 30, qname=f"d{k}")))
    recs.sort(key=lambda x: x[0])
    write_bam(tmp_path / "d.bam", [("c1", len(ref))], [r for _, r in recs])
    write_fasta(tmp_path / "d.fa", [("c1", ref)])
    compare_cli(tmp_path, [str(tmp_path / "d.fa"), str(tmp_path / "d.bam"), "--CHH", "--CHG"])
    n_chunks, n_reads, n_segs = both_ways([str(tmp_path / "d.fa"), str(tmp_path / "d.bam"), "--CHH", "-o", str(tmp_path / "x")])
    assert n_segs == 10 * n_reads


def random_bam(tmp_path, seed, n_reads=700, strand0=True):
    """records the synthetic generator never writes: every aux type (incl. B arrays, H strings, double, malformed tails), NH and
    XG tags of every integer/char type before and after other tags, names of 1..60 characters, names shared by 1-5 records,
    CIGARs with I/D/N/S/H/P/=/X and more query bases than the record stores, every flag combination, mates on other contigs"""
    import random
    import struct as st
    from bamwriter import aux_Z, aux_i
    rng = random.Random(seed)
    L = 6000
    ref = "".join(rng.choice("ACGTCGCGN" if i % 97 else "acgt") for i in range(L))
    names = [("n%d" % k) * rng.randint(1, 6) for k in range(n_reads // 2)] + ["x", "a-very-long-read-name/with:punctuation.and_more_characters_0123456789"]
    recs = []
    for _ in range(n_reads):
        pos = rng.randrange(0, L - 400)
        ops = []
        q = 0
        for _k in range(rng.randint(1, 6)):
            op = rng.choice("MMMMM=XIDNSHP")
            ln = rng.randint(1, 60)
            ops.append((ln, op))
            if op in "M=XIS":
                q += ln
        if not any(op in "M=X" for _, op in ops):
            ops.append((rng.randint(1, 80), "M")); q += ops[-1][0]
        cig = "".join(f"{ln}{op}" for ln, op in ops)
        lq = q if rng.random() < 0.9 else max(0, q - rng.randint(1, 10))           # sometimes fewer bases than the CIGAR consumes
        seq = "".join(rng.choice("ACGTN") for _ in range(lq))
        qual = [rng.choice([0, 3, 5, 20, 37, 41, 93, 214, 255]) for _ in range(lq)]
        flag = rng.choice([0, 16, 99, 147, 83, 163, 65, 129, 113, 177, 73, 89, 1024, 512, 256, 2048, 4, 69, 0x63 | 0x400] + ([1, 3] if strand0 else []))       # 1, 3: paired without read number = strand 0
        aux = b""
        tags = []
        if rng.random() < 0.5:
            t = rng.choice("cCsSiI")
            tags.append(aux_i("NH", rng.choice([0, 1, 2, 5, -1] if t in "csi" else [0, 1, 2, 5, 200]), t) if rng.random() < 0.9 else aux_Z("NH", "2"))
        if rng.random() < 0.5:
            tags.append(aux_Z("XG", rng.choice(["CT", "GA", "C", "G", "", "TT"])) if rng.random() < 0.85 else aux_i("XG", 7, "C"))
        for _k in range(rng.randint(0, 4)):
            t = rng.choice(["AS", "XM", "MD", "ZB", "ZH", "ZD", "ZA"])
            if t == "AS": tags.append(aux_i("AS", rng.randint(-100, 100), "i"))
            elif t == "XM": tags.append(aux_Z("XM", "".join(rng.choice("zZ.hHxX") for _ in range(rng.randint(0, 150)))))
            elif t == "MD": tags.append(aux_Z("MD", str(rng.randint(1, 150))))
            elif t == "ZB": sub = rng.choice("cCsSiIf"); n = rng.randint(0, 9); w = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]; tags.append(b"ZBB" + sub.encode() + st.pack("<i", n) + bytes(rng.randrange(256) for _ in range(n * w)))
            elif t == "ZH": tags.append(b"ZHH" + b"1AE301" + b"\0")
            elif t == "ZD": tags.append(b"ZDd" + st.pack("<d", rng.random()))
            else: tags.append(b"ZAA" + bytes([rng.randrange(33, 127)]))
        rng.shuffle(tags)
        aux = b"".join(tags)
        if rng.random() < 0.04:
            aux += rng.choice([b"XGZCT", b"NH", b"NHq\x01", b"XGB?\x01\0\0\0", b"ZZZunterminated"])        # malformed tail: the walk stops there
        recs.append((pos, len(recs), record(0, pos, flag, cig, seq, qual, qname=rng.choice(names), mapq=rng.choice([0, 3, 10, 40, 255]),
                                            mtid=rng.choice([0, 0, 0, 1, -1]), mpos=rng.randrange(0, L), aux=aux)))
    recs.sort(key=lambda x: (x[0], x[1]))
    tag = f"r{seed}{'s' if strand0 else 'n'}"
    write_bam(tmp_path / f"{tag}.bam", [("c1", L), ("c2", 100)], [r for _, _, r in recs])
    write_fasta(tmp_path / f"{tag}.fa", [("c1", ref), ("c2", "ACGT" * 25)])
    return str(tmp_path / f"{tag}.fa"), str(tmp_path / f"{tag}.bam")


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_device_prep_on_random_records(tmp_path, seed):
    """host preparation == device preparation (admission, strand, pairing, segments) chunk by chunk on adversarial records, under
    several option sets; chunks the device hands back (names with many records) are counted, not compared"""
    with_s0, without_s0 = random_bam(tmp_path, seed), random_bam(tmp_path, seed, strand0=False)
    for extra in (["-q", "0", "-F", "0", "--keepDupes", "--keepSingleton", "--keepDiscordant", "--CHH", "--chunkSize", "700"],
                  ["-q", "5", "--ignoreNH", "--chunkSize", "1500", "--CHG"], ["-R", "1", "-F", "1024", "--chunkSize", "6000"],
                  ["-q", "0", "--minConversionEfficiency", "0.3", "--CHH", "--keepDiscordant", "--keepSingleton", "--chunkSize", "900"]):
        # (the conversion-efficiency filter aborts the process on a strand-0 read, host and reference alike: those records stay out)
        fa, bam = without_s0 if "--minConversionEfficiency" in extra else with_s0
        args = [fa, bam] + extra + ["-o", str(tmp_path / "x")]
        ph, pd = mdk.Plan(args), mdk.Plan(args)
        pd.set_prep(1)
        dev = mdk.Device(ph.dev_cfg()); dev.set_prep(pd.prep_cfg())
        compared = handed_back = 0
        while True:
            ch, cd = ph.next_chunk(), pd.next_chunk()
            assert (ch is None) == (cd is None)
            if ch is None:
                break
            if ch.skipped:
                continue
            ph.ensure_reference(dev, ch.tid); pd.ensure_reference(dev, cd.tid)
            dev.submit_raw(1, cd.raw)
            s = mdk.md_sites()
            rc = dev.L.md_dev_download(dev.h, 1, C.byref(s))
            if rc == -7:
                handed_back += 1
                continue
            if rc == -5:                       # a strand-0 read reached a call: the host-prepared chunk must say the same
                dev.submit(0, ch.batch)
                assert dev.L.md_dev_download(dev.h, 0, C.byref(mdk.md_sites())) == -5
                continue
            assert rc == 0, dev.L.md_dev_last_error()
            got = sites_list(s)
            segs, n, nr = dev.debug_segments(1)
            dev.submit(0, ch.batch)
            assert nr == ch.batch.n_reads and n == ch.batch.n_segs, (seed, extra, ch.index)
            assert Counter(seg_key(segs[i]) for i in range(n)) == Counter(seg_key(ch.batch.seg[i]) for i in range(ch.batch.n_segs)), (seed, extra, ch.index)
            assert got == sites_list(dev.download(0)), (seed, extra, ch.index)
            compared += 1
        assert compared >= 1
        dev.close(); ph.close(); pd.close()

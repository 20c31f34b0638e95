"""CPU-only: what a plan in device-preparation mode hands out (include/mdk_extract.h mdk_plan_set_prep, md_raw_batch).
No per-record work happens on the host in that mode, so the description must be exactly the candidate set of the chunk's
region query (common.c:413 via extract.c:379: same contig, pos < end, bam_endpos > beg, file order), and every rec_off must
point at a record's block_size word in the concatenation of the ranges."""
import ctypes as C
import gzip
import struct

import pytest

import methyldackel_amd as mdk
from conftest import GOLDEN, synth


def all_records(bam):
    d = gzip.open(bam).read()
    p = 4; lt, = struct.unpack_from("<i", d, p); p += 4 + lt
    nr, = struct.unpack_from("<i", d, p); p += 4
    for _ in range(nr):
        ln, = struct.unpack_from("<i", d, p); p += 8 + ln
    out = []
    while p < len(d):
        bs, = struct.unpack_from("<i", d, p)
        r = d[p + 4:p + 4 + bs]
        tid, pos, lrn, mq, bn, nc, fl, ls = struct.unpack_from("<iiBBHHHi", r, 0)
        cig = struct.unpack_from("<%dI" % nc, r, 32 + lrn)
        rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
        out.append((tid, pos, pos + max(rlen, 1), bytes(d[p:p + 4 + bs])))
        p += 4 + bs
    return out


def raw_chunks(args):
    plan = mdk.Plan(args)
    plan.set_prep(1)
    out = []
    while (c := plan.next_chunk()) is not None:
        if c.skipped:
            out.append((c.index, c.tid, c.beg, c.end, None))
            continue
        assert c.prep == 1 and c.batch.n_segs == 0
        cat = b"".join(C.string_at(c.raw.range[i].ptr, c.raw.range[i].bytes) for i in range(c.raw.n_ranges))
        recs = []
        offs = mdk.raw_record_offsets(c.raw)
        assert len(offs) == c.raw.n_records
        for o in offs:
            bs, = struct.unpack_from("<I", cat, o)
            recs.append(cat[o:o + 4 + bs])
        assert sum(len(r) for r in recs) == len(cat), "the ranges hold exactly the listed records"
        assert (c.raw.tid, c.raw.beg, c.raw.end) == (c.tid, c.beg, c.end)
        out.append((c.index, c.tid, c.beg, c.end, recs))
    plan.close()
    return out


@pytest.mark.parametrize("chunk", [1000000, 7000, 333])
def test_raw_batch_is_the_region_query(tmp_path, chunk):
    synth(tmp_path / "s", "-L", "40000,9000", "-c", "12", "-s", "5", "--extras")
    recs = all_records(tmp_path / "s.bam")
    chunks = raw_chunks([str(tmp_path / "s.fa"), str(tmp_path / "s.bam"), "--chunkSize", str(chunk), "-o", str(tmp_path / "x")])
    assert len(chunks) >= 2
    for _, tid, beg, end, got in chunks:
        want = [raw for t, p, e, raw in recs if t == tid and p < end and e > beg]
        assert got == want


def test_raw_batch_split_records_and_fixture(tmp_path):
    """records that straddle BGZF members (the slow scanner path) and the reference's own fixture"""
    synth(tmp_path / "s", "-L", "30000", "-c", "10", "-s", "6", "--split-records")
    recs = all_records(tmp_path / "s.bam")
    for _, tid, beg, end, got in raw_chunks([str(tmp_path / "s.fa"), str(tmp_path / "s.bam"), "--chunkSize", "4000", "-o", str(tmp_path / "x")]):
        assert got == [raw for t, p, e, raw in recs if t == tid and p < end and e > beg]
    recs = all_records(GOLDEN / "cg_aln.bam")
    (c,) = raw_chunks([str(GOLDEN / "cg100.fa"), str(GOLDEN / "cg_aln.bam"), "-o", str(tmp_path / "y")])
    assert c[4] == [raw for _, _, _, raw in recs] and len(c[4]) == 4


def test_prep_cfg_mirrors_the_options(tmp_path):
    plan = mdk.Plan([str(GOLDEN / "cg100.fa"), str(GOLDEN / "cg_aln.bam"), "-q", "3", "-p", "7", "-F", "1024", "-R", "2", "--keepDupes", "--ignoreNH",
                     "--keepSingleton", "--minConversionEfficiency", "0.5", "-o", str(tmp_path / "z")])
    c = plan.prep_cfg()
    assert (c.min_mapq, c.min_phred, c.ignore_flags, c.require_flags, c.keep_dupes, c.ignore_nh, c.keep_singleton, c.keep_discordant) == (3, 7, 0, 2, 1, 1, 1, 0)
    assert abs(c.min_conv_eff - 0.5) < 1e-6 and c.map_on == 0 and c.no_pairing == 0
    plan.close()

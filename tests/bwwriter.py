"""TEST INFRASTRUCTURE: write a tiny bigWig (bedGraph sections, zlib-compressed, one-level R-tree; UCSC bbi layout as in SURVEY.md Appendix D)
from explicit runs, so that tests can place values at the rounding edges of the mappability rule (extract.c:1138-1144)."""
import struct
import zlib


def write_bigwig(path, contigs, runs):
    """contigs: [(name, length)]; runs: {contig index: [(beg, end, float value)]} sorted, non-overlapping; uncovered bases read as NaN"""
    nct = len(contigs)
    key = max(len(n) for n, _ in contigs)
    chrom_tree = 64 + 40
    data_off = chrom_tree + 32 + 4 + nct * (key + 8)
    body = bytearray(struct.pack("<Q", 0)); idx = bytearray(); nblocks = 0; maxraw = 0
    for t in range(nct):
        items = runs.get(t, [])
        for k0 in range(0, len(items), 512):
            part = items[k0:k0 + 512]
            raw = struct.pack("<IIIIIBBH", t, part[0][0], part[-1][1], 0, 0, 1, 0, len(part)) + b"".join(struct.pack("<IIf", b, e, v) for b, e, v in part)
            maxraw = max(maxraw, len(raw))
            comp = zlib.compress(raw, 6)
            idx += struct.pack("<IIIIQQ", t, part[0][0], t, part[-1][1], data_off + len(body), len(comp))
            body += comp; nblocks += 1
    body[0:8] = struct.pack("<Q", nblocks)
    index_off = data_off + len(body)
    hdr = struct.pack("<IHHQQQHHQQIQ", 0x888FFC26, 4, 0, chrom_tree, data_off, index_off, 0, 0, 0, 64, maxraw, 0) + bytes(40)
    hdr += struct.pack("<IIIIQQ", 0x78CA8C91, nct, key, 8, nct, 0) + struct.pack("<BBH", 1, 0, nct)
    for t, (n, l) in enumerate(contigs):
        hdr += n.encode().ljust(key, b"\0") + struct.pack("<II", t, l)
    assert len(hdr) == data_off
    rt = struct.pack("<IIQIIIIQII", 0x2468ACE0, 256, nblocks, 0, 0, nct - 1, contigs[-1][1], index_off, 512, 0) + struct.pack("<BBH", 1, 0, nblocks)
    open(path, "wb").write(hdr + bytes(body) + rt + bytes(idx))

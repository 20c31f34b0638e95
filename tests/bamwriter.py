"""TEST INFRASTRUCTURE: write small BAM + FASTA files record by record (BGZF, uncompressed-level deflate is fine)."""
import struct
import zlib

OPS = {c: i for i, c in enumerate("MIDNSHP=X")}
NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def cigar(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | OPS[ch])
            num = ""
    return out


def reg2bin(beg, end):
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return off + (beg >> sh)
    return 0


def record(tid, pos, flag, cig, seq, qual, qname="r", mapq=40, mtid=None, mpos=-1, tlen=0, aux=b""):
    """seq: string over ACGTN (query order as stored in BAM), qual: list of ints or one int"""
    c = cigar(cig) if isinstance(cig, str) else cig
    l = len(seq)
    if isinstance(qual, int):
        qual = [qual] * l
    nib = bytearray((l + 1) // 2)
    for i, ch in enumerate(seq):
        nib[i >> 1] |= NT16[ch] << (0 if i & 1 else 4)
    rlen = sum(x >> 4 for x in c if (x & 15) in (0, 2, 3, 7, 8))
    qn = qname.encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), mapq, reg2bin(pos, pos + max(rlen, 1)), len(c), flag, l,
                       tid if mtid is None else mtid, mpos, tlen)
    body += qn + struct.pack("<%dI" % len(c), *c) + bytes(nib) + bytes(qual) + aux
    return struct.pack("<i", len(body)) + body


def bgzf(data: bytes, level=1) -> bytes:
    out = bytearray()
    for o in range(0, len(data), 60000):
        blk = data[o:o + 60000]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(blk) + co.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25)
        out += comp + struct.pack("<II", zlib.crc32(blk), len(blk))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def write_bam(path, contigs, records):
    """contigs: [(name, length)], records: list of bytes from record(), already coordinate sorted"""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in contigs)
    h = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
    for n, l in contigs:
        h += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    open(path, "wb").write(bgzf(h + b"".join(records)))


def write_fasta(path, seqs, width=60):
    with open(path, "w") as f:
        for n, s in seqs:
            f.write(f">{n}\n")
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def aux_Z(tag, s):
    return tag.encode() + b"Z" + s.encode() + b"\0"


def aux_i(tag, v, t="i"):
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[t]
    return tag.encode() + t.encode() + struct.pack(fmt, v)

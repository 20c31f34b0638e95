"""Reference vector t8 (tests/test.py:82-88: `--nOT 50,50,40,40 -q 2` on cg_aln.bam asserts 12 lines; common.c:174-208 as written gives 11):
which line would be the twelfth, and which single base of which read would have to stay unmasked for it -- so that one run of a real
binary (tests/golden/with_reference.sh) settles it.  Decodes the fixture BAM itself (no oracle, no product).  Usage: python tests/t8_trace.py"""
import struct
import zlib
from pathlib import Path

G = Path(__file__).resolve().parent / "golden"


def records(path):
    raw = path.read_bytes(); data = b""; o = 0
    while o + 18 <= len(raw):
        xlen = struct.unpack_from("<H", raw, o + 10)[0]; bs = struct.unpack_from("<H", raw, o + 16)[0] + 1
        data += zlib.decompress(raw[o + 12 + xlen:o + bs - 8], wbits=-15); o += bs
    lt = struct.unpack_from("<i", data, 4)[0]; p = 8 + lt; nref = struct.unpack_from("<i", data, p)[0]; p += 4
    for _ in range(nref):
        ln = struct.unpack_from("<i", data, p)[0]; p += 8 + ln
    while p + 4 <= len(data):
        bs = struct.unpack_from("<i", data, p)[0]; r = data[p + 4:p + 4 + bs]; p += 4 + bs
        tid, pos, lqn, mapq, _bin, ncig, flag, lq = struct.unpack_from("<iiBBHHHi", r, 0)
        name = r[32:32 + lqn - 1].decode(); cig = struct.unpack_from("<%dI" % ncig, r, 32 + lqn)
        so = 32 + lqn + 4 * ncig; seq = r[so:so + (lq + 1) // 2]; qual = r[so + (lq + 1) // 2:so + (lq + 1) // 2 + lq]
        yield dict(name=name, pos=pos, flag=flag, mapq=mapq, cig=cig, lq=lq, seq=seq, qual=qual)


def main():
    ref = "".join(l.strip() for l in open(G / "cg100.fa") if not l.startswith(">"))
    lb1, rb1, lb2, rb2 = 50, 50, 40, 40                 # --nOT: read #1 left/right, read #2 left/right (common.c:174-208, index 4*(strand-1)+{0,1,2,3})
    print("reads of cg_aln.bam (MAPQ >= 2) and the query interval --nOT 50,50,40,40 leaves unmasked, by the code and one base more on the right:")
    cover = {}
    for r in records(G / "cg_aln.bam"):
        if r["mapq"] < 2 or r["flag"] & 0x4:
            continue
        paired = r["flag"] & 1
        strand = (2 if (r["flag"] & 0x50) == 0x50 else 1 if r["flag"] & 0x40 else 1 if (r["flag"] & 0x90) == 0x90 else 2 if r["flag"] & 0x80 else 0) if paired else (2 if r["flag"] & 0x10 else 1)
        read2 = bool(r["flag"] & 0x80)
        lb, rb = (lb2, rb2) if read2 else (lb1, rb1)
        lo, hi = min(lb, r["lq"]), r["lq"] - min(rb, r["lq"])          # kept query indices [lo, hi) by the code (OT only: strand 1)
        if strand != 1:
            lo, hi = 0, r["lq"]
        q, p = 0, r["pos"]
        for c in r["cig"]:
            op, ln = c & 15, c >> 4
            if op in (0, 7, 8):
                for k in range(ln):
                    cover.setdefault(p + k, []).append((r["name"], "read2" if read2 else "read1", strand, q + k, lo, hi, r["lq"]))
                q += ln; p += ln
            elif op in (1, 4):
                q += ln
            elif op in (2, 3):
                p += ln
        print(f"  {r['name']:>12} {'read2' if read2 else 'read1'} flag {r['flag']:4d} strand {strand} pos {r['pos']:3d} l_qseq {r['lq']:3d}: kept query [{lo},{hi}) by the code, [{lo},{hi + 1}) if the right trim masked rb-1 bases")
    print("\nCpG C positions (0-based) where a base is kept only under the 'rb-1' reading -- the candidates for the twelfth line:")
    for p in sorted(cover):
        if ref[p].upper() != "C" or p + 1 >= len(ref) or ref[p + 1].upper() != "G":
            continue
        by_code = [x for x in cover[p] if x[2] == 1 and x[4] <= x[3] < x[5]]
        extra = [x for x in cover[p] if x[2] == 1 and x[3] == x[5] and x[5] < x[6]]
        if extra and not by_code:
            for x in extra:
                print(f"  position {p}: only read {x[0]} ({x[1]}) covers it with a kept base, and only if query index {x[3]} (= l_qseq {x[6]} - rb) is NOT masked")
    print("\nBy common.c:198-204 (`for(i=0; i<rb; i++) qual[l_qseq-1-i] = 0`) that index IS masked: 11 lines.  A binary that prints 12 keeps it.")


if __name__ == "__main__":
    main()

"""GPU tests of the device BGZF inflate (include/mdk_hip.h md_piece_*, csrc/mdk_inflate.hip):
  * C-ABI level: every member of a piece inflates byte-identical to zlib, the record table equals a walk of the records, the
    member digests equal what a walk computes (this is what csrc/host/mdk_io.c note_records leaves for host-inflated slabs);
  * command level: `MethylDackel extract` with every piece but the header's inflated on the device (MDK_DEVICE_INFLATE_ONLY=1)
    must produce the oracle's bytes, over chunk sizes far below a member's span (a member then serves many chunks), regions
    through the index, dense contexts, --mergeContext, a file whose records straddle BGZF members (read back to the host) and a
    chunk the device preparation hands back to the host (its records exist only on the device)."""
import ctypes as C
import struct
import zlib

import pytest

import methyldackel_amd as mdk
from conftest import REPO, synth
from test_gpu_parity import compare_cli

pytestmark = pytest.mark.gpu


CRCS = {}            # deflate stream offset -> CRC32 of the member's trailer (filled by bgzf_members)


def bgzf_members(raw):
    out, o = [], 0
    while o + 18 <= len(raw):
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        bs = struct.unpack_from("<H", raw, o + 16)[0] + 1
        isz = struct.unpack_from("<I", raw, o + bs - 4)[0]
        out.append((o + 12 + xlen, bs - 12 - xlen - 8, isz))
        CRCS[o + 12 + xlen] = struct.unpack_from("<I", raw, o + bs - 8)[0]
        o += bs
    return out


def walk(data):
    """records of one inflated member -> (offsets, digest) or None when it does not start and end on record boundaries"""
    offs, o, L = [], 0, len(data)
    tid0 = pos0 = tidN = posN = -1; mn, mx, srt = 0x7fffffff, -0x80000000, 1
    while o + 4 <= L:
        bs = struct.unpack_from("<I", data, o)[0]
        if bs < 32 or o + 4 + bs > L:
            return None
        tid, pos = struct.unpack_from("<ii", data, o + 4)
        lq, nc = data[o + 12], struct.unpack_from("<H", data, o + 16)[0]
        if 32 + lq + 4 * nc > bs:
            return None
        rl = 0
        for k in range(nc):
            v = struct.unpack_from("<I", data, o + 36 + lq + 4 * k)[0]
            if (v & 15) in (0, 2, 3, 7, 8):
                rl += v >> 4
        endp = pos + (rl if rl > 0 else 1)
        if not offs:
            tid0, pos0 = tid, pos
        elif tid < 0 or tid < tidN or (tid == tidN and pos < posN):
            srt = 0
        if tid < 0:
            srt = 0
        tidN, posN = tid, pos; mn, mx = min(mn, endp), max(mx, endp)
        offs.append(o); o += 4 + bs
    if o != L:
        return None
    return offs, (len(offs), tid0, pos0, tidN, posN, mn, mx, srt)


def test_piece_inflate_equals_zlib_and_record_walk(tmp_path):
    synth(tmp_path / "s", "-L", "400000", "-c", "30", "-s", "77", "--extras")
    raw = (tmp_path / "s.bam").read_bytes()
    mem = bgzf_members(raw)
    L = mdk.lib_hip()
    cfg = mdk.md_dev_cfg(); cfg.keepCpG = 1; cfg.minPhred = 5
    dev = mdk.Device(cfg)
    L.md_host_alloc.restype = C.c_void_p
    stage = L.md_host_alloc(C.c_uint64(len(raw) + 64)); C.memmove(stage, raw, len(raw))
    tab = (mdk.md_inf_member * len(mem))(); o = 0
    for i, (io, il, isz) in enumerate(mem):
        tab[i].in_off, tab[i].in_len, tab[i].out_len, tab[i].out_off, tab[i].crc32 = io, il, isz, o, CRCS[io]; o += isz
    piece = C.c_void_p(); info = mdk.md_piece_info()
    assert L.md_piece_create(dev.h, C.byref(piece)) == 0, L.md_dev_last_error()
    for variant_pass in range(2):       # twice: the second submit reuses the piece's buffers
        assert L.md_piece_submit(piece, C.c_void_p(stage), len(raw), tab, len(mem)) == 0, L.md_dev_last_error()
        assert L.md_piece_wait(piece, C.byref(info)) == 0, L.md_dev_last_error()
        assert info.n_mem == len(mem) and info.out_bytes == o
        got = C.create_string_buffer(o)
        assert L.md_piece_read(piece, 0, o, got) == 0
        recs = (C.c_uint32 * max(1, info.n_records))()
        assert L.md_piece_read_records(piece, 0, info.n_records, recs) == 0
        first = 0
        for i, (io, il, isz) in enumerate(mem):
            ref = zlib.decompress(raw[io:io + il], wbits=-15) if isz else b""
            assert len(ref) == isz and got.raw[tab[i].out_off:tab[i].out_off + isz] == ref, f"member {i}"
            w = walk(ref); g = info.digest[i]
            assert g.first_rec == first
            if w is None:               # the member holding the BAM header (and the records behind it in the same member)
                assert g.ok == 0 and g.n_rec == 0
                continue
            assert g.ok == 1
            offs, dg = w
            assert (g.n_rec, ) + ((g.tid0, g.pos0, g.tidN, g.posN, g.min_endp, g.max_endp, g.sorted) if offs else ()) == (dg[0], ) + (dg[1:] if offs else ())
            assert [recs[first + k] for k in range(len(offs))] == [tab[i].out_off + x for x in offs]
            first += len(offs)
        assert first == info.n_records
    L.md_piece_destroy(piece); L.md_host_free(C.c_void_p(stage)); dev.close()


DEV = {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "3", "MDK_HOST_PROFILE": "1"}      # test hooks (csrc/host/mdk_io.c): the host teams leave after the header, small device pieces


@pytest.fixture(scope="module")
def mid_synth(tmp_path_factory):
    d = tmp_path_factory.mktemp("inflate")
    synth(d / "pe", "-L", "1500000,700000", "-c", "25", "-s", "31", "--extras")
    return d


@pytest.mark.parametrize("extra", [[], ["--chunkSize", "7000"], ["--chunkSize", "250000", "--CHG", "--CHH"], ["--mergeContext", "--chunkSize", "90000"],
                                   ["-r", "chr2:100000-420000"], ["--minOppositeDepth", "3", "--maxVariantFrac", "0.2", "--chunkSize", "333333"], ["--keepDupes", "--keepSingleton", "--keepDiscordant", "-q", "0"]])
def test_cli_with_device_inflate_equals_oracle(mid_synth, tmp_path, extra):
    od, gd = compare_cli(tmp_path, [str(mid_synth / "pe.fa"), str(mid_synth / "pe.bam")] + extra, env=DEV)


def test_device_pieces_were_used(mid_synth, tmp_path):
    r = mdk.run_cli([str(mid_synth / "pe.fa"), str(mid_synth / "pe.bam"), "-o", "out"], cwd=tmp_path, env=DEV)
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stderr.splitlines() if "on the device" in l]
    assert line, r.stderr[-2000:]
    n_dev = int(line[0].split("on the device ")[1].split()[0])
    assert n_dev >= 1


def test_records_straddling_members_are_read_back(tmp_path):
    """a BAM whose BGZF members are cut at arbitrary bytes (legal BGZF, never written by htslib): the device's walk reports members that
    do not start on a record, the slab comes back to the host and is scanned there"""
    synth(tmp_path / "s", "-L", "300000", "-c", "20", "-s", "5")
    raw = (tmp_path / "s.bam").read_bytes()
    data = b"".join(zlib.decompress(raw[io:io + il], wbits=-15) for io, il, isz in bgzf_members(raw) if isz)
    out, o = bytearray(), 0
    while o < len(data):
        blk = data[o:o + 40001]; o += len(blk)
        c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(blk) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(blk), len(blk))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    (tmp_path / "cut.bam").write_bytes(bytes(out))
    compare_cli(tmp_path, [str(tmp_path / "s.fa"), str(tmp_path / "cut.bam")], env=DEV)


def test_chunk_handed_back_to_the_host_with_device_resident_records(tmp_path):
    """20 records of one read name in a chunk: the device preparation gives the chunk up (MDK_ERR_PREP_HOST); its records were
    inflated on the device, so they are read back for the host preparation"""
    import random
    from bamwriter import record, write_fasta
    rnd = random.Random(3)
    ref = "".join(rnd.choice("ACGT") for _ in range(30000))
    write_fasta(tmp_path / "r.fa", [("chrA", ref)])
    recs = []
    for i in range(6000):
        pos = rnd.randrange(0, 29800)
        dup = i % 300 == 0          # 20 records of one name, paired flags: the name's chain is walked (a lane keeps 16) -> the chunk goes back to the host
        recs.append((pos, record(0, pos, 0x43 if dup else 0, "100M", ref[pos:pos + 100], 30, qname="dup" if dup else f"r{i}")))
    recs.sort(key=lambda r: r[0])
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrA\tLN:30000\n"
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chrA\0" + struct.pack("<i", 30000)
    out, blk = bytearray(), bytearray()

    def flush():
        nonlocal blk
        if blk:
            c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(bytes(blk)) + c.flush()
            out.extend(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(bytes(blk)), len(blk)))
            blk = bytearray()
    blk += hdr; flush()                 # the header in a member of its own, then members that end on record boundaries
    for _, r in recs:
        if len(blk) + len(r) > 50000:
            flush()
        blk += r
    flush()
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    (tmp_path / "r.bam").write_bytes(bytes(out))
    od, gd = compare_cli(tmp_path, [str(tmp_path / "r.fa"), str(tmp_path / "r.bam"), "-q", "0"], env=dict(DEV, MDK_HOST_PROFILE="1"))
    assert fell_back(tmp_path) >= 1, "the chunk was expected to fall back to the host preparation"


def fell_back(tmp_path):
    """chunks the command prepared on the host after all (MDK_HOST_PROFILE line of extract_main)"""
    import re
    m = re.search(r"chunks prepared on the host after all: (\d+)", (tmp_path / "gpu_stderr.txt").read_text())
    return int(m.group(1)) if m else -1


def test_handed_back_chunks_drop_the_neighbours_records(tmp_path):
    """The same with two contigs and chunks smaller than a BGZF member's span: the members read back for a handed-back chunk also hold
    the records of the neighbouring chunks and -- the member that straddles the contig boundary -- of the other contig; the host
    preparation must redo the chunk's region query on them (chrB is short: its single chunk starts at 0, where chrA's reads would
    otherwise be counted at chrB's coordinates)."""
    import random
    from bamwriter import record, write_fasta
    rnd = random.Random(5)
    refs = [("chrA", "".join(rnd.choice("ACGT") for _ in range(30000))), ("chrB", "".join(rnd.choice("ACGT") for _ in range(2500)))]
    write_fasta(tmp_path / "r.fa", refs)
    recs = []
    for tid, (_, ref) in enumerate(refs):
        n = 6000 if tid == 0 else 900
        for i in range(n):
            pos = rnd.randrange(0, len(ref) - 120)
            dup = i % 10 == 0
            recs.append((tid, pos, record(tid, pos, 0x43 if dup else 0, "100M", ref[pos:pos + 100], 30, qname=f"dup{tid}" if dup else f"r{tid}_{i}")))
    recs.sort(key=lambda r: (r[0], r[1]))
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrA\tLN:30000\n@SQ\tSN:chrB\tLN:2500\n"
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 2) + struct.pack("<i", 5) + b"chrA\0" + struct.pack("<i", 30000) + struct.pack("<i", 5) + b"chrB\0" + struct.pack("<i", 2500)
    out, blk = bytearray(), bytearray()

    def flush():
        nonlocal blk
        if blk:
            c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(bytes(blk)) + c.flush()
            out.extend(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(bytes(blk)), len(blk)))
            blk = bytearray()
    blk += hdr; flush()
    for _, _, r in recs:                  # members end on record boundaries, wherever the contig changes
        if len(blk) + len(r) > 50000:
            flush()
        blk += r
    flush()
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    (tmp_path / "r.bam").write_bytes(bytes(out))
    od, gd = compare_cli(tmp_path, [str(tmp_path / "r.fa"), str(tmp_path / "r.bam"), "-q", "0", "--chunkSize", "2000"], env=dict(DEV, MDK_HOST_PROFILE="1"))
    assert fell_back(tmp_path) >= 2


def stored_members_bam(raw, flip_member=None):
    """the same BAM with every member re-written as STORED deflate blocks (level 0); optionally one payload byte of one member flipped, the
    member's CRC32 and ISIZE left as they were: it still inflates to ISIZE bytes, but not to the bytes the trailer vouches for"""
    out = bytearray()
    for k, (io, il, isz) in enumerate(bgzf_members(raw)):
        data = zlib.decompress(raw[io:io + il], wbits=-15) if isz else b""
        c = zlib.compressobj(0, zlib.DEFLATED, -15); comp = bytearray(c.compress(data) + c.flush())
        if flip_member is not None and k == flip_member:
            assert len(comp) > 40 and isz
            comp[20] ^= 0x5a                  # (5 bytes of stored-block header, then the bytes themselves)
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + bytes(comp) + struct.pack("<II", zlib.crc32(data), len(data))
    return bytes(out)


def test_a_byte_flipped_inside_a_stored_block_fails_the_crc32_check(tmp_path):
    """What htslib verifies behind sam_itr_next (bgzf_read_block: CRC32 and ISIZE of every block): a damaged byte that raw inflate cannot
    notice must stop the run on the host's inflate path and on the device's (k_crc32), and stops the oracle; the undamaged file passes"""
    synth(tmp_path / "s", "-L", "120000", "-c", "20", "-s", "9")
    raw = (tmp_path / "s.bam").read_bytes()
    (tmp_path / "ok.bam").write_bytes(stored_members_bam(raw))
    (tmp_path / "bad.bam").write_bytes(stored_members_bam(raw, flip_member=7))
    compare_cli(tmp_path, [str(tmp_path / "s.fa"), str(tmp_path / "ok.bam")], env=DEV)
    from conftest import run_oracle
    (tmp_path / "o2").mkdir()
    assert run_oracle([str(tmp_path / "s.fa"), str(tmp_path / "bad.bam"), "-o", "out"], cwd=tmp_path / "o2").returncode != 0
    for name, env in (("device", DEV), ("host", {"MDK_HOST_INFLATE": "1"})):
        d = tmp_path / name; d.mkdir()
        r = mdk.run_cli([str(tmp_path / "s.fa"), str(tmp_path / "bad.bam"), "-o", "out"], cwd=d, env=env)
        assert r.returncode != 0 and "CRC32" in r.stderr, (name, r.returncode, r.stderr[-800:])
        r = mdk.run_cli([str(tmp_path / "s.fa"), str(tmp_path / "bad.bam"), "-o", "out"], cwd=d, env=dict(env, MDK_NO_CRC="1"))
        assert r.returncode == 0, (name, "without the check the damage goes unnoticed", r.stderr[-800:])

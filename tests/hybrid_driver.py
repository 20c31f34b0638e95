"""Driver of tests/test_hybrid_chunks.py (run in a process of its own, with tools/_build/libmdk_piece_standin.so preloaded when the
"device" is wanted): one `extract` plan in device-preparation mode, every chunk it hands out printed as index, interval, number of
records and a digest of the records in order.  A range that lies in device memory is a run of whole BGZF members -- at a chunk's edges
it holds records of the neighbouring chunk or contig too, which the scan kernel drops again with the region query it redoes per record
(same contig, pos < end, bam_endpos > beg: csrc/mdk_prep.hip k_prep_scan) --, so that query is applied here as well; on host ranges it
must be a no-op (tests/test_raw_batch.py).  usage: hybrid_driver.py ATTACH(0|1) [extract arguments]"""
import ctypes as C
import hashlib
import json
import struct
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import methyldackel_amd as mdk

attach = int(sys.argv[1])
plan = mdk.Plan(sys.argv[2:])
plan.set_prep(1)
if attach:      # any non-null handle will do: the stand-in never looks at it, and nothing else of the device library is called in this mode
    fake = C.create_string_buffer(4096)
    assert plan.L.mdk_plan_attach_device(plan.p, C.cast(fake, C.c_void_p)) == 0
n_dev_ranges = 0


def in_region(rec, c):          # rec: block_size word + body
    tid, pos, lrn, mq, bn, nc = struct.unpack_from("<iiBBHH", rec, 4)
    cig = struct.unpack_from("<%dI" % nc, rec, 4 + 32 + lrn)
    rlen = sum(v >> 4 for v in cig if (v & 15) in (0, 2, 3, 7, 8))
    return tid == c.tid and pos < c.end and pos + max(rlen, 1) > c.beg


while (c := plan.next_chunk()) is not None:
    if c.skipped:
        print(json.dumps({"index": c.index, "tid": c.tid, "beg": c.beg, "end": c.end, "skipped": int(c.skipped)}))
        continue
    raw = c.raw
    any_dev = any(bool(raw.range[i].d_rec_off) for i in range(raw.n_ranges))
    h = hashlib.sha1(); n = 0; o = 0; hidx = 0; host_bytes = []
    if not any_dev:
        cat = b"".join(C.string_at(raw.range[i].ptr, raw.range[i].bytes) for i in range(raw.n_ranges))
        offs = mdk.raw_record_offsets(raw); assert len(offs) == raw.n_records
        for off in offs:
            bs, = struct.unpack_from("<I", cat, off)
            assert in_region(cat[off:off + 4 + bs], c), "host ranges hold the region query's records only"
            h.update(cat[off:off + 4 + bs]); n += 1
        assert sum(raw.range[i].bytes for i in range(raw.n_ranges)) == len(cat)
    else:
        for i in range(raw.n_ranges):
            r = raw.range[i]
            data = C.string_at(r.ptr, r.bytes) if r.bytes else b""
            covered = 0
            for k in range(r.n_records):
                off = (r.d_rec_off[k] - r.rec_delta) if r.d_rec_off else (r.h_rec_off[k] - r.rec_delta) if r.h_rec_off else (raw.rec_off[hidx + k] - o)
                bs, = struct.unpack_from("<I", data, off)
                covered += 4 + bs
                if in_region(data[off:off + 4 + bs], c):
                    h.update(data[off:off + 4 + bs]); n += 1
                else:
                    assert r.d_rec_off, "only a device range (whole members) may hold records outside the chunk"
            assert covered == r.bytes, "a range holds exactly its records"
            if r.d_rec_off:
                n_dev_ranges += 1
            elif not r.h_rec_off:
                hidx += r.n_records
            o += r.bytes
    print(json.dumps({"index": c.index, "tid": c.tid, "beg": c.beg, "end": c.end, "n": n, "sha1": h.hexdigest()}))
print(json.dumps({"device_ranges": n_dev_ranges}))
if attach:
    plan.L.mdk_plan_detach_device(plan.p)
plan.close()

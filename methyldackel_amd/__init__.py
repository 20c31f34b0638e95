"""methyldackel_amd -- MI355X-native `MethylDackel extract` hot path.

The product is native code: ``csrc/mdk_hip.hip`` (HIP kernels + device C-ABI, ``include/mdk_hip.h``) and
``csrc/host/*.c`` (C host: BGZF/BAM decode, admission, pairing, chunk schedule, text emitters;
``include/mdk_extract.h``).  This package is only the ctypes view of those two C-ABIs that the tests, bench.py
and the multi-GPU driver use; it contains no compute and no fallback -- importing works anywhere, but every
device call fails loudly when ``libmdk_hip.so`` is missing or no GPU is visible.

Reference interface mirrored: ``extract_main(argc, argv)`` (reference extract.c:706) and the per-chunk
pipeline of ``extractCalls`` (reference extract.c:247-560).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent
REPO = ROOT.parent
BUILD = Path(os.environ["MDK_BUILD_DIR"]) if os.environ.get("MDK_BUILD_DIR") else ROOT / "_build"      # MDK_BUILD_DIR: experiment builds (tools/kbench.py)
LIB_HIP = BUILD / "libmdk_hip.so"
LIB_EXTRACT = BUILD / "libmdk_extract.so"
CLI = BUILD / "MethylDackel"

CHUNK_NOREF, CHUNK_FOREIGN, CHUNK_BED = 1, 2, 4
CHUNK_EMPTY = CHUNK_NOREF | CHUNK_BED      # passed over by every rank: nothing is packed and nothing is emitted
MDK_ERR = {-1: "HIP call failed", -2: "no device", -3: "bad argument", -4: "reference not uploaded",
           -5: "strand 0 read reached a call", -6: "out of memory"}


class MdkError(RuntimeError):
    pass


class md_dev_cfg(C.Structure):
    _fields_ = [("keepCpG", C.c_int32), ("keepCHG", C.c_int32), ("keepCHH", C.c_int32), ("minPhred", C.c_int32),
                ("minOppositeDepth", C.c_int32), ("bounds", C.c_int32 * 16), ("absoluteBounds", C.c_int32 * 16),
                ("tile", C.c_int32), ("n_slots", C.c_int32), ("n_streams", C.c_int32)]


class md_seg(C.Structure):
    _fields_ = [("rpos", C.c_int32), ("off4", C.c_uint32), ("l_qseq", C.c_uint32), ("q0", C.c_uint32), ("len", C.c_uint16),
                ("sf", C.c_uint8), ("msf", C.c_uint8), ("m_off4", C.c_uint32), ("m_l_qseq", C.c_uint32), ("m_q0", C.c_uint32)]


class md_read_batch(C.Structure):
    _fields_ = [("tid", C.c_int32), ("beg", C.c_int64), ("end", C.c_int64), ("n_segs", C.c_int32),
                ("seg", C.POINTER(md_seg)), ("blob", C.POINTER(C.c_uint8)), ("blob_bytes", C.c_uint64),
                ("n_reads", C.c_int32), ("algo_bytes", C.c_uint64)]


class md_prep_cfg(C.Structure):
    _fields_ = [("min_mapq", C.c_int32), ("ignore_flags", C.c_int32), ("require_flags", C.c_int32), ("keep_dupes", C.c_int32), ("ignore_nh", C.c_int32),
                ("keep_singleton", C.c_int32), ("keep_discordant", C.c_int32), ("min_phred", C.c_int32), ("min_conv_eff", C.c_float),
                ("map_on", C.c_int32), ("min_mappable", C.c_int32), ("no_pairing", C.c_int32), ("perread", C.c_int32)]


class md_raw_range(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_uint8)), ("bytes", C.c_uint64), ("d_rec_off", C.POINTER(C.c_uint32)), ("n_records", C.c_uint32), ("rec_delta", C.c_uint32), ("h_rec_off", C.POINTER(C.c_uint32))]


class md_raw_batch(C.Structure):
    _fields_ = [("tid", C.c_int32), ("beg", C.c_int64), ("end", C.c_int64), ("n_ranges", C.c_int32), ("range", C.POINTER(md_raw_range)),
                ("n_records", C.c_int32), ("rec_off", C.POINTER(C.c_uint32)), ("woff", C.c_int64), ("wlen", C.c_int64)]


class md_inf_member(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint32), ("out_len", C.c_uint32), ("out_off", C.c_uint64), ("crc32", C.c_uint32), ("reserved", C.c_uint32)]


class md_inf_digest(C.Structure):
    _fields_ = [("n_rec", C.c_uint32), ("first_rec", C.c_uint32), ("tid0", C.c_int32), ("pos0", C.c_int32), ("tidN", C.c_int32), ("posN", C.c_int32),
                ("min_endp", C.c_int32), ("max_endp", C.c_int32), ("ok", C.c_int32), ("sorted", C.c_int32)]


class md_piece_info(C.Structure):
    _fields_ = [("n_mem", C.c_int32), ("digest", C.POINTER(md_inf_digest)), ("n_records", C.c_uint32), ("out_bytes", C.c_uint64),
                ("d_out", C.c_void_p), ("d_rec_off", C.c_void_p)]


class md_region(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("strand", C.c_int32)]


class md_pr_read(C.Structure):
    _fields_ = [("pos", C.c_int32), ("off4", C.c_uint32), ("l_qseq", C.c_uint32), ("cig_off", C.c_uint32), ("n_cigar", C.c_uint16),
                ("strand", C.c_uint8), ("reserved", C.c_uint8)]


class md_pr_batch(C.Structure):
    _fields_ = [("tid", C.c_int32), ("beg", C.c_int64), ("end", C.c_int64), ("n_reads", C.c_int32), ("read", C.POINTER(md_pr_read)),
                ("cigar", C.POINTER(C.c_uint32)), ("n_cigar", C.c_uint64), ("blob", C.POINTER(C.c_uint8)), ("blob_bytes", C.c_uint64)]


class md_pr_count(C.Structure):
    _fields_ = [("nmeth", C.c_uint32), ("nunmeth", C.c_uint32)]


class md_mbias(C.Structure):
    _fields_ = [("len", C.c_int32), ("count", C.POINTER(C.c_uint32))]


class md_site(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("nmeth", C.c_uint32), ("nunmeth", C.c_uint32), ("meta", C.c_uint32)]


class md_site_var(C.Structure):
    _fields_ = [("noff", C.c_uint32), ("nvar", C.c_uint32)]


class md_tile_seg(C.Structure):
    _fields_ = [("off", C.c_uint32), ("cnt", C.c_uint32)]


class md_sites(C.Structure):
    _fields_ = [("n_sites", C.c_int64), ("site", C.POINTER(md_site)), ("var", C.POINTER(md_site_var))]


class md_sites_dev(C.Structure):
    _fields_ = [("n_slots", C.c_int64), ("n_tiles", C.c_int32), ("d_site", C.c_void_p), ("d_var", C.c_void_p), ("d_seg", C.c_void_p)]


class md_bench_result(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_pileup", C.c_float), ("algo_bytes", C.c_uint64), ("n_sites", C.c_uint64),
                ("tile", C.c_int32), ("n_tiles", C.c_int32), ("lds_bytes", C.c_int32)]


class mdk_chunk(C.Structure):
    _fields_ = [("index", C.c_uint32), ("tid", C.c_int32), ("beg", C.c_int64), ("end", C.c_int64), ("skipped", C.c_int32),
                ("batch", md_read_batch), ("n_records_seen", C.c_uint64), ("pr", md_pr_batch), ("host", C.c_void_p),
                ("prep", C.c_int32), ("raw", md_raw_batch)]


class md_bench_run_result(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("slots_last", C.c_uint64), ("exchanges", C.c_uint64), ("bytes_per_exchange", C.c_uint64)]


COMM_ID_BYTES = 128
md_comm_oob_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)      # out-of-band all-gather for md_comm_open_rank_shared

def raw_record_offsets(raw):
    """offset, in the concatenation of the batch's ranges, of every record of an md_raw_batch whose ranges all lie in HOST memory: from each
    range's own table (h_rec_off) or, for ranges without one, from the batch's rec_off array (include/mdk_hip.h md_raw_range)"""
    out, o, hidx = [], 0, 0
    tables = any(bool(raw.range[i].h_rec_off) or bool(raw.range[i].d_rec_off) for i in range(raw.n_ranges))
    if not tables:
        return [raw.rec_off[i] for i in range(raw.n_records)]
    for i in range(raw.n_ranges):
        r = raw.range[i]
        if r.d_rec_off:
            raise MdkError("raw_record_offsets: a range lies in device memory")
        if r.h_rec_off:
            out += [r.h_rec_off[k] - r.rec_delta + o for k in range(r.n_records)]
        else:
            out += [raw.rec_off[hidx + k] for k in range(r.n_records)]; hidx += r.n_records
        o += r.bytes
    return out


HIP_SYMBOLS = ["md_dev_count", "md_dev_warm", "md_dev_quiesce", "md_dev_reserve_hint", "md_dev_open", "md_dev_close", "md_dev_last_error", "md_dev_tile", "md_dev_set_reference", "md_dev_set_regions",
               "md_dev_upload", "md_dev_launch", "md_dev_submit", "md_dev_download", "md_dev_sync", "md_dev_bind_output", "md_dev_wait", "md_sites_order",
               "md_dev_bench", "md_dev_bench_rotate", "md_dev_launch_group", "md_dev_group_max", "md_dev_download_group", "md_dev_reserve_contigs", "md_comm_unique_id", "md_comm_open_rank", "md_comm_open_rank_shared", "md_comm_close", "md_comm_world", "md_comm_gather", "md_comm_wait", "md_comm_result_header", "md_comm_result_send", "md_comm_result_recv", "md_dev_pci_bus_id",
               "md_bench_open", "md_bench_run", "md_bench_verify", "md_bench_region_bytes", "md_bench_close", "md_dev_debug_effective", "md_host_alloc", "md_host_free", "md_host_set_pinned", "md_host_profile", "md_dev_profile_text", "md_host_register", "md_host_register_all",
               "md_dev_set_prep", "md_dev_set_mappability", "md_dev_upload_raw", "md_dev_upload_raw_inplace", "md_dev_upload_wait", "md_dev_upload_done", "md_dev_submit_raw", "md_dev_debug_segments", "md_dev_bench_prep", "md_dev_bench_prep_rotate", "md_bench_set_prep",
               "md_dev_mbias_submit", "md_dev_mbias_submit_raw", "md_dev_mbias_read", "md_dev_mbias_reset", "md_dev_slot_sync",
               "md_dev_perread_submit", "md_dev_perread_download", "md_dev_perread_submit_raw", "md_dev_perread_download_raw", "md_dev_read_raw",
               "md_piece_members_per_round", "md_piece_create", "md_piece_destroy", "md_piece_submit", "md_piece_wait", "md_piece_read", "md_piece_read_records", "md_piece_bench", "md_piece_bench_crc"]
EXTRACT_SYMBOLS = ["extract_main", "mdk_plan_open", "mdk_plan_close", "mdk_plan_dev_cfg", "mdk_plan_ensure_reference",
                   "mdk_plan_next_chunk", "mdk_plan_try_next_chunk", "mdk_plan_emit", "mdk_plan_finish", "mdk_plan_set_shard", "mdk_plan_n_targets", "mdk_plan_target_name",
                   "mdk_plan_target_len", "mdk_plan_regions", "mdk_plan_set_prep", "mdk_plan_set_hold", "mdk_plan_prep_cfg", "mdk_plan_host_prepare",
                   "mdk_plan_host_prepare_from", "mdk_plan_release_records", "mdk_plan_attach_device", "mdk_plan_detach_device",
                   "mbias_main", "mdk_cli_quiesce", "mdk_plan_open_mbias", "mdk_plan_mbias_outputs", "mdk_mbias_report",
                   "perRead_main", "mdk_plan_open_perread", "mdk_plan_emit_perread", "mdk_plan_emit_perread_raw", "mergeContext_main", "mdk_bind_to_device_node"]

_hip = None
_ext = None


def build(verbose: bool = False) -> None:
    """Compile everything in-tree (hipcc --offload-arch=gfx950 for the kernels, gcc for the host)."""
    r = subprocess.run(["make", "-C", str(REPO), "all"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise MdkError("build failed")


def lib_hip():
    global _hip
    if _hip is None:
        if not LIB_HIP.exists():
            raise MdkError(f"{LIB_HIP} is missing: the HIP extension was not built (run `make` / __graft_entry__.build()); there is no fallback")
        L = C.CDLL(str(LIB_HIP), mode=C.RTLD_GLOBAL)
        L.md_dev_last_error.restype = C.c_char_p
        L.md_dev_open.argtypes = [C.c_int, C.POINTER(md_dev_cfg), C.POINTER(C.c_void_p)]
        L.md_dev_close.argtypes = [C.c_void_p]
        L.md_dev_tile.argtypes = [C.c_void_p]
        L.md_dev_set_reference.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]
        L.md_dev_set_regions.argtypes = [C.c_void_p, C.c_int32, C.POINTER(md_region), C.c_int64]
        for f in ("md_dev_upload", "md_dev_submit"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.POINTER(md_read_batch)]
        L.md_dev_launch.argtypes = [C.c_void_p, C.c_int]
        L.md_dev_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_sites)]
        L.md_dev_sync.argtypes = [C.c_void_p]
        L.md_dev_bind_output.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.md_dev_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_sites_dev)]
        L.md_sites_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
        L.md_sites_order.restype = C.c_int64
        L.md_dev_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(md_bench_result)]
        L.md_dev_bench_rotate.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(md_bench_result)]
        L.md_dev_launch_group.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.md_comm_unique_id.argtypes = [C.c_char_p]
        L.md_comm_open_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.md_comm_open_rank_shared.argtypes = [C.c_void_p, C.c_int, C.c_int, md_comm_oob_fn, C.c_void_p, C.POINTER(C.c_void_p)]
        L.md_bench_set_prep.argtypes = [C.c_void_p, C.c_int]
        L.md_dev_bench_prep_rotate.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.md_comm_close.argtypes = [C.c_void_p]; L.md_comm_close.restype = None
        L.md_comm_world.argtypes = [C.c_void_p]
        L.md_comm_gather.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.md_comm_wait.argtypes = [C.c_void_p]
        L.md_bench_open.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.md_bench_run.argtypes = [C.c_void_p, C.c_int64, C.POINTER(md_bench_run_result)]
        L.md_bench_verify.argtypes = [C.c_void_p]
        L.md_bench_region_bytes.argtypes = [C.c_void_p]; L.md_bench_region_bytes.restype = C.c_int64
        L.md_bench_close.argtypes = [C.c_void_p]; L.md_bench_close.restype = None
        L.md_dev_debug_effective.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.md_dev_set_prep.argtypes = [C.c_void_p, C.POINTER(md_prep_cfg)]
        L.md_dev_set_mappability.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.md_dev_upload_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_raw_batch)]
        L.md_dev_submit_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_raw_batch)]
        L.md_dev_bench_prep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.md_dev_debug_segments.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_seg), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.md_dev_mbias_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_read_batch)]
        L.md_dev_mbias_submit_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_raw_batch)]
        L.md_dev_mbias_read.argtypes = [C.c_void_p, C.POINTER(md_mbias)]
        L.md_dev_mbias_reset.argtypes = [C.c_void_p]
        L.md_dev_slot_sync.argtypes = [C.c_void_p, C.c_int]
        L.md_dev_perread_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_pr_batch)]
        L.md_dev_perread_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(md_pr_count)), C.POINTER(C.c_int64)]
        L.md_dev_perread_submit_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(md_raw_batch)]
        L.md_dev_read_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint32)]
        L.md_piece_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.md_piece_destroy.argtypes = [C.c_void_p]; L.md_piece_destroy.restype = None
        L.md_piece_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(md_inf_member), C.c_int32]
        L.md_piece_wait.argtypes = [C.c_void_p, C.POINTER(md_piece_info)]
        L.md_piece_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.md_piece_read_records.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.md_piece_bench.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.md_dev_perread_download_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(md_pr_count)), C.POINTER(C.c_int64)]
        L.md_host_alloc.restype = C.c_void_p
        L.md_host_alloc.argtypes = [C.c_uint64]
        L.md_host_free.argtypes = [C.c_void_p]
        _hip = L
    return _hip


def lib_extract():
    global _ext
    if _ext is None:
        lib_hip()
        if not LIB_EXTRACT.exists():
            raise MdkError(f"{LIB_EXTRACT} is missing (run `make`)")
        L = C.CDLL(str(LIB_EXTRACT))
        L.extract_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.mdk_plan_open.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        L.mdk_plan_close.argtypes = [C.c_void_p]
        L.mdk_plan_dev_cfg.argtypes = [C.c_void_p, C.POINTER(md_dev_cfg)]
        L.mdk_plan_ensure_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.mdk_plan_next_chunk.argtypes = [C.c_void_p, C.POINTER(mdk_chunk)]
        L.mdk_plan_emit.argtypes = [C.c_void_p, C.POINTER(mdk_chunk), C.POINTER(md_sites)]
        L.mdk_plan_finish.argtypes = [C.c_void_p]
        L.mdk_plan_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mdk_plan_n_targets.argtypes = [C.c_void_p]
        L.mdk_plan_target_name.argtypes = [C.c_void_p, C.c_int32]
        L.mdk_plan_target_name.restype = C.c_char_p
        L.mdk_plan_target_len.argtypes = [C.c_void_p, C.c_int32]
        L.mdk_plan_target_len.restype = C.c_int64
        L.mbias_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.mdk_plan_open_mbias.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        L.mdk_plan_mbias_outputs.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mdk_mbias_report.argtypes = [C.POINTER(md_mbias), C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.perRead_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.mergeContext_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.mdk_plan_open_perread.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        L.mdk_plan_emit_perread.argtypes = [C.c_void_p, C.POINTER(mdk_chunk), C.POINTER(md_pr_count), C.c_int64]
        L.mdk_plan_emit_perread_raw.argtypes = [C.c_void_p, C.POINTER(mdk_chunk), C.POINTER(C.c_uint32), C.POINTER(md_pr_count), C.c_int64]
        L.mdk_plan_regions.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(md_region)), C.POINTER(C.c_int64)]
        L.mdk_plan_set_prep.argtypes = [C.c_void_p, C.c_int]
        L.mdk_plan_prep_cfg.argtypes = [C.c_void_p, C.POINTER(md_prep_cfg)]; L.mdk_plan_prep_cfg.restype = None
        L.mdk_plan_host_prepare.argtypes = [C.c_void_p, C.POINTER(mdk_chunk)]
        L.mdk_plan_host_prepare_from.argtypes = [C.c_void_p, C.POINTER(mdk_chunk), C.c_void_p, C.c_int]
        L.mdk_plan_attach_device.argtypes = [C.c_void_p, C.c_void_p]
        L.mdk_plan_detach_device.argtypes = [C.c_void_p]; L.mdk_plan_detach_device.restype = None
        _ext = L
    return _ext


def _argv(args):
    arr = (C.c_char_p * (len(args) + 1))()
    for i, a in enumerate(args):
        arr[i] = os.fsencode(str(a))
    return arr


class Device:
    """One GPU handle (md_dev_open).  Raises MdkError when no device / no HIP library: there is no CPU path."""

    def __init__(self, cfg: md_dev_cfg, device: int = 0):
        L = lib_hip()
        self.h = C.c_void_p()
        rc = L.md_dev_open(device, C.byref(cfg), C.byref(self.h))
        if rc:
            raise MdkError(f"md_dev_open failed ({rc}): {L.md_dev_last_error().decode()}")
        self.L = L

    def _chk(self, rc, what):
        if rc:
            raise MdkError(f"{what} failed ({rc}): {self.L.md_dev_last_error().decode()}")

    def set_reference(self, tid: int, seq: bytes):
        self._chk(self.L.md_dev_set_reference(self.h, tid, seq, len(seq)), "md_dev_set_reference")

    def set_regions(self, tid: int, runs):
        """runs: [(start, end, strand)], sorted and disjoint (-l/--keepStrand)"""
        arr = (md_region * max(len(runs), 1))(*[md_region(*r) for r in runs])
        self._chk(self.L.md_dev_set_regions(self.h, tid, arr, len(runs)), "md_dev_set_regions")

    def submit(self, slot: int, batch: md_read_batch):
        self._chk(self.L.md_dev_submit(self.h, slot, C.byref(batch)), "md_dev_submit")

    def set_prep(self, cfg: "md_prep_cfg"):
        self._chk(self.L.md_dev_set_prep(self.h, C.byref(cfg)), "md_dev_set_prep")

    def upload_raw(self, slot: int, raw: "md_raw_batch"):
        """H2D of a chunk's BAM records + the device preparation (admission, pairing, segments)"""
        self._chk(self.L.md_dev_upload_raw(self.h, slot, C.byref(raw)), "md_dev_upload_raw")

    def submit_raw(self, slot: int, raw: "md_raw_batch"):
        self._chk(self.L.md_dev_submit_raw(self.h, slot, C.byref(raw)), "md_dev_submit_raw")

    def debug_segments(self, slot: int):
        """-> (list of md_seg as the device preparation built them, number of admitted reads)"""
        n, nr = C.c_int64(), C.c_int64()
        self._chk(self.L.md_dev_debug_segments(self.h, slot, None, 0, C.byref(n), C.byref(nr)), "md_dev_debug_segments")
        arr = (md_seg * max(1, n.value))()
        self._chk(self.L.md_dev_debug_segments(self.h, slot, arr, n.value, C.byref(n), C.byref(nr)), "md_dev_debug_segments")
        return arr, n.value, nr.value

    def upload(self, slot: int, batch: md_read_batch):
        self._chk(self.L.md_dev_upload(self.h, slot, C.byref(batch)), "md_dev_upload")

    def launch(self, slot: int):
        self._chk(self.L.md_dev_launch(self.h, slot), "md_dev_launch")

    def download(self, slot: int) -> md_sites:
        s = md_sites()
        self._chk(self.L.md_dev_download(self.h, slot, C.byref(s)), "md_dev_download")
        return s

    def wait(self, slot: int) -> md_sites_dev:
        s = md_sites_dev()
        self._chk(self.L.md_dev_wait(self.h, slot, C.byref(s)), "md_dev_wait")
        return s

    def bind_output(self, slot: int, d_site, d_var, d_seg, cap_sites: int, cap_tiles: int):
        self._chk(self.L.md_dev_bind_output(self.h, slot, d_site, d_var, d_seg, cap_sites, cap_tiles), "md_dev_bind_output")

    def bench_rotate(self, slots, warmup: int, iters: int, per_launch: int = 1) -> md_bench_result:
        """kernel time with HIP events while rotating over several resident intervals (working set beyond the Infinity Cache),
        `per_launch` intervals per kernel launch"""
        r = md_bench_result()
        arr = (C.c_int * len(slots))(*slots)
        self._chk(self.L.md_dev_bench_rotate(self.h, arr, len(slots), per_launch, warmup, iters, C.byref(r)), "md_dev_bench_rotate")
        return r

    def launch_group(self, slots):
        arr = (C.c_int * len(slots))(*slots)
        self._chk(self.L.md_dev_launch_group(self.h, arr, len(slots)), "md_dev_launch_group")

    def bench(self, slot: int, warmup: int, iters: int) -> md_bench_result:
        r = md_bench_result()
        self._chk(self.L.md_dev_bench(self.h, slot, warmup, iters, C.byref(r)), "md_dev_bench")
        return r

    def mbias_submit(self, slot: int, batch: md_read_batch):
        """accumulate the batch's calls into the device histogram; the batch must stay alive until slot_sync(slot)"""
        self._chk(self.L.md_dev_mbias_submit(self.h, slot, C.byref(batch)), "md_dev_mbias_submit")

    def mbias_submit_raw(self, slot: int, raw: "md_raw_batch"):
        self._chk(self.L.md_dev_mbias_submit_raw(self.h, slot, C.byref(raw)), "md_dev_mbias_submit_raw")

    def slot_sync(self, slot: int):
        self._chk(self.L.md_dev_slot_sync(self.h, slot), "md_dev_slot_sync")

    def mbias_read(self):
        """-> numpy uint32 array [len, 4 strands, 2 reads, (meth, unmeth)]"""
        import numpy as np
        m = md_mbias()
        self._chk(self.L.md_dev_mbias_read(self.h, C.byref(m)), "md_dev_mbias_read")
        if m.len <= 0:
            return np.zeros((0, 4, 2, 2), dtype=np.uint32)
        return np.ctypeslib.as_array(m.count, shape=(m.len * 16,)).reshape(m.len, 4, 2, 2).copy()

    def perread(self, slot: int, batch: md_pr_batch):
        """per-read CpG counts of a perRead chunk -> [(nmeth, nunmeth)] (submit + download)"""
        self._chk(self.L.md_dev_perread_submit(self.h, slot, C.byref(batch)), "md_dev_perread_submit")
        out, n = C.POINTER(md_pr_count)(), C.c_int64()
        self._chk(self.L.md_dev_perread_download(self.h, slot, C.byref(out), C.byref(n)), "md_dev_perread_download")
        return [(out[i].nmeth, out[i].nunmeth) for i in range(n.value)]

    def perread_raw(self, slot: int, raw: "md_raw_batch"):
        """perRead from a chunk's raw records: -> (kept record indices, [(nmeth, nunmeth)]) as the device selected and walked them"""
        self._chk(self.L.md_dev_perread_submit_raw(self.h, slot, C.byref(raw)), "md_dev_perread_submit_raw")
        kept, out, n = C.POINTER(C.c_uint32)(), C.POINTER(md_pr_count)(), C.c_int64()
        self._chk(self.L.md_dev_perread_download_raw(self.h, slot, C.byref(kept), C.byref(out), C.byref(n)), "md_dev_perread_download_raw")
        return [kept[i] for i in range(n.value)], [(out[i].nmeth, out[i].nunmeth) for i in range(n.value)]

    def mbias_reset(self):
        self._chk(self.L.md_dev_mbias_reset(self.h), "md_dev_mbias_reset")

    def sync(self):
        self._chk(self.L.md_dev_sync(self.h), "md_dev_sync")

    def close(self):
        if self.h:
            self.L.md_dev_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """The host pipeline of one `extract` command line, a chunk at a time (mdk_plan_*)."""

    def __init__(self, args, command: str = "extract"):
        L = lib_extract()
        self.L = L
        self.command = command
        self.args = [command] + [str(a) for a in args]
        self._argv = _argv(self.args)
        self.p = C.c_void_p()
        opener = {"extract": L.mdk_plan_open, "mbias": L.mdk_plan_open_mbias, "perRead": L.mdk_plan_open_perread}[command]
        self.rc = opener(len(self.args), self._argv, C.byref(self.p))
        if self.rc or not self.p:
            raise MdkError(f"{opener.__name__} returned {self.rc}")

    def mbias_outputs(self):
        """(prefix or None, svg, txt, which) of an mbias plan"""
        pre, svg, txt, which = C.c_char_p(), C.c_int(), C.c_int(), C.c_int()
        if self.L.mdk_plan_mbias_outputs(self.p, C.byref(pre), C.byref(svg), C.byref(txt), C.byref(which)):
            raise MdkError("not an mbias plan")
        return (pre.value.decode() if pre.value else None), svg.value, txt.value, which.value

    def dev_cfg(self) -> md_dev_cfg:
        cfg = md_dev_cfg()
        self.L.mdk_plan_dev_cfg(self.p, C.byref(cfg))
        return cfg

    def set_shard(self, rank: int, world: int):
        if self.L.mdk_plan_set_shard(self.p, rank, world):
            raise MdkError("mdk_plan_set_shard: bad rank/world")

    def set_prep(self, mode: int):
        """0: chunks are prepared on the host (`batch`); 1: they come as raw records for the device (`raw`)"""
        if self.L.mdk_plan_set_prep(self.p, mode):
            raise MdkError("mdk_plan_set_prep: not possible for this plan (already started, or a perRead/mbias plan)")

    def prep_cfg(self) -> "md_prep_cfg":
        c = md_prep_cfg()
        self.L.mdk_plan_prep_cfg(self.p, C.byref(c))
        return c

    def attach_device(self, dev: "Device"):
        """BGZF inflate on the device too (include/mdk_extract.h mdk_plan_attach_device); detach_device before the device is closed"""
        rc = self.L.mdk_plan_attach_device(self.p, dev.h)
        if rc != 0:
            raise MdkError(f"mdk_plan_attach_device rc={rc}")
        self._attached = True

    def detach_device(self):
        if getattr(self, "_attached", False) and self.p:
            self.L.mdk_plan_detach_device(self.p)
            self._attached = False

    def host_prepare_from(self, chunk: "mdk_chunk", dev: "Device", slot: int):
        rc = self.L.mdk_plan_host_prepare_from(self.p, C.byref(chunk), dev.h, slot)
        if rc != 0:
            raise MdkError(f"mdk_plan_host_prepare_from rc={rc}")

    def host_prepare(self, chunk: "mdk_chunk"):
        rc = self.L.mdk_plan_host_prepare(self.p, C.byref(chunk))
        if rc:
            raise MdkError(f"mdk_plan_host_prepare failed ({rc})")

    def next_chunk(self):
        c = mdk_chunk()
        rc = self.L.mdk_plan_next_chunk(self.p, C.byref(c))
        if rc < 0:
            raise MdkError(f"mdk_plan_next_chunk failed ({rc})")
        return c if rc == 1 else None

    def ensure_reference(self, dev: Device, tid: int):
        rc = self.L.mdk_plan_ensure_reference(self.p, dev.h, tid)
        if rc:
            raise MdkError(f"mdk_plan_ensure_reference failed ({rc}): {lib_hip().md_dev_last_error().decode()}")

    def emit(self, chunk: mdk_chunk, sites: md_sites):
        rc = self.L.mdk_plan_emit(self.p, C.byref(chunk), C.byref(sites))
        if rc:
            raise MdkError(f"mdk_plan_emit failed ({rc})")

    def emit_perread(self, chunk: mdk_chunk, counts):
        """counts: [(nmeth, nunmeth)] per read of the chunk, or None for a chunk whose contig the FASTA lacks"""
        arr, n = None, 0
        if counts is not None:
            n = len(counts)
            arr = (md_pr_count * max(n, 1))(*[md_pr_count(m, u) for m, u in counts])
        rc = self.L.mdk_plan_emit_perread(self.p, C.byref(chunk), arr, n)
        if rc:
            raise MdkError(f"mdk_plan_emit_perread failed ({rc})")

    def finish(self):
        self.L.mdk_plan_finish(self.p)

    def regions(self, tid: int):
        """None without -l, else the [(start, end, strand)] runs the contig's sites are restricted to"""
        ptr, n = C.POINTER(md_region)(), C.c_int64()
        if self.L.mdk_plan_regions(self.p, tid, C.byref(ptr), C.byref(n)):
            raise MdkError("mdk_plan_regions failed")
        return None if n.value < 0 else [(ptr[i].start, ptr[i].end, ptr[i].strand) for i in range(n.value)]

    def target_name(self, tid: int) -> str:
        return self.L.mdk_plan_target_name(self.p, tid).decode()

    def close(self):
        if self.p:
            self.detach_device()
            self.L.mdk_plan_close(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mbias_report(hist, opref, svg: bool, txt: bool, which: int) -> int:
    """makeSVGs/makeTXT of the reference over a [len,4,2,2] uint32 histogram (writes <opref>_<strand>.svg, prints to
    the process's stdout/stderr)"""
    import numpy as np
    a = np.ascontiguousarray(hist, dtype=np.uint32).reshape(-1)
    m = md_mbias(len(a) // 16, a.ctypes.data_as(C.POINTER(C.c_uint32)))
    return lib_extract().mdk_mbias_report(C.byref(m), os.fsencode(str(opref)) if opref is not None else None, int(svg), int(txt), which)


def sites_to_rows(s: md_sites):
    """md_sites -> list of (pos, type, isG, nmeth, nunmeth, noff, nvar) tuples (for tests)."""
    rows = []
    for i in range(s.n_sites):
        r = s.site[i]
        rows.append((r.pos, (r.meta >> 1) & 3, r.meta & 1, r.nmeth, r.nunmeth, s.var[i].noff if s.var else 0, s.var[i].nvar if s.var else 0))
    return rows


def run_cli(args, cwd=None, env=None, command="extract", ranks=None, timeout=None):
    """Run the `MethylDackel extract` (or `mbias`) command of this build; returns CompletedProcess.  `ranks=N` runs the command
    as N processes, one per GPU (csrc/host/mdk_ranks.c), and returns rank 0's."""
    if not CLI.exists():
        raise MdkError(f"{CLI} is missing (run `make`)")
    e = dict(os.environ)
    # a GPU exception in the command: the runtime prints what it was (address, reason) and aborts, instead of piping a GPU core dump to the
    # host's core_pattern helper, which no container holds (that attempt is all round 4's one faulting run left on stderr)
    e.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")
    if env:
        e.update(env)
    if ranks:
        return run_ranks(args, ranks, cwd=cwd, env=e, command=command, timeout=timeout or 900)
    if "MDK_WORLD" not in e:
        e["MDK_NO_RANKS"] = "1"        # a caller that is itself a torchrun rank (bench.py) runs the command alone, not as its rank
    return subprocess.run([str(CLI), command] + [str(a) for a in args], cwd=cwd, env=e, capture_output=True, text=True, timeout=timeout)


def run_ranks(args, n, cwd=None, env=None, command="extract", devices=None, timeout=900):
    """`MethylDackel extract` as N processes: rank k takes chunks k, k+N, ... of the one schedule every rank derives from the
    same inputs, and rank 0 collects and writes (csrc/host/mdk_ranks.c; the launcher a site would use is `torchrun
    --no-python` or tools/extract_ranks.sh).  `devices`: GPU ordinal per rank (default: rank k on GPU k modulo the visible
    GPUs).  Returns rank 0's CompletedProcess with `.rank_returncodes` and `.rank_stderr` of all ranks."""
    import socket
    if not CLI.exists():
        raise MdkError(f"{CLI} is missing (run `make`)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update({"MDK_RANK": str(r), "MDK_WORLD": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        if devices is not None:
            e["MDK_DEVICE"] = str(devices[r])
        procs.append(subprocess.Popen([str(CLI), command] + [str(a) for a in args], cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        import threading
        res = [None] * n

        def drain(i):
            res[i] = procs[i].communicate(timeout=timeout)
        ths = [threading.Thread(target=drain, args=(i,)) for i in range(n)]
        [t.start() for t in ths]; [t.join() for t in ths]
        outs = res
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    out0 = outs[0] or ("", "")
    cp = subprocess.CompletedProcess(procs[0].args, procs[0].returncode, out0[0], out0[1])
    cp.rank_returncodes = [p_.returncode for p_ in procs]
    cp.rank_stderr = [(o or ("", ""))[1] for o in outs]
    return cp

/*
 * mdk_extract.h -- C ABI of the host side of the MI355X `extract` path (libmdk_extract.so).
 *
 * Drop-in symbol (what the reference's main.c:18,49-50 binds):
 *     int extract_main(int argc, char *argv[]);            reference: extract.c:706
 * Same argv contract (argv[0] == "extract"), option surface, validation order, messages, output
 * file naming/format and return codes as the reference; the per-chunk work (extractCalls,
 * extract.c:247-560) runs on the GPU through include/mdk_hip.h.  There is no CPU fallback: without
 * a usable device extract_main prints the HIP error and returns -20.
 *
 * The staged API below exposes the same pipeline one reference chunk at a time (used by the
 * parity tests, bench.py and the multi-GPU sharded driver).
 */
#ifndef MDK_EXTRACT_H
#define MDK_EXTRACT_H
#include <stdint.h>
#include "mdk_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MDK_RC_NODEVICE (-20)   /* GPU path unavailable */
#define MDK_RC_DEVICE   (-21)   /* a device operation failed mid-run */
#define MDK_RC_OUTPUT   (-22)   /* writing an output file failed (message on stderr) */

int extract_main(int argc, char *argv[]);

typedef struct mdk_plan mdk_plan;
#define MDK_CHUNK_NOREF   1   /* contig missing from the FASTA -> the reference skips the chunk; nothing is emitted */
#define MDK_CHUNK_BED     4   /* -l: no BED region touches the chunk -> the reference passes it over (extract.c:352-369) */
#define MDK_CHUNK_FOREIGN 2   /* interval sharding: another rank owns this chunk; its sites arrive through the gather */

/* One chunk of the reference's schedule (extract.c:325-350 + adjustBounds) with its admitted reads packed
 * for the device.  `batch` arrays are owned by the plan and stay valid until the second-next
 * mdk_plan_next_chunk call (two rotating pinned buffers). */
typedef struct {
    uint32_t index;            /* localBin: output order */
    int32_t  tid;
    int64_t  beg, end;         /* [localPos, localEnd) */
    int32_t  skipped;          /* MDK_CHUNK_* flags; non-zero: nothing was packed for this chunk */
    md_read_batch batch;
    uint64_t n_records_seen;   /* BAM records examined for this chunk (before admission) */
    md_pr_batch pr;            /* perRead plans: the reads that start in the chunk (batch is unused then) */
    const void *host;          /* perRead plans: the plan's own record of those reads (names), for mdk_plan_emit_perread */
    int32_t  prep;             /* 1: device preparation -- `raw` describes the chunk's records for md_dev_submit_raw and `batch` is empty
                                  (until mdk_plan_host_prepare) */
    md_raw_batch raw;
} mdk_chunk;

/* Parse an `extract` command line (argv[0] = "extract"), open inputs.  rc follows extract_main:
 * *out is NULL with rc==0 when the command line only asked for help/version. */
int  mdk_plan_open(int argc, char *argv[], mdk_plan **out);
void mdk_plan_close(mdk_plan *p);
/* device configuration implied by the options */
void mdk_plan_dev_cfg(const mdk_plan *p, md_dev_cfg *cfg);
/* upload the contig a chunk needs (no-op if already resident on that device handle) */
int  mdk_plan_ensure_reference(mdk_plan *p, md_dev *dev, int32_t tid);
/* 1: chunk produced; 0: schedule finished; <0: error */
int  mdk_plan_next_chunk(mdk_plan *p, mdk_chunk *c);
/* the same without waiting: 2 (nothing handed out) when the next chunk of the schedule is not ready yet */
int  mdk_plan_try_next_chunk(mdk_plan *p, mdk_chunk *c);
/* Where a chunk's per-record work happens.  mode 0 (what mdk_plan_open gives): on the host -- chunks carry `batch`.
 * mode 1 (what the commands use): on the device -- chunks carry `raw`, and md_dev_set_prep must be given
 * mdk_plan_prep_cfg's configuration (plus md_dev_set_mappability per contig, done by mdk_plan_ensure_reference).
 * Must be called before the first mdk_plan_next_chunk.  mdk_plan_host_prepare fills `batch` of a mode-1 chunk after all
 * (for a chunk the device answered with MDK_ERR_PREP_HOST); valid until the second-next mdk_plan_next_chunk. */
int  mdk_plan_set_prep(mdk_plan *p, int mode);
/* how many of the chunks handed out last stay valid (default 2: a chunk's arrays live until the second-next
 * mdk_plan_next_chunk); a caller that keeps more chunks in flight (several GPUs) raises it first.  2..40. */
int  mdk_plan_set_hold(mdk_plan *p, int n);
void mdk_plan_prep_cfg(const mdk_plan *p, md_prep_cfg *cfg);
int  mdk_plan_host_prepare(mdk_plan *p, mdk_chunk *c);
/* the same when the chunk was uploaded to `slot` of `dev` and some of its records were inflated on the device (they are read back) */
int  mdk_plan_host_prepare_from(mdk_plan *p, mdk_chunk *c, md_dev *dev, int slot);
/* the host memory behind a mode-1 chunk's record ranges is no longer needed (its upload has completed: md_dev_upload_wait): the inflate
 * teams may reuse it now instead of when the chunk is recycled.  The chunk stays valid otherwise; mdk_plan_host_prepare_from then reads
 * all of its records back from the device. */
int  mdk_plan_release_records(mdk_plan *p, const mdk_chunk *c);
/* BGZF inflate of the plan's BAM on this device as well (include/mdk_hip.h md_piece_*): from now on pieces of the
 * file are inflated by whoever is free, a host inflate team or the device, and chunks may name device-resident ranges.  Only for
 * plans in device-preparation mode of the `extract` command; mdk_plan_detach_device must precede md_dev_close. */
int  mdk_plan_attach_device(mdk_plan *p, md_dev *dev);
void mdk_plan_detach_device(mdk_plan *p);
/* host post-pass for one chunk (variant filter, --mergeContext, formats; extract.c:443-510), appended to
 * the plan's output files.  Chunks must be emitted in index order. */
int  mdk_plan_emit(mdk_plan *p, const mdk_chunk *c, const md_sites *sites);
/* print the variant-position line (extract.c:1489), close outputs */
int  mdk_plan_finish(mdk_plan *p);
/* interval sharding over `world` processes (one per GPU): this process admits and packs only the chunks whose index
 * is congruent to `rank`; every process still walks the whole schedule, so chunk indices agree everywhere. */
int  mdk_plan_set_shard(mdk_plan *p, int rank, int world);
int  mdk_plan_n_targets(const mdk_plan *p);
/* -l: the disjoint runs (include/mdk_hip.h: md_region) the sites of a contig are restricted to, as handed to
 * md_dev_set_regions by mdk_plan_ensure_reference; *n = -1 when no BED file was given.  Replaces the cursor walk over
 * config->bed in extractCalls (extract.c:352-369,402-405; bed.c:22-53). */
int  mdk_plan_regions(const mdk_plan *p, int32_t tid, const md_region **runs, int64_t *n);
const char *mdk_plan_target_name(const mdk_plan *p, int32_t tid);
int64_t mdk_plan_target_len(const mdk_plan *p, int32_t tid);

/* ---- `mbias` (MBias.c; main.c:17,51-52 dispatches to mbias_main) ----
 * mbias_main: drop-in for the reference symbol (argv[0] = "mbias"); same options, messages, return codes, SVG / --txt
 * output and "Suggested inclusion options" line; -20/-21 when the GPU is unavailable/fails.
 * mdk_plan_open_mbias: the same plan object for an `mbias` command line: chunks come out of mdk_plan_next_chunk with
 * batches built WITHOUT mate pairing (mbias installs no overlap handler, MBias.c:158-161), to be fed to
 * md_dev_mbias_submit.  mdk_plan_mbias_outputs tells what the command line asked to be written (prefix is NULL with
 * --noSVG; which = keepCpG + 2 keepCHG + 4 keepCHH as passed to makeSVGs, MBias.c:556).
 * mdk_mbias_report: makeSVGs + makeTXT (svg.c:300-454) over the histogram the device returns. */
int  mbias_main(int argc, char *argv[]);
/* for a caller that leaves with _exit after one of the entry points has returned (the `MethylDackel` command): joins the thread that brings the
 * HIP runtime up and the device library's helper threads, so that none of them is inside the runtime when the process goes */
void mdk_cli_quiesce(void);
int  mdk_plan_open_mbias(int argc, char *argv[], mdk_plan **out);
int  mdk_plan_mbias_outputs(const mdk_plan *p, const char **opref, int *svg, int *txt, int *which);
int  mdk_mbias_report(const md_mbias *hist, const char *opref, int svg, int txt, int which);

/* ---- `perRead` (perRead.c; main.c:20,55-56 dispatches to perRead_main) ----
 * perRead_main: drop-in for the reference symbol (argv[0] = "perRead").  mdk_plan_open_perread: chunks without
 * adjustBounds, `pr` filled with the alignments that start in the chunk and pass -F/-R/-q (perRead.c:178-183), for
 * md_dev_perread_submit; a chunk of a contig the FASTA lacks comes out with MDK_CHUNK_NOREF set AND its reads listed
 * (the reference prints them with zero calls).  mdk_plan_emit_perread writes addRead's lines (perRead.c:16-36) for a
 * chunk, in chunk order; counts may be NULL for a MDK_CHUNK_NOREF chunk. */
int  perRead_main(int argc, char *argv[]);
int  mdk_plan_open_perread(int argc, char *argv[], mdk_plan **out);
int  mdk_plan_emit_perread(mdk_plan *p, const mdk_chunk *c, const md_pr_count *counts, int64_t n);
/* the same for a chunk handed out as raw records (mdk_plan_set_prep(p, 1)): kept[i] = index into c->raw.rec_off of the i-th kept
 * read, counts[i] its calls (md_dev_perread_download_raw); names and positions are read from the records themselves */
int  mdk_plan_emit_perread_raw(mdk_plan *p, const mdk_chunk *c, const uint32_t *kept, const md_pr_count *counts, int64_t n);

/* Bind the calling thread, and every thread it creates from now on, to the CPUs next to the index-th AMD GPU (sysfs
 * local_cpulist, GPUs in PCI order), if that leaves at least half of the CPUs the process may use.  Returns the number of CPUs
 * bound to, 0 if nothing was changed (no such GPU, one NUMA node, MDK_NO_BIND set).  The `MethylDackel` command calls it for
 * its single-GPU commands; a library caller decides for itself.  No counterpart in the reference. */
int  mdk_bind_to_device_node(int index);

/* ---- `mergeContext` (mergeContext.c; main.c:19,53-54): text-to-text host tool, no device work ---- */
int  mergeContext_main(int argc, char *argv[]);

#ifdef __cplusplus
}
#endif
#endif

/* mdk_plan.h -- INTERNAL to csrc/host: the plan object behind include/mdk_extract.h and what its translation units share.
 *   mdk_plan.c       options, inputs (BAM, FASTA, mappability, BED), plan lifetime
 *   mdk_pipeline.c   per-record work (admission, strand, pairing, CIGAR expansion) and the reader/worker chunk pipeline
 *   mdk_emit.c       text post-pass and the ordered multi-threaded emitter
 *   mdk_extract.c    extract_main (the drop-in entry point) and the process-level helpers of the commands
 *   mdk_cmd_mbias.c, mdk_cmd_perread.c, mdk_mbias.c, mdk_mergecontext.c   the other commands
 * Nothing here is part of the C ABI. */
#ifndef MDK_PLAN_H
#define MDK_PLAN_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <ctype.h>
#include <errno.h>
#include <getopt.h>
#include <inttypes.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "mdk_extract.h"
#include "mdk_io.h"

static inline double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

#define MDK_VERSION "0.6.1"

#define MDK_LOCAL __attribute__((visibility("hidden")))

/* ------------------------------------------------------------------------------------------------ */
/* options (the reference's Config, MethylDackel.h:90-126; defaults extract.c:715-753)               */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    int ctx_on[3];                 /* CpG, CHG, CHH */
    int min_mapq, min_phred, keep_dupes, min_depth, keep_discordant, keep_singleton;
    int ignore_flags, require_flags, merge, methylkit, min_opp_depth, ignore_nh;
    double max_variant_frac;
    int fraction, counts, logit, cytosine_report;
    float min_conv_eff, map_cutoff; int min_mappable;
    int rel_bounds[16], abs_bounds[16];
    int n_threads; unsigned long chunk_size;
    char *region, *opref, *bbm_name, *bw_name, *bed_name, *out_bbm_name;     /* opref and out_bbm_name are owned */
    int output_bb, no_bam, keep_strand;
    int perread;                                        /* `perRead` command: reads that START in the chunk, flags/MAPQ only (perRead.c) */
    int mbias, svg, txt; char *mb_opref;                 /* `mbias` command: no pairing, no outputs of its own (MBias.c) */
    const char *fasta_name, *bam_name;
} opts_t;

/* text buffer */
typedef struct { char *s; size_t l, m; } sbuf;
static inline void sb_put(sbuf *b, const char *s, size_t n) {
    if(b->l + n + 1 > b->m) { b->m = (b->l + n + 1) * 2; b->s = realloc(b->s, b->m); if(!b->s) { fprintf(stderr, "[mdk] out of memory while formatting output\n"); abort(); } }
    memcpy(b->s + b->l, s, n); b->l += n; b->s[b->l] = 0;
}
/* room for n more bytes (+ the terminating 0): the caller writes at b->s + b->l and advances b->l itself */
static inline char *sb_room(sbuf *b, size_t n) {
    if(b->l + n + 1 > b->m) { b->m = (b->l + n + 1) * 2; b->s = realloc(b->s, b->m); if(!b->s) { fprintf(stderr, "[mdk] out of memory while formatting output\n"); abort(); } }
    return b->s + b->l;
}

/* one admitted read, host-side only (the device gets segments) */
typedef struct { int32_t pos, rend, mate; uint32_t off4, lq, cig_off, qn_off, qn_hash; uint16_t ncig, bamflag; uint8_t strand, second; } rinfo;
/* growable batch arrays; blob and seg are pinned (they are what gets uploaded) */
typedef struct {
    rinfo *ri; size_t n, cap_ri;
    uint32_t *cig; size_t cig_len, cig_cap;
    char *qn; size_t qn_len, qn_cap;
    uint8_t *blob; size_t blob_len, cap_blob;
    md_seg *seg; size_t n_seg, cap_seg;
    md_pr_read *pr; size_t cap_pr;                      /* perRead: one device record per kept read */
    uint64_t algo_bytes;
} batchbuf;

/* qname table entry for the pairing pass */
typedef struct { uint32_t h, qoff; int32_t pending, used; int32_t live[2]; int32_t nlive; int32_t more; } qent;     /* 32 bytes; more = head of a side list (index + 1) for the rare third and later records of a name */

/* what formatting one chunk needs and produces: the text per output file, the pending --mergeContext sites (they never
 * cross a chunk: extract.c:496-507) and the count of positions dropped as likely variants */
typedef struct {
    sbuf ob[3];
    int32_t lastcpg_tid, lastcpg_pos, lastchg_tid, lastchg_pos; uint32_t lastcpg_m, lastcpg_u, lastchg_m, lastchg_u;
    uint64_t n_variant;
} emit_ctx;

#define MDK_HOLD_MAX 56        /* chunks a consumer may hold at once (mdk_plan_set_hold): 6 groups of 8 in flight + 2, with room */
struct mdk_plan {
    opts_t o;
    mdk_bam *bam; mdk_bai *bai; int need_seek; mdk_fasta fa; int *fa_of_tid;
    /* schedule cursor (main.c:10-13 globals) */
    uint32_t g_tid, g_pos, g_end, bin;
    int dev_prep;                      /* chunks are handed out as raw records for the device to prepare (extract; see mdk_plan_set_prep) */
    int shard_rank, shard_world;       /* interval sharding: this process packs chunk k iff k % world == rank */
    int (*claim)(void *ctx, uint32_t index); void *claim_ctx;      /* ... or, when set (mdk_ranks.c, MDK_CLAIM=1), iff claim(ctx, k): the ranks claim chunks as they get to them, as the reference's workers do (extract.c:325-350) */
    uint64_t n_variant_positions;
    /* stream state */
    int32_t last_tid, last_pos; int at_eof;
    uint8_t *carry; size_t carry_len, carry_cap; int32_t carry_tid;
    uint8_t *carry2; size_t carry2_len, carry2_cap;
    struct { mdk_slab *slab; int mi; size_t mark; } *dm, *dm2; int n_dm, cap_dm, n_dm2, cap_dm2;      /* members of device-inflated slabs the next chunk looks at again (mdk_pipeline.c) */
    /* chunk pipeline: reader thread -> worker threads -> ordered delivery (see the pipeline section) */
    int (*slot_state)(const struct mdk_plan *, int);      /* (diagnostics) state of pipeline slot k */
    struct pslot *slot; int n_slot, n_workers; pthread_t reader_th, *worker_th; int started, quit, pipe_rc, reader_done;
    pthread_mutex_t mu; pthread_cond_t cv_free, cv_raw, cv_done;
    uint32_t next_out; int held[MDK_HOLD_MAX], n_hold;      /* the chunks handed out last (newest first); the oldest is recycled by the next hand-out */
    /* mappability */
    int map_on; uint32_t map_n; char **map_names; uint32_t *map_len; uint8_t **map_bits; int *map_of_tid;
    /* -l: per contig, the disjoint runs a position must fall in (and the strand a read must have there) */
    int bed_on; md_region **bed_run; int64_t *bed_nrun;
    FILE *pr_out; int pr_out_owned;                     /* perRead: -o file or stdout */
    /* outputs */
    FILE *out[3]; sbuf ob[3]; emit_ctx ec;
    uint32_t next_emit;
    double t_collect, t_pair, t_segs, t_emit, t_rfill, t_rwait, t_widle, t_wbusy;      /* MDK_HOST_PROFILE=1: seconds per host stage */
    /* device references already uploaded: (dev handle, tid) pairs */
    md_dev **ref_dev; int32_t *ref_tid; int n_ref, cap_ref;
};


/* extract_main's emitter: chunks are formatted by a few threads and written in chunk order (mdk_emit.c) */
enum { EJ_FREE = 0, EJ_READY, EJ_BUSY };
typedef struct { int state; mdk_chunk c; md_sites s; md_site *site; md_site_var *var; int64_t cap; emit_ctx e; const md_site *src_site; const md_site_var *src_var; int need_copy; } ejob;      /* need_copy: the sites still lie in the caller's buffer (emitter_push_lazy): the thread that takes the job copies them first */
typedef struct {
    mdk_plan *p; ejob *job; int n_job, n_th; pthread_t *th; pthread_mutex_t mu; pthread_cond_t cv_job, cv_free, cv_turn, cv_copied; int n_uncopied;
    uint32_t next_write; int quit; double t_format;
    int pw, fd[3], failed; int64_t woff[3];      /* pw: the outputs are regular files -> a chunk reserves its byte range in turn and is written with pwrite outside the lock */
} emitter;

/* opening the device on its own thread while the host pipeline is already running (mdk_extract.c) */
typedef struct { int device; md_dev_cfg cfg; md_dev *dev; int rc; char err[512]; } devopen_t;

MDK_LOCAL void plan_free(mdk_plan *p);
MDK_LOCAL int plan_open_ex(int argc, char *argv[], mdk_plan **out, void (*after_options)(mdk_plan *, void *), void *ctx);
MDK_LOCAL int plan_attach_inputs(mdk_plan *p, char *argv[], int first_positional);
MDK_LOCAL void parse_bounds(const char *arg, int *dst);
MDK_LOCAL void bb_free(batchbuf *b);
MDK_LOCAL int bed_touches(const mdk_plan *p, int32_t tid, int64_t beg, int64_t end);
MDK_LOCAL int map_window_passes(const mdk_plan *p, int c, int64_t start, int l);
MDK_LOCAL int ctx_code(const char *seq, int64_t len, int64_t i);
MDK_LOCAL int pipeline_start(mdk_plan *p);
MDK_LOCAL void pipeline_stop(mdk_plan *p);
MDK_LOCAL int emitter_start(emitter *E, mdk_plan *p, int n_th);
MDK_LOCAL int emitter_push(emitter *E, const mdk_chunk *c, const md_sites *s);
MDK_LOCAL int emitter_push_lazy(emitter *E, const mdk_chunk *c, const md_sites *s);      /* a large site array is copied by the emitter thread that takes the job: ... */
MDK_LOCAL void emitter_wait_copied(emitter *E);                                              /* ... and the caller waits here before it recycles the buffers it handed over */
MDK_LOCAL void emitter_stop(emitter *E);
/* the device index as the user gave it (the command narrows the runtime's view to that device, which then is number 0: csrc/host/main.c) */
static inline int user_device(int d) { const char *u = getenv("MDK_DEVICE_USER"); return u ? atoi(u) : d; }
MDK_LOCAL int fast_exit_wanted(void);
MDK_LOCAL void leave_fast(int ret);
MDK_LOCAL void leave_fast_plan(struct mdk_plan *p, int ret);      /* the same with the plan whose reaper thread and device inflate teams must have left the HIP runtime first */
MDK_LOCAL void hip_warm_up(void);
MDK_LOCAL void *devopen_main(void *arg);
MDK_LOCAL int ranks_from_env(int *rank, int *world);
MDK_LOCAL int extract_ranks(int argc, char *argv[], int rank, int world);
#endif

/* mdk_io.h -- host I/O substrate (plain C, zlib only): BGZF/BAM streaming reader with parallel inflate and a
 * FASTA loader.  htslib is not available in this image, and on the MI355X path the host's only jobs are
 * "inflate, find record boundaries, admit, pack" -- the reference gets these from htslib
 * (sam_itr_next at common.c:413, faidx_fetch_seq at extract.c:381). */
#ifndef MDK_IO_H
#define MDK_IO_H
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#pragma GCC visibility push(hidden)      /* internal to libmdk_extract.so: not part of the C ABI */

/* set-up allocations (options, name tables, bitmaps, region lists): running out of memory there is fatal, as in the reference */
static inline void *mdk_fatal_oom(void) { fprintf(stderr, "[mdk] out of memory\n"); abort(); return NULL; }
static inline void *xmalloc(size_t n) { void *q = malloc(n ? n : 1); return q ? q : mdk_fatal_oom(); }
static inline void *xcalloc(size_t n, size_t m) { void *q = calloc(n ? n : 1, m ? m : 1); return q ? q : mdk_fatal_oom(); }
static inline void *xrealloc(void *o, size_t n) { void *q = realloc(o, n ? n : 1); return q ? q : mdk_fatal_oom(); }
static inline char *xstrdup(const char *s) { char *q = strdup(s); return q ? q : (char *)mdk_fatal_oom(); }


/* Inflated data travels in reference-counted SLABS so that the chunk workers can parse records in place (no copy):
 * an inflater thread (which fans the BGZF members of a slab out to a pool of threads) runs ahead of the scanner;
 * the scanner hands [slab, begin, end) ranges to the chunks; a slab returns to the pool when its last user drops it.
 * A record cut by a slab boundary is completed in the headroom in front of the next slab's data. */
#define MDK_SLAB_HEADROOM (16u << 20)
/* Record summaries.  htslib never lets a BAM record straddle two BGZF members (bam_write1 flushes the block first when the
 * record would not fit), so in the files this path normally sees every member starts on a record boundary.  The thread
 * that inflates a member therefore also walks it from its first byte while the data is still in its cache, noting each
 * record's place, contig, start and end; a member whose walk ends exactly at its last byte is `ok`.  When the scanner
 * reaches the first byte of an ok member at a record boundary, the records of that member are exactly the ones noted
 * (induction over the members), and it reads them from the table instead of chasing block_size words through memory
 * that other cores have just written.  Any other file (records split across members) simply never matches and is
 * scanned the slow way. */
typedef struct { uint32_t off, len; int32_t tid, pos, endp; } mdk_rsum;          /* off: of the block_size word in the slab; len: block_size */
typedef struct { uint32_t off, len, n_sum; uint32_t sum0; int ok;     /* off, len: its bytes in the slab; records sum[sum0 .. sum0+n_sum) */
                 int32_t tid0, pos0, tidN, posN, min_endp, max_endp; int sorted; } mdk_member;    /* digest of an ok member: first/last record, extent of the ends, coordinate order inside */
/* A slab inflated ON THE DEVICE (piece != NULL) has no host bytes and no record summaries: buf and sum are NULL,
 * d_buf / d_rec_off are device pointers (inflated bytes; offset of every record in d_buf), mem[] holds the members' digests with
 * off/len in d_buf and sum0/n_sum indexing d_rec_off.  The reader applies the chunk schedule to such a slab member by member. */
struct md_piece;
typedef struct mdk_slab { uint8_t *buf; size_t cap, beg, end; int refs;
                          mdk_rsum *sum; size_t n_sum, cap_sum; mdk_member *mem; int n_mem, cap_mem;
                          uint32_t *off32; size_t cap_off32;        /* off32[i] = sum[i].off: the records' places as one array, which the device takes as it is (md_raw_range.h_rec_off) */
                          struct md_piece *piece; const uint8_t *d_buf; const uint32_t *d_rec_off; uint64_t d_bytes; uint32_t d_records;
                          size_t file_beg, file_end; int spec, spec_fail; } mdk_slab;      /* spec: the piece was cut without the file's lock (mdk_io.c claim_range): file_beg/file_end = its members' bytes in the file, checked against the piece before it when the scanner takes it; spec_fail: its team could not frame or inflate it */

#define MDK_GPU_TEAMS_MAX 16       /* device inflate teams: a host thread, a pinned staging block and pieces in flight each */
typedef struct mdk_bam {
    FILE *f;
    int nthreads;
    /* inflater teams + slab queue/pool.  A team takes the next piece of the file under io_mu (read + member headers: serial
     * and cheap), inflates it with its share of the threads while another team is already reading the following piece,
     * and queues the slab when its turn comes (seq order) */
    pthread_t inf_th[8]; int n_teams, team_threads, inf_started;
    pthread_mutex_t mu, io_mu, life_mu; pthread_cond_t cv_q, cv_pool;      /* life_mu: starting / stopping the teams (a seek by the reader thread vs. mdk_bam_attach_device by the caller's) */
    uint64_t next_seq; int io_status;               /* (io_mu) pieces handed out; 0 reading, 1 end of file, <0 error */
    /* (mu) finished slabs wait in ready[seq % MDK_READY] until the scanner has taken every earlier one: a team that finishes early
     * goes on with its next piece instead of waiting for its turn (device pieces are several times larger than host pieces) */
#define MDK_READY 64
    mdk_slab *ready[MDK_READY]; int n_ready; uint64_t pop_seq; int io_end;      /* io_end: copy of io_status once non-zero */
    int inf_done, quit, host_leaves, header_done; size_t gpu_piece_bytes; int gpu_piece_members;      /* gpu_piece_members: a device piece ends after this many members (a whole number of the device's rounds of wavefronts), 0 = by bytes only */
    /* teams that inflate on the device (mdk_bam_attach_device): each stages a piece of the file in registered memory and hands it
     * to the device library (md_piece_*); they share the piece counter with the host teams */
    struct md_dev *dev; pthread_t gpu_th[MDK_GPU_TEAMS_MAX]; int n_gpu_teams, gpu_started; uint8_t *gpu_stage[MDK_GPU_TEAMS_MAX]; size_t gpu_stage_cap[MDK_GPU_TEAMS_MAX];
    mdk_slab **dpool; int n_dpool, cap_dpool, n_dalloc, max_dalloc; uint64_t n_dev_pieces, n_host_pieces, n_materialized;
    mdk_slab **pool; int n_pool, cap_pool, n_alloc, max_alloc;
    /* Once the last piece of the file has been handed to a team (io_end), a slab that comes back is not kept for a next piece that will never
     * come: a reaper thread gives its staging memory back (hipHostUnregister + unmap) while the last chunks are still on their way through
     * the device -- memory that is still registered when the process ends is taken down by the kernel page by page on one core, 0.3 s for
     * the ~40 slabs of a 128 Mb run (gpurun_out/r05h/e2e.json: 0.39 s inside the process, 0.67 s for its caller). */
    mdk_slab **reap; int n_reap, cap_reap, n_pool_wait, reap_started, reap_quit, reap_busy; pthread_t reap_th; pthread_cond_t cv_reap, cv_reaped; int n_reaped; double t_reap;
    double tt_next[2], tt_slab[2], tt_copy[2], tt_dev[2], tt_deliver[2], tt_host[2]; int tt_pieces[2];      /* MDK_HOST_PROFILE: where the inflate teams' time went, summed over the teams ([0] host teams, [1] device teams) */
    uint8_t *cbuf; size_t ccap, clen; int file_eof;
    const uint8_t *map; size_t map_len, map_pos;   /* the file mapped read-only: the inflate threads read the compressed bytes where the page cache has them (no copy) */
    /* the mapping's page-table entries are made AHEAD of the framing (mdk_io.c populate_ahead): the walk over the members' headers runs under io_mu, one team at
     * a time, and a first touch of a page there costs the whole feed a microsecond per member */
    double t_frame;                                /* (io_mu) seconds inside next_piece */
    /* pieces cut WITHOUT walking the members under the lock (mdk_io.c claim_range): spec_pos = where the next piece nominally starts, spec_start = the exact
     * member boundary the reading began at (the first piece's start needs no search); spec_verified (scanner) = the end of the last piece taken; spec_off:
     * a piece did not begin where the one before it ended (or could not be framed): the rest of the file is framed under the lock as before */
    size_t spec_pos, spec_start, spec_verified; int spec_on, spec_off, spec_active; size_t spec_avg_member; uint64_t n_spec_redo;
    int seeked;                                    /* (mu) mdk_bam_seek has been called: the end of the file is not the end of the reading */
    size_t pop_next;                               /* (atomic) up to where the mapping's entries have been made or are being made */
    /* scanner position */
    mdk_slab *cur; size_t off;
    int mem_i; size_t sum_i, sum_end;        /* next member to look at; records of the current ok member still to hand out */
    uint64_t n_fast, n_slow;                 /* records taken from the tables / found by walking */
    int32_t n_targets; char **target_name; uint32_t *target_len;
    char *text; uint32_t l_text;
    uint64_t n_records;
    double t_inflate;                        /* seconds the scanner waited for inflated data */
    char err[256];
} mdk_bam;

/* a decoded view of one BAM record (pointers into the reader's buffer; valid until the next mdk_bam_peek
 * after mdk_bam_advance) */
typedef struct {
    int32_t tid, pos, l_qseq, mtid, mpos;
    uint16_t flag, n_cigar; uint8_t mapq, l_qname;
    const char *qname; const uint8_t *cigar, *seq, *qual, *aux; int32_t aux_len;
    const uint8_t *raw; uint32_t raw_len;    /* whole record after block_size */
} mdk_rec;

mdk_bam *mdk_bam_open(const char *fn, int nthreads);
/* slab holding the record last returned by mdk_bam_peek, and the byte offset of that record's block_size word in it */
mdk_slab *mdk_bam_cur_slab(mdk_bam *b, size_t *off);
void mdk_slab_ref(mdk_bam *b, mdk_slab *s);
void mdk_slab_unref(mdk_bam *b, mdk_slab *s);
void mdk_bam_reap_wait(mdk_bam *b);
void mdk_bam_teams_leave(mdk_bam *b);       /* stop and join the device inflate teams (a command on its way out) */      /* until the slabs given up after the end of the file have been unregistered and unmapped */
void mdk_bam_close(mdk_bam *b);
/* from now on, pieces of the file are also inflated on this device (n_teams threads, each with its own device piece); safe while
 * the host teams are running.  mdk_bam_detach_device stops those threads and frees the device pieces: before md_dev_close. */
int mdk_bam_attach_device(mdk_bam *b, struct md_dev *dev, int n_teams);
void mdk_bam_detach_device(mdk_bam *b);
/* stream position for slabs inflated on the device: 1 = the scanner stands at member *mi of device slab *s; 0 = it stands in a host
 * slab (use mdk_bam_peek_sum), or at the end of the data; <0 error.  mdk_bam_dev_advance consumes that member. */
int mdk_bam_at_device(mdk_bam *b, mdk_slab **s, int *mi);
void mdk_bam_dev_advance(mdk_bam *b);
/* make every blocked or future read return end-of-data (used to stop a reader thread) */
void mdk_bam_abort(mdk_bam *b);
/* 1 = record available, 0 = end of file, <0 = error (b->err) */
int mdk_bam_peek(mdk_bam *b, mdk_rec *r);
void mdk_bam_advance(mdk_bam *b, const mdk_rec *r);
/* the same walk for callers that only need a record's place and extent: *raw = first byte after block_size */
int mdk_bam_peek_sum(mdk_bam *b, mdk_rsum *r, const uint8_t **raw);
void mdk_bam_advance_sum(mdk_bam *b, const mdk_rsum *r);
/* when the scanner stands inside an ok member (after a mdk_bam_peek_sum that returned 1): the summaries it has not handed out
 * yet, v[0..n), and the member's digest; 0 otherwise.  mdk_bam_advance_run consumes the first k of them at once. */
int mdk_bam_member_run(mdk_bam *b, const mdk_rsum **v, size_t *n, const mdk_member **m);
void mdk_bam_advance_run(mdk_bam *b, size_t k);
/* decode a raw record (bytes after block_size) */
int mdk_rec_parse(const uint8_t *raw, uint32_t len, mdk_rec *r);

typedef struct { int n; char **name; char **seq; int64_t *len; char *pool; } mdk_fasta;
int mdk_fasta_load(const char *fn, mdk_fasta *fa);
void mdk_fasta_free(mdk_fasta *fa);
int mdk_fasta_find(const mdk_fasta *fa, const char *name);


/* BAI index (SAM spec 5.2): only the 16 kb linear index is kept -- enough to find where to start reading for a
 * reference window; starting too early is harmless because the caller skips records that end before its window. */
typedef struct { int32_t n_ref; int32_t *n_intv; uint64_t **ioff; uint64_t *first; } mdk_bai;
mdk_bai *mdk_bai_load(const char *bam_fn);      /* <bam>.bai or <bam without .bam>.bai; NULL if absent/unreadable */
void mdk_bai_free(mdk_bai *x);
/* virtual offset to start reading at for records overlapping [beg, ...) of tid; 0 = that reference has no records at or after beg */
uint64_t mdk_bai_start(const mdk_bai *x, int32_t tid, int64_t beg);
/* reposition the record stream at a BGZF virtual offset (compressed offset << 16 | offset in the inflated member) */
int mdk_bam_seek(mdk_bam *b, uint64_t voffset);

/* bigWig (mappability track for -M) */
typedef struct { FILE *f; uint64_t chrom_tree, index; uint32_t uncompress; uint32_t n, cap; char **name; uint32_t *len, *id; } mdk_bigwig;
mdk_bigwig *mdk_bigwig_open(const char *fn);
void mdk_bigwig_close(mdk_bigwig *bw);
float *mdk_bigwig_values(mdk_bigwig *bw, uint32_t k);

#pragma GCC visibility pop
#endif

/* mdk_io.c -- BGZF/BAM streaming reader (parallel block inflate) and FASTA loader.  See mdk_io.h. */
#define _GNU_SOURCE
#include "mdk_io.h"
#include "mdk_hip.h"       /* md_host_alloc / md_host_free: staging memory for the slabs */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <time.h>
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
static double io_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

#define CCHUNK (8u << 20)      /* compressed bytes a host team inflates into one slab */
#define GCHUNK (96u << 20)     /* compressed bytes of a piece inflated on the device: ~5,100 members, one wavefront each, on a device that holds ~6,100 (64 MB: ~3,400, whose k_inflate ran mostly alone and left the device under-filled; 512 Mb: 1.22 -> 1.10 s, profiles/r05z_e2e_piece_size.log) */

static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

typedef struct { const uint8_t *in; uint32_t in_len; uint8_t *out; uint32_t out_len, crc; int th; size_t sum0; uint32_t n_sum; int ok;      /* crc: the CRC32 of the member's trailer */
                 int32_t tid0, pos0, tidN, posN, min_endp, max_endp; int sorted; } blk_t;
typedef struct { mdk_rsum *v; size_t n, cap; } sumbuf;
typedef struct { blk_t *blk; int n; int next; int failed; int n_th; int next_th; const uint8_t *base; sumbuf sb[64]; pthread_mutex_t mu; } inflate_job;

/* walk one inflated member from its first byte (see mdk_io.h); the notes go to the calling thread's buffer */
static void note_records(blk_t *b, sumbuf *sb, const uint8_t *base) {
    const uint8_t *d = b->out; uint32_t L = b->out_len, o = 0;
    b->sum0 = sb->n; b->n_sum = 0; b->ok = 0; b->sorted = 1; b->min_endp = 0x7fffffff; b->max_endp = (int32_t)0x80000000; b->tid0 = b->tidN = -1; b->pos0 = b->posN = -1;
    while(o + 4 <= L) {
        uint32_t bs = le32(d + o), lq, nc, k, rl = 0; const uint8_t *r = d + o + 4, *c; mdk_rsum *q;      /* (rl: unsigned -- a member that starts inside a record is walked as if it started one, and what stands where a CIGAR would adds up to anything) */
        if(bs < 32 || (uint64_t)o + 4 + bs > L) return;
        lq = r[8]; nc = le16(r + 12);
        if(32u + lq + 4u * nc > bs) return;
        c = r + 32 + lq;
        for(k = 0; k < nc; k++) { uint32_t v = le32(c + 4 * k), op = v & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += v >> 4; }
        if(sb->n == sb->cap) { sb->cap = sb->cap ? sb->cap * 2 : 4096; sb->v = realloc(sb->v, sizeof(mdk_rsum) * sb->cap); if(!sb->v) { sb->cap = sb->n = 0; return; } }
        q = &sb->v[sb->n++];
        q->off = (uint32_t)(d + o - base); q->len = bs; q->tid = (int32_t)le32(r); q->pos = (int32_t)le32(r + 4); q->endp = (int32_t)((uint32_t)q->pos + ((int32_t)rl > 0 ? rl : 1u));
        if(b->n_sum == 0) { b->tid0 = q->tid; b->pos0 = q->pos; }
        else if(q->tid < 0 || q->tid < b->tidN || (q->tid == b->tidN && q->pos < b->posN)) b->sorted = 0;
        if(q->tid < 0) b->sorted = 0;                       /* unplaced records: leave them to the record-by-record path */
        b->tidN = q->tid; b->posN = q->pos;
        if(q->endp < b->min_endp) b->min_endp = q->endp;
        if(q->endp > b->max_endp) b->max_endp = q->endp;
        b->n_sum++;
        o += 4 + bs;
    }
    b->ok = o == L;
}

/* libdeflate inflates a BGZF member two to three times faster than zlib (it is what htslib itself uses when it is built with
 * it).  The image ships the runtime library without its header, so it is bound by name; without it zlib does the work. */
typedef struct { void *(*alloc)(void); int (*run)(void *, const void *, size_t, void *, size_t, size_t *); void (*release)(void *); uint32_t (*crc)(uint32_t, const void *, size_t); } ldeflate_t;
static const ldeflate_t *ldeflate(void) {
    static ldeflate_t L; static int state = 0; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    if(state == 0) {
        void *so = getenv("MDK_ZLIB_INFLATE") ? NULL : dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        state = -1;
        if(so) {
            L.alloc = (void *(*)(void))dlsym(so, "libdeflate_alloc_decompressor");
            L.run = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(so, "libdeflate_deflate_decompress");
            L.release = (void (*)(void *))dlsym(so, "libdeflate_free_decompressor");
            L.crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(so, "libdeflate_crc32");
            if(L.alloc && L.run && L.release && L.crc) state = 1;
        }
    }
    pthread_mutex_unlock(&mu);
    return state == 1 ? &L : NULL;
}

/* what htslib's bgzf_read_block checks behind sam_itr_next (common.c:413): the CRC32 of the inflated bytes against the member's trailer
 * (ISIZE is checked by the inflate itself: it must produce exactly that many bytes).  MDK_NO_CRC=1 skips it (timing comparisons). */
static int crc_wanted(void) { static int w = -1; if(w < 0) w = getenv("MDK_NO_CRC") ? 0 : 1; return w; }
static void *inflate_worker(void *arg) {
    inflate_job *job = arg; z_stream zs; int inited = 0, me; const ldeflate_t *LD = ldeflate(); void *ld = LD ? LD->alloc() : NULL;
    pthread_mutex_lock(&job->mu); me = job->next_th++; pthread_mutex_unlock(&job->mu);
    for(;;) {
        int i;
        pthread_mutex_lock(&job->mu); i = job->next; job->next += 8; pthread_mutex_unlock(&job->mu);
        if(i >= job->n) break;
        for(int k = i; k < i + 8 && k < job->n; k++) {
            blk_t *b = &job->blk[k];
            b->th = me; b->sum0 = job->sb[me].n; b->n_sum = 0; b->ok = 1;      /* an empty member (the EOF marker) holds no record and ends where it starts */
            if(!b->out_len) continue;
            if(ld) {
                size_t got = 0;
                if(LD->run(ld, b->in, b->in_len, b->out, b->out_len, &got) != 0 || got != b->out_len) { job->failed = 1; b->ok = 0; continue; }
                if(crc_wanted() && LD->crc(0, b->out, b->out_len) != b->crc) { job->failed = 2; b->ok = 0; continue; }
                note_records(b, &job->sb[me], job->base);
                continue;
            }
            if(!inited) { memset(&zs, 0, sizeof(zs)); if(inflateInit2(&zs, -15) != Z_OK) { job->failed = 1; return NULL; } inited = 1; }
            else inflateReset(&zs);
            zs.next_in = (Bytef *)b->in; zs.avail_in = b->in_len; zs.next_out = b->out; zs.avail_out = b->out_len;
            if(inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { job->failed = 1; b->ok = 0; continue; }
            if(crc_wanted() && (uint32_t)crc32(0L, b->out, b->out_len) != b->crc) { job->failed = 2; b->ok = 0; continue; }
            note_records(b, &job->sb[me], job->base);
        }
    }
    if(inited) inflateEnd(&zs);
    if(ld) LD->release(ld);
    return NULL;
}

/* ---- the mapped file's pages, ahead of the teams ----
 * next_piece walks the BGZF headers of a piece under io_mu.  In a fresh mapping every page it touches is a page fault (the pages are in the page
 * cache, but this process has no entry for them yet): ~0.8 us per member, 0.4 s per 9 GB of BAM, and serial for all teams -- the feed ran at the
 * speed of that walk (gpurun_out r06r: 0.40 s of a 0.5 s streaming phase inside the lock, teams queueing for it 1.3 s in sum).  So every team,
 * once it holds a piece and has let go of the lock, makes the entries of ONE block further ahead (MADV_POPULATE_READ, Linux 5.14; by touching a
 * byte per page where the kernel does not know it): together the teams keep a frontier POP_AHEAD in front of the read position, at ~2 ms of
 * each piece's team.  (Threads of their own that ran ahead through the whole file held the address space's lock against the runtime's start-up:
 * the device was usable 0.15 s later, gpurun_out r06n.)
 * OFF since the end of round 6 (MDK_POPULATE=1 turns it on): measured against each other by interleaved runs on three boxes (profiles/r06_e2e_ab.txt), the 512 Mb
 * run takes 1.03-1.07 s without it and 1.12-1.16 s with it -- a dozen teams making entries at once fight over the same address-space lock the walk's single faults take. */
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
#define POP_BLOCK ((size_t)64 << 20)
#define POP_AHEAD ((size_t)1 << 30)
static void populate_ahead(mdk_bam *b) {
    static int by_touch = 0, off = -1;
    if(off < 0) off = getenv("MDK_POPULATE") && !getenv("MDK_NO_POPULATE") ? 0 : 1;      /* off unless asked for: see above */
    if(off || !b->map) return;
    { const size_t pos = __atomic_load_n(&b->map_pos, __ATOMIC_RELAXED); size_t at = __atomic_load_n(&b->pop_next, __ATOMIC_RELAXED), len;
      if(at < pos) { __atomic_compare_exchange_n(&b->pop_next, &at, pos, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED); at = __atomic_load_n(&b->pop_next, __ATOMIC_RELAXED); }      /* (a seek, or readers that overtook the frontier) */
      if(at >= b->map_len || at > pos + POP_AHEAD) return;
      at = __atomic_fetch_add(&b->pop_next, POP_BLOCK, __ATOMIC_RELAXED);
      if(at >= b->map_len) return;
      at &= ~(size_t)4095; len = at + POP_BLOCK <= b->map_len ? POP_BLOCK : b->map_len - at;
      if(!__atomic_load_n(&by_touch, __ATOMIC_RELAXED) && madvise((void *)(b->map + at), len, MADV_POPULATE_READ) != 0) __atomic_store_n(&by_touch, 1, __ATOMIC_RELAXED);
      if(__atomic_load_n(&by_touch, __ATOMIC_RELAXED)) { volatile uint8_t sink = 0; size_t o; for(o = 0; o < len; o += 4096) sink ^= b->map[at + o]; (void)sink; } }
}

/* ---- slabs ---- */
#define SEQ_FORCE UINT64_MAX
static mdk_slab *slab_get_ex(mdk_bam *b, size_t need_cap, uint64_t seq) {          /* a free slab with room for need_cap bytes, for piece `seq`; SEQ_FORCE: never wait for one to come back */
    mdk_slab *s = NULL;
    pthread_mutex_lock(&b->mu);
    while(!b->quit) {
        if(b->n_pool) { s = b->pool[--b->n_pool]; break; }
        /* max_alloc only bounds how far the inflaters run AHEAD.  The piece the scanner takes next (seq == pop_seq) never waits: the
         * scanner may be collecting a chunk that spans more slabs than the cap (huge --chunkSize, deep coverage), and the chunks the
         * consumer still holds keep their slabs until it has been given further chunks -- slabs that only come back if the scanner
         * goes on.  (Slabs delivered out of order by other teams say nothing about that: n_ready > 0 is not "the scanner has work".) */
        if(seq == SEQ_FORCE || b->n_alloc < b->max_alloc || seq == b->pop_seq) { b->n_alloc++; s = calloc(1, sizeof(*s)); break; }
        b->n_pool_wait++; pthread_cond_wait(&b->cv_pool, &b->mu); b->n_pool_wait--;
    }
    pthread_mutex_unlock(&b->mu);
    if(!s) return NULL;
    /* slabs are what the device preparation uploads from: staging memory (pinned for inputs large enough to repay the pinning,
     * md_host_set_pinned in mdk_plan.c), so that the H2D copies of a chunk are asynchronous and run at the link's speed */
    if(s->cap < need_cap) { md_host_free(s->buf); s->cap = need_cap + (need_cap >> 3); s->buf = md_host_alloc(s->cap); if(!s->buf) { free(s); return NULL; } }
    s->refs = 1; s->beg = s->end = MDK_SLAB_HEADROOM; s->n_mem = 0; s->n_sum = 0; s->spec = s->spec_fail = 0;
    return s;
}
/* ---- the reaper (see mdk_io.h) ---- */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define MDK_POOL_KEEP 2
static void slab_destroy(mdk_slab *s);
static void reap_push(mdk_bam *b, mdk_slab *s) {        /* (mu held) */
    if(b->n_reap == b->cap_reap) { b->cap_reap = b->cap_reap ? b->cap_reap * 2 : 32; b->reap = xrealloc(b->reap, sizeof(mdk_slab *) * b->cap_reap); }
    b->reap[b->n_reap++] = s; b->n_alloc--;
    pthread_cond_signal(&b->cv_reap);
}
/* Giving registered memory back while the last chunks are on their way pays for files whose run holds gigabytes of it for seconds; for a small file it
 * is a loss: a 0.5 GB BAM (32 Mb at 30x) is through in 0.26 s, and the command then left in 0.02 s without the reaper and in 0.24 s with it -- nine
 * interleaved runs each, 0.257 against 0.523 s for the caller (profiles/r06_e2e_ab.txt r06pw) --, while at 2.2 GB the two are level (0.73 against 0.69 s) and
 * above that the reaper is ahead.  Below MDK_REAP_MIN_MB (default 1536) of BAM there is none; MDK_NO_REAP=1: never. */
static int reap_wanted(const mdk_bam *b) {
    static long min_mb = -1;
    if(getenv("MDK_NO_REAP")) return 0;
    if(min_mb < 0) min_mb = getenv("MDK_REAP_MIN_MB") ? atol(getenv("MDK_REAP_MIN_MB")) : 1536;
    return !b->map || (long)(b->map_len >> 20) >= min_mb;
}
static void *reaper_main(void *arg) {
    mdk_bam *b = arg;
    pthread_mutex_lock(&b->mu);
    for(;;) {
        while(!b->n_reap && !b->reap_quit) pthread_cond_wait(&b->cv_reap, &b->mu);
        if(!b->n_reap) break;
        { mdk_slab *s = b->reap[--b->n_reap]; const double t0 = now_s(); b->reap_busy = 1; pthread_mutex_unlock(&b->mu); slab_destroy(s); pthread_mutex_lock(&b->mu); b->reap_busy = 0; b->n_reaped++; b->t_reap += now_s() - t0; }
        if(!b->n_reap) pthread_cond_broadcast(&b->cv_reaped);
    }
    pthread_mutex_unlock(&b->mu);
    return NULL;
}
/* the end of the file has been reached: what the pool holds beyond a couple of slabs goes too */
static void reap_pool(mdk_bam *b) {                       /* (mu held) */
    if(!b->reap_started || b->n_pool_wait || b->seeked) return;      /* (a reader that seeks -- a rank of a sharded run, a region list -- reaches the end of the file before every seek: its slabs stay until it is closed) */
    while(b->n_pool > MDK_POOL_KEEP) reap_push(b, b->pool[--b->n_pool]);
}
/* wait until the reaper has nothing left to give back (a command about to leave: what it has not unregistered the kernel will, slowly) */
/* a command about to leave with _exit: the device teams are told to stop and joined -- none of them may be inside the HIP runtime (md_piece_*, md_host_free of
 * its staging block) when the process goes */
void mdk_bam_teams_leave(mdk_bam *b) {
    int i;
    if(!b || !b->gpu_started) return;
    pthread_mutex_lock(&b->life_mu);
    pthread_mutex_lock(&b->mu); b->quit = 1; pthread_cond_broadcast(&b->cv_pool); pthread_cond_broadcast(&b->cv_q); pthread_mutex_unlock(&b->mu);
    if(b->gpu_started) { for(i = 0; i < b->n_gpu_teams; i++) pthread_join(b->gpu_th[i], NULL); b->gpu_started = 0; }
    pthread_mutex_unlock(&b->life_mu);
}
void mdk_bam_reap_wait(mdk_bam *b) {
    if(!b) return;
    pthread_mutex_lock(&b->mu);
    if(b->reap_started) reap_pool(b);
    { const double t0 = now_s();
      while(b->reap_started && (b->n_reap || b->reap_busy)) pthread_cond_wait(&b->cv_reaped, &b->mu);
      if(getenv("MDK_HOST_PROFILE") && !b->reap_started) fprintf(stderr, "[mdk host] reaper: none (a BAM below MDK_REAP_MIN_MB, or MDK_NO_REAP)\n");
      else if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk host] reaper: %d slabs given back in %.3fs of its own thread's time, %d left in the pool, %d still referenced; waited for it %.3fs at the end\n", b->n_reaped, b->t_reap, b->n_pool, b->n_alloc - b->n_pool, now_s() - t0);
      if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk host] framing the pieces (the walk over the BGZF headers, under the file's lock): %.3fs in all\n", b->t_frame);
      if(getenv("MDK_HOST_PROFILE")) { int k; for(k = 0; k < 2; k++) fprintf(stderr, "[mdk host] %s teams, summed over the teams that have left: %d pieces; waiting for the file's lock + framing %.3fs, inflating %.3fs (device teams: waiting for a device slab %.3fs, staging copy %.3fs, device %.3fs), handing over in file order %.3fs\n", k ? "device" : "host", b->tt_pieces[k], b->tt_next[k], b->tt_host[k], b->tt_slab[k], b->tt_copy[k], b->tt_dev[k], b->tt_deliver[k]); } }
    pthread_mutex_unlock(&b->mu);
}
static void reaper_stop(mdk_bam *b) {
    if(!b->reap_started) return;
    pthread_mutex_lock(&b->mu); b->reap_quit = 1; pthread_cond_broadcast(&b->cv_reap); pthread_mutex_unlock(&b->mu);
    pthread_join(b->reap_th, NULL);
    b->reap_started = 0; b->reap_quit = 0;
}
void mdk_slab_ref(mdk_bam *b, mdk_slab *s) { pthread_mutex_lock(&b->mu); s->refs++; pthread_mutex_unlock(&b->mu); }
void mdk_slab_unref(mdk_bam *b, mdk_slab *s) {
    pthread_mutex_lock(&b->mu);
    if(--s->refs == 0) {
        if(s->piece) {
            if(b->n_dpool == b->cap_dpool) { b->cap_dpool = b->cap_dpool ? b->cap_dpool * 2 : 8; b->dpool = xrealloc(b->dpool, sizeof(mdk_slab *) * b->cap_dpool); }
            b->dpool[b->n_dpool++] = s;
        } else if(b->io_end && !b->seeked && b->reap_started && !b->n_pool_wait && b->n_pool >= MDK_POOL_KEEP) {
            reap_push(b, s);                              /* nobody will ask for it again (a team still inflating its last piece finds MDK_POOL_KEEP in the pool, or makes one) */
        } else {
            if(b->n_pool == b->cap_pool) { b->cap_pool = b->cap_pool ? b->cap_pool * 2 : 16; b->pool = xrealloc(b->pool, sizeof(mdk_slab *) * b->cap_pool); }
            b->pool[b->n_pool++] = s;
        }
        pthread_cond_broadcast(&b->cv_pool);
    }
    pthread_mutex_unlock(&b->mu);
}

/* a piece of the file: the complete BGZF members found in the read window, with the compressed bytes they live in */
typedef struct { uint8_t *cbuf; blk_t *blk; int nb; size_t total; uint64_t seq; size_t map_beg, map_end; } piece;      /* map_beg/end: its bytes in the mapped file */

/* under io_mu: read CCHUNK more compressed bytes, list every complete member; the unfinished tail moves to a fresh window.
 * status: 0 a piece was produced, 1 end of file, <0 error */
static int next_piece(mdk_bam *b, piece *pc, size_t want, int max_members) {
    size_t n, off = 0, total = 0; blk_t *blk = NULL; int nb = 0, mb = 0;
    memset(pc, 0, sizeof(*pc));
    if(b->map) {        /* mapped file: the window is a view, nothing is read or copied here */
        b->cbuf = (uint8_t *)b->map + b->map_pos; b->clen = b->map_len - b->map_pos;
        if(b->clen > want + (1u << 17)) b->clen = want + (1u << 17); else b->file_eof = 1;
    } else
    if(!b->file_eof) {
        if(b->ccap < b->clen + want) { b->ccap = b->clen + want; b->cbuf = realloc(b->cbuf, b->ccap); if(!b->cbuf) return -1; }
        n = fread(b->cbuf + b->clen, 1, want, b->f);
        b->clen += n;
        if(n < want) b->file_eof = 1;
    }
    while(off + 18 <= b->clen && (max_members <= 0 || nb < max_members)) {
        const uint8_t *p = b->cbuf + off; uint16_t xlen; uint32_t bsize = 0, isize; size_t x; int have = 0;
        if(p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { snprintf(b->err, sizeof(b->err), "not a BGZF file (bad gzip member header)"); free(blk); return -2; }
        xlen = le16(p + 10);
        if(off + 12 + xlen > b->clen) break;
        for(x = 12; x + 4 <= 12u + xlen;) { uint16_t sl = le16(p + x + 2); if(p[x] == 'B' && p[x + 1] == 'C' && sl == 2) { bsize = le16(p + x + 4) + 1u; have = 1; } x += 4 + sl; }
        if(!have || bsize < 12u + xlen + 8u) { snprintf(b->err, sizeof(b->err), "BGZF member without a valid BC field"); free(blk); return -2; }
        if(off + bsize > b->clen) break;
        isize = le32(p + bsize - 4);
        if(nb == mb) { mb = mb ? mb * 2 : 1024; blk = realloc(blk, sizeof(blk_t) * mb); if(!blk) return -1; }
        blk[nb].in = p + 12 + xlen; blk[nb].in_len = bsize - 12 - xlen - 8; blk[nb].out = NULL; blk[nb].out_len = isize; blk[nb].crc = le32(p + bsize - 8); nb++;
        total += isize; off += bsize;
    }
    if(nb == 0) {
        free(blk);
        if(b->file_eof) { int trunc = b->clen != 0; if(b->map) { b->cbuf = NULL; b->clen = 0; } if(trunc) { snprintf(b->err, sizeof(b->err), "truncated BGZF member at end of file"); return -2; } return 1; }
        if(b->map) { b->cbuf = NULL; b->clen = 0; }
        snprintf(b->err, sizeof(b->err), "BGZF member larger than the read window"); return -2;
    }
    if(b->map) { pc->map_beg = b->map_pos; pc->map_end = b->map_pos + off; b->map_pos += off; b->cbuf = NULL; b->clen = 0; if(b->map_pos < b->map_len) b->file_eof = 0; }
    else {   /* the piece keeps this window; the tail that belongs to the next member starts a new one */
        size_t left = b->clen - off; uint8_t *nw = malloc(left + GCHUNK + 64);
        if(!nw) { free(blk); return -1; }
        memcpy(nw, b->cbuf + off, left);
        pc->cbuf = b->cbuf; b->cbuf = nw; b->ccap = left + GCHUNK + 64; b->clen = left;
    }
    pc->blk = blk; pc->nb = nb; pc->total = total;
    return 0;
}

/* ---- pieces cut without the lock ----
 * next_piece's walk over the members' headers is a chain (a member's BSIZE says where the next begins), run under io_mu by one team at a time, and in
 * a mapped file every member costs it a first touch of a page: 0.3-0.4 s for the 460,000 members of a 9 GB BAM, which the whole feed stood behind
 * (profiles/r06pm_*: the teams queued 1.1-2.1 s in sum for the lock, the thread that uploads waited 0.2 s of a 0.5 s streaming phase for chunks).
 * Here a team takes a NOMINAL range of the file under the lock (two additions) and finds the members in it by itself, next to the other teams: the first
 * member that begins at or after a nominal boundary is found by looking for a BGZF header whose BSIZE leads to another header, three times over; the
 * range's members are then walked exactly, header by header as next_piece does, from that start to the start found the same way at the range's nominal end.
 * What makes this exact and not merely likely: the scanner takes the pieces in file order and checks that every piece begins where the piece before it ENDED
 * (the first begins at a boundary known exactly).  By induction every piece it accepts was walked from a true member boundary.  A piece that does not
 * fit -- a false header inside compressed bytes passing the test, a damaged file -- and a piece its team could not frame or inflate are thrown away with
 * everything after them, and the rest of the file is framed under the lock as before, from the last verified boundary (spec_redo): errors are found and
 * reported by that path alone.  MDK_SPEC_FAULT=n (tests): piece n starts one member late.
 * NOT THE DEFAULT (MDK_SPEC_FRAMING=1 turns it on): measured by interleaved runs (profiles/r06_e2e_ab.txt) the 512 Mb run takes 1.11-1.23 s this way and 1.03-1.07 s
 * with the walk under the lock.  The streaming phase itself is shorter (the device's inflate lane is full: 8 pieces in flight, 4 ms apart), but with nothing
 * holding the teams back the runtime's start-up -- which maps and registers memory under the address-space lock the teams' threads keep taking -- needs 0.20 s
 * instead of 0.07 s for its first step alone, and the device is usable at 0.36 s instead of 0.24 (gpurun_out r06pp); holding all host teams but one back until
 * the device is attached (MDK_NO_SPEC_HOLD) recovers a third of that.  Nor does it pay when it only takes over once the device is attached (what
 * MDK_SPEC_FRAMING=1 does now; MDK_SPEC_AT_ONCE=1: from the file's first byte): 1.09 against 1.05 s at 512 Mb, 1.66 against 1.53 s at 1 Gb (r06pv) -- in the
 * steady state the device is the limit (a kernel runs 88 % of the streaming phase, profiles/r06pu_trace_summary.txt), and a dozen teams that never wait only
 * queue more work behind it.  Kept, with its tests (tests/test_feed_harness.py), for a feed that is not device-bound (several GPUs behind one reader). */
/* `v` is a view of the file: v[y] is the file's byte y for vis_beg <= y < vis_end (the mapped file itself, or a range read into a team's buffer) */
typedef struct { const uint8_t *v; size_t vis_beg, vis_end, file_len; } fview;
static int member_at(const fview *f, size_t y, size_t *next) {        /* is there a well-formed BGZF member header at y?  *next = where the member ends */
    const uint8_t *p = f->v + y; uint16_t xlen; size_t x; uint32_t bsize = 0; int have = 0;
    if(y < f->vis_beg || y + 18 > f->vis_end || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    xlen = le16(p + 10);
    if(y + 12 + xlen > f->vis_end) return 0;
    for(x = 12; x + 4 <= 12u + xlen;) { uint16_t sl = le16(p + x + 2); if(p[x] == 'B' && p[x + 1] == 'C' && sl == 2) { bsize = le16(p + x + 4) + 1u; have = 1; } x += 4 + sl; }
    if(!have || bsize < 12u + xlen + 8u || y + bsize > f->file_len) return 0;
    *next = y + bsize;
    return 1;
}
static size_t find_member_start(const fview *f, size_t x) {        /* the first y >= x where a chain of three members (or fewer, up to the file's end) begins; file_len: none up to the file's end; (size_t)-1: none within a member's reach */
    const size_t len = f->file_len, lim = x + 65536 + 64 < len ? x + 65536 + 64 : len;
    size_t y;
    for(y = x; y + 18 <= lim && y + 18 <= f->vis_end; y++) {
        size_t n1, n2, n3;
        if(f->v[y] != 0x1f || f->v[y + 1] != 0x8b) continue;
        if(!member_at(f, y, &n1)) continue;
        if(n1 != len && (!member_at(f, n1, &n2) || (n2 != len && !member_at(f, n2, &n3)))) continue;
        return y;
    }
    return lim == len ? len : (size_t)-1;
}
/* under io_mu: status 0 = a range was handed out, 1 = end of the file */
static int claim_range(mdk_bam *b, size_t want, size_t *nom_beg, size_t *nom_end, int *first) {
    if(b->spec_pos >= b->map_len) return 1;
    *nom_beg = b->spec_pos; *first = b->spec_pos == b->spec_start;
    *nom_end = b->map_len - b->spec_pos > want + (256u << 10) ? b->spec_pos + want : b->map_len;       /* (no sliver of a last range) */
    b->spec_pos = *nom_end;
    return 0;
}
#define SPEC_SLACK ((size_t)4 * 65536 + 256)            /* past a range's nominal end: the boundary found there (<= 64 KB on) and the two members its test follows */
/* outside any lock: the members of the range.  The range's bytes are READ into the team's buffer (pread: no page of the mapped file is touched -- a first
 * touch there is a page fault under the address space's lock, which scales badly over a dozen teams, and a copy into the staging block was due anyway);
 * buf == NULL: taken from the mapping.  0 = framed (pc filled; nb may be 0); -9 = it cannot be framed from here (the caller delivers a failed piece) */
static int frame_range(mdk_bam *b, piece *pc, size_t nom_beg, size_t nom_end, int first, uint64_t seq, uint8_t *buf, size_t cap) {
    const size_t len = b->map_len; fview f; size_t beg, end, off, total = 0;
    blk_t *blk = NULL; int nb = 0, mb = 0;
    static long fault = -2; if(fault == -2) fault = getenv("MDK_SPEC_FAULT") ? atol(getenv("MDK_SPEC_FAULT")) : -1;
    pc->cbuf = NULL; pc->blk = NULL; pc->nb = 0; pc->total = 0; pc->map_beg = pc->map_end = nom_beg;
    f.v = b->map; f.vis_beg = 0; f.vis_end = len; f.file_len = len;
    if(buf) {
        const size_t want = (nom_end + SPEC_SLACK < len ? nom_end + SPEC_SLACK : len) - nom_beg; size_t got = 0;
        if(want > cap) return -9;
        while(got < want) { const ssize_t r = pread(fileno(b->f), buf + got, want - got, (off_t)(nom_beg + got)); if(r <= 0) return -9; got += (size_t)r; }
        f.v = buf - nom_beg; f.vis_beg = nom_beg; f.vis_end = nom_beg + want;
    }
    beg = first ? nom_beg : find_member_start(&f, nom_beg); end = nom_end >= len ? len : find_member_start(&f, nom_end);
    if(beg == (size_t)-1 || end == (size_t)-1 || beg > end) return -9;
    pc->map_beg = pc->map_end = beg;
    if(fault >= 0 && (uint64_t)fault == seq && beg < end) { size_t n1; if(member_at(&f, beg, &n1)) { beg = n1; pc->map_beg = pc->map_end = beg; } }
    for(off = beg; off < end;) {
        const uint8_t *p = f.v + off; size_t nx; uint16_t xlen; uint32_t bsize;
        if(!member_at(&f, off, &nx) || nx > f.vis_end) { free(blk); return -9; }
        xlen = le16(p + 10); bsize = (uint32_t)(nx - off);
        if(le32(p + bsize - 4) > 65536u) { free(blk); return -9; }          /* (no BGZF member inflates to more: not a boundary, or a file for the other path to refuse) */
        if(nb == mb) { blk_t *nw; mb = mb ? mb * 2 : 1024; nw = realloc(blk, sizeof(blk_t) * (size_t)mb); if(!nw) { free(blk); return -9; } blk = nw; }
        blk[nb].in = p + 12 + xlen; blk[nb].in_len = bsize - 12 - xlen - 8; blk[nb].out = NULL; blk[nb].out_len = le32(p + bsize - 4); blk[nb].crc = le32(p + bsize - 8); nb++;
        total += le32(p + bsize - 4); off = nx;
    }
    if(off != end) { free(blk); return -9; }
    pc->blk = blk; pc->nb = nb; pc->total = total; pc->map_beg = beg; pc->map_end = end;
    return 0;
}

/* inflate the members of a piece into a slab (fanned out to nthreads) and gather the record tables */
static mdk_slab *inflate_piece(mdk_bam *b, piece *pc, int nthreads, int *status) {
    blk_t *blk = pc->blk; int nb = pc->nb; mdk_slab *s;
    *status = 0;
    s = slab_get_ex(b, MDK_SLAB_HEADROOM + pc->total + 64, pc->seq);
    if(!s) { *status = b->quit ? 1 : -1; return NULL; }
    { size_t o = s->beg; for(int i = 0; i < nb; i++) { blk[i].out = s->buf + o; o += blk[i].out_len; } s->end = o; }
    {
        inflate_job job; int nt = nthreads, i; pthread_t th[64];
        memset(&job, 0, sizeof(job));
        job.blk = blk; job.n = nb; job.next = 0; job.failed = 0; job.base = s->buf; pthread_mutex_init(&job.mu, NULL);
        if(nt > 64) nt = 64;
        if(nt > (nb + 7) / 8) nt = (nb + 7) / 8;
        if(nt < 1) nt = 1;
        if(nt == 1) inflate_worker(&job);
        else {      /* the workers pull members from one counter: whoever could be started shares the slab, this thread included */
            int made = 0;
            for(i = 0; i < nt - 1; i++) { if(pthread_create(&th[made], NULL, inflate_worker, &job)) break; made++; }
            inflate_worker(&job);
            for(i = 0; i < made; i++) pthread_join(th[i], NULL);
        }
        pthread_mutex_destroy(&job.mu);
        if(!job.failed) {      /* the notes of all members, in stream order, into the slab's table */
            size_t tot = 0, o = 0;
            for(i = 0; i < nb; i++) tot += blk[i].n_sum;
            if(s->cap_sum < tot + 1) { free(s->sum); s->cap_sum = tot + tot / 8 + 1024; s->sum = malloc(sizeof(mdk_rsum) * s->cap_sum); }
            if(s->cap_mem < nb) { free(s->mem); s->cap_mem = nb + 64; s->mem = malloc(sizeof(mdk_member) * (size_t)s->cap_mem); }
            if(s->cap_off32 < tot + 1) { free(s->off32); s->cap_off32 = tot + tot / 8 + 1024; s->off32 = malloc(sizeof(uint32_t) * s->cap_off32); }
            s->n_mem = 0; s->n_sum = 0;
            if(s->sum && s->mem && s->off32) {
                for(i = 0; i < nb; i++) {
                    mdk_member *m = &s->mem[i];
                    m->off = (uint32_t)(blk[i].out - s->buf); m->len = blk[i].out_len; m->n_sum = blk[i].n_sum; m->sum0 = (uint32_t)o; m->ok = blk[i].ok && (blk[i].n_sum == 0 || job.sb[blk[i].th].v != NULL);
                    m->tid0 = blk[i].tid0; m->pos0 = blk[i].pos0; m->tidN = blk[i].tidN; m->posN = blk[i].posN; m->min_endp = blk[i].min_endp; m->max_endp = blk[i].max_endp; m->sorted = blk[i].sorted;
                    if(m->ok && m->n_sum) { uint32_t k; memcpy(s->sum + o, job.sb[blk[i].th].v + blk[i].sum0, sizeof(mdk_rsum) * m->n_sum); for(k = 0; k < m->n_sum; k++) s->off32[o + k] = s->sum[o + k].off; }
                    if(m->ok) o += m->n_sum; else m->n_sum = 0;
                }
                s->n_mem = nb; s->n_sum = o;
            }
        }
        for(i = 0; i < 64; i++) free(job.sb[i].v);
        if(job.failed) { pthread_mutex_lock(&b->mu); snprintf(b->err, sizeof(b->err), job.failed == 2 ? "a BGZF member fails its CRC32 check (corrupt file)" : "BGZF inflate failed (corrupt file?)"); pthread_mutex_unlock(&b->mu); mdk_slab_unref(b, s); *status = -2; return NULL; }
    }
    return s;
}

/* ---- slabs inflated on the device ---- */
static mdk_slab *dslab_get(mdk_bam *b, uint64_t seq) {          /* a free device slab (its piece keeps its device buffers from use to use) */
    mdk_slab *s = NULL;
    pthread_mutex_lock(&b->mu);
    while(!b->quit) {
        if(b->n_dpool) { s = b->dpool[--b->n_dpool]; break; }
        if(b->n_dalloc < b->max_dalloc || seq == b->pop_seq) { b->n_dalloc++;      /* (as slab_get_ex) */ s = calloc(1, sizeof(*s)); break; }
        pthread_cond_wait(&b->cv_pool, &b->mu);
    }
    pthread_mutex_unlock(&b->mu);
    if(!s) return NULL;
    if(!s->piece && md_piece_create(b->dev, &s->piece)) { pthread_mutex_lock(&b->mu); b->n_dalloc--; snprintf(b->err, sizeof(b->err), "%s", md_dev_last_error()); pthread_mutex_unlock(&b->mu); free(s); return NULL; }
    s->refs = 1; s->beg = s->end = 0; s->n_mem = 0; s->n_sum = 0; s->spec = s->spec_fail = 0;
    return s;
}
/* one piece through the device: the compressed bytes are staged in registered memory, md_piece_submit/wait inflates them and frames
 * the records; what comes back to the host is one digest per member */
static mdk_slab *inflate_piece_device(mdk_bam *b, piece *pc, int team, int *status, double *tt) {
    blk_t *blk = pc->blk; const int nb = pc->nb; mdk_slab *s; md_inf_member *mt; md_piece_info info; int i; uint64_t o = 0;
    const uint8_t *c0 = blk[0].in; const size_t span = (size_t)((blk[nb - 1].in + blk[nb - 1].in_len) - c0);
    double t0 = now_s(), t1;
    *status = 0;
    s = dslab_get(b, pc->seq);
    t1 = now_s(); tt[0] += t1 - t0; t0 = t1;
    if(!s) { *status = b->quit ? 1 : -1; return NULL; }
    const int staged = b->gpu_stage[team] && c0 >= b->gpu_stage[team] && c0 + span <= b->gpu_stage[team] + b->gpu_stage_cap[team];      /* the range was read into the staging block (frame_range): uploaded from where it lies */
    if(!staged && b->gpu_stage_cap[team] < span + 64) { md_host_free(b->gpu_stage[team]); b->gpu_stage_cap[team] = span + (span >> 3) + (1u << 20); b->gpu_stage[team] = md_host_alloc(b->gpu_stage_cap[team]); if(!b->gpu_stage[team]) { b->gpu_stage_cap[team] = 0; mdk_slab_unref(b, s); *status = -1; return NULL; } }
    const uint8_t *const src = staged ? c0 : b->gpu_stage[team];
    if(!staged) memcpy(b->gpu_stage[team], c0, span);
    t1 = now_s(); tt[1] += t1 - t0; t0 = t1;
    mt = malloc(sizeof(*mt) * (size_t)nb);
    if(!mt) { mdk_slab_unref(b, s); *status = -1; return NULL; }
    for(i = 0; i < nb; i++) { mt[i].in_off = (uint64_t)(blk[i].in - c0); mt[i].in_len = blk[i].in_len; mt[i].out_len = blk[i].out_len; mt[i].out_off = o; mt[i].crc32 = blk[i].crc; mt[i].reserved = 0; o += blk[i].out_len; }
    { int rs = md_piece_submit(s->piece, src, span, mt, nb);
      if(rs == MDK_ERR_NOMEM) { free(mt); mdk_slab_unref(b, s); *status = -1; return NULL; }      /* no device memory for this piece: the host's inflate takes it */
      if(rs || md_piece_wait(s->piece, &info)) {
        pthread_mutex_lock(&b->mu); snprintf(b->err, sizeof(b->err), "%s", md_dev_last_error()); pthread_mutex_unlock(&b->mu);
        free(mt); mdk_slab_unref(b, s); *status = -2; return NULL;
      } }
    tt[2] += now_s() - t0;
    if(s->cap_mem < nb) { free(s->mem); s->cap_mem = nb + 64; s->mem = malloc(sizeof(mdk_member) * (size_t)s->cap_mem); }
    if(!s->mem) { s->cap_mem = 0; free(mt); mdk_slab_unref(b, s); *status = -1; return NULL; }
    for(i = 0; i < nb; i++) {
        mdk_member *m = &s->mem[i]; const md_inf_digest *g = &info.digest[i];
        m->off = (uint32_t)mt[i].out_off; m->len = mt[i].out_len; m->n_sum = g->n_rec; m->sum0 = g->first_rec; m->ok = g->ok;
        m->tid0 = g->tid0; m->pos0 = g->pos0; m->tidN = g->tidN; m->posN = g->posN; m->min_endp = g->min_endp; m->max_endp = g->max_endp; m->sorted = g->sorted;
    }
    s->n_mem = nb; s->d_buf = info.d_out; s->d_rec_off = info.d_rec_off; s->d_bytes = info.out_bytes; s->d_records = info.n_records; s->beg = 0; s->end = (size_t)info.out_bytes;
    free(mt);
    return s;
}
/* a device slab the scanner cannot take member by member (a record straddles two members, or the slab before it ended inside a
 * record): its bytes come back to the host and it is walked like a slab a host team inflated */
static mdk_slab *slab_materialize(mdk_bam *b, mdk_slab *d) {
    mdk_slab *s = slab_get_ex(b, MDK_SLAB_HEADROOM + (size_t)d->d_bytes + 64, SEQ_FORCE); int i;      /* the scanner itself is asking: it must not wait for slabs only it can release */
    if(!s) return NULL;
    s->end = s->beg + (size_t)d->d_bytes;
    if(md_piece_read(d->piece, 0, d->d_bytes, s->buf + s->beg)) { pthread_mutex_lock(&b->mu); snprintf(b->err, sizeof(b->err), "%s", md_dev_last_error()); pthread_mutex_unlock(&b->mu); mdk_slab_unref(b, s); return NULL; }
    if(s->cap_mem < d->n_mem) { free(s->mem); s->cap_mem = d->n_mem + 64; s->mem = malloc(sizeof(mdk_member) * (size_t)s->cap_mem); if(!s->mem) { s->cap_mem = 0; mdk_slab_unref(b, s); return NULL; } }
    for(i = 0; i < d->n_mem; i++) { s->mem[i] = d->mem[i]; s->mem[i].off += (uint32_t)s->beg; s->mem[i].ok = 0; s->mem[i].n_sum = 0; }      /* no summaries: record by record */
    s->n_mem = d->n_mem; s->n_sum = 0;
    b->n_materialized++;
    return s;
}

/* a finished slab (or the news that there will be no more) goes to the scanner; slabs are taken in piece order */
static int deliver(mdk_bam *b, mdk_slab *s, uint64_t seq) {
    pthread_mutex_lock(&b->mu);
    while(seq >= b->pop_seq + MDK_READY && !b->quit) pthread_cond_wait(&b->cv_pool, &b->mu);
    if(b->quit) { pthread_mutex_unlock(&b->mu); mdk_slab_unref(b, s); return -1; }
    b->ready[seq % MDK_READY] = s; b->n_ready++;
    if(s->piece) b->n_dev_pieces++; else b->n_host_pieces++;
    pthread_cond_broadcast(&b->cv_q);
    pthread_mutex_unlock(&b->mu);
    return 0;
}
typedef struct { mdk_bam *b; int gpu_team, idx; } team_arg;        /* gpu_team < 0: a host team (the idx-th) */
static void *inflater_main(void *arg) {
    team_arg *ta = arg; mdk_bam *b = ta->b; const int gt = ta->gpu_team;
    double t_next = 0, t_host = 0, t_deliver = 0, td[3] = {0, 0, 0}; int n_pieces = 0; uint8_t *hbuf = NULL; size_t hbuf_cap = 0;      /* hbuf: a host team's range of the file (frame_range) */
    /* While the runtime is starting, ONE host team reads: with the file's lock out of their way four teams took a piece every few milliseconds between them, and
     * the runtime -- whose start-up maps and registers memory under the same address-space lock their page faults and registrations take -- had the device usable
     * 0.11 s later (0.36 s instead of 0.25, gpurun_out r06pp).  The others join when the device is attached, or after 0.4 s (a caller that attaches none). */
    if(gt < 0 && ta->idx > 0 && b->map && b->spec_on && getenv("MDK_SPEC_AT_ONCE") && !getenv("MDK_NO_SPEC_HOLD")) {
        static double th0 = 0; if(th0 == 0) th0 = now_s();      /* (the process's first reading: later restarts -- seeks -- find the 0.4 s over) */
        for(;;) { int q; if(__atomic_load_n(&b->dev, __ATOMIC_ACQUIRE) || now_s() - th0 > 0.4) break; pthread_mutex_lock(&b->mu); q = b->quit; pthread_mutex_unlock(&b->mu); if(q) break; usleep(1000); }
    }
    for(;;) {
        piece pc; int st; mdk_slab *s = NULL; double t0 = now_s(), t1;
        int spec = 0, fr = 0; size_t nom_beg = 0, nom_end = 0; int first = 0;
        pthread_mutex_lock(&b->io_mu);
        if(b->io_status) { pthread_mutex_unlock(&b->io_mu); break; }              /* another team has seen the end (or an error) */
        spec = b->map && b->spec_on && !b->spec_off;
        /* ... but only once the device is attached: until then the pieces are cut by the walk under the lock, which holds the teams back while the runtime starts
         * (see above).  The switch: the first range begins where the walk has got to -- a boundary known exactly. */
        if(spec && !b->spec_active) { if(__atomic_load_n(&b->dev, __ATOMIC_ACQUIRE) || getenv("MDK_SPEC_AT_ONCE")) { b->spec_active = 1; b->spec_pos = b->spec_start = b->map_pos; } else spec = 0; }
        if(spec) {      /* a nominal range, framed below next to the other teams (claim_range / frame_range) */
            size_t want = gt >= 0 ? b->gpu_piece_bytes : b->host_leaves ? (256u << 10) : CCHUNK;
            if(gt >= 0 && b->gpu_piece_members > 0) { const size_t avg = __atomic_load_n(&b->spec_avg_member, __ATOMIC_RELAXED); want = (size_t)((double)(avg ? avg : 16384u) * b->gpu_piece_members * 0.985); }      /* (a whole number of the device's rounds of members, a little under) */
            memset(&pc, 0, sizeof(pc));
            st = claim_range(b, want, &nom_beg, &nom_end, &first);
        } else {
            const double tf = now_s();
            st = next_piece(b, &pc, gt >= 0 ? b->gpu_piece_bytes : b->host_leaves ? (256u << 10) : CCHUNK, gt >= 0 ? b->gpu_piece_members : 0);
            b->t_frame += now_s() - tf;
        }
        if(st == 0) pc.seq = b->next_seq++; else b->io_status = st;
        pthread_mutex_unlock(&b->io_mu);
        if(spec && st == 0) {      /* the range's bytes into this team's buffer: a device team's staging block (what its piece is uploaded from), a host team's own */
            uint8_t *rb = NULL; size_t rcap = 0; const size_t need_cap = nom_end - nom_beg + SPEC_SLACK + 64;
            if(getenv("MDK_SPEC_PREAD")) {
                if(gt >= 0) {
                    if(b->gpu_stage_cap[gt] < need_cap) { md_host_free(b->gpu_stage[gt]); b->gpu_stage_cap[gt] = need_cap + (need_cap >> 3) + (1u << 20); b->gpu_stage[gt] = md_host_alloc(b->gpu_stage_cap[gt]); if(!b->gpu_stage[gt]) b->gpu_stage_cap[gt] = 0; }
                    rb = b->gpu_stage[gt]; rcap = b->gpu_stage_cap[gt];
                } else {
                    if(hbuf_cap < need_cap) { free(hbuf); hbuf_cap = need_cap + (need_cap >> 3); hbuf = malloc(hbuf_cap); if(!hbuf) hbuf_cap = 0; }
                    rb = hbuf; rcap = hbuf_cap;
                }
            }
            fr = frame_range(b, &pc, nom_beg, nom_end, first, pc.seq, rb, rcap);
        }
        if(spec && st == 0) { if(!fr && pc.nb) __atomic_store_n(&b->spec_avg_member, (pc.map_end - pc.map_beg) / (size_t)pc.nb, __ATOMIC_RELAXED); }
        t1 = now_s(); t_next += t1 - t0; t0 = t1;
        if(st == 0 && !spec) { populate_ahead(b); if(gt >= 0) populate_ahead(b); }       /* (a device team's piece is larger than a block) */
        if(st == 0 && spec && (fr || pc.nb == 0)) {      /* nothing to inflate: an empty piece, or one that could not be framed (the scanner starts the other path when it gets there) */
            s = slab_get_ex(b, MDK_SLAB_HEADROOM + 64, SEQ_FORCE);
            if(!s) { pthread_mutex_lock(&b->mu); if(!b->quit && b->inf_done >= 0) b->inf_done = -1; pthread_cond_broadcast(&b->cv_q); pthread_mutex_unlock(&b->mu); free(pc.blk); break; }
            s->spec = 1; s->spec_fail = fr ? 1 : 0; s->file_beg = pc.map_beg; s->file_end = pc.map_end;
            free(pc.blk);
            if(deliver(b, s, pc.seq)) break;
            continue;
        }
        if(st == 0) {
            /* a device team's piece goes to the host's inflate after all when it would inflate to more than the device addresses in one piece
             * (4 GiB: a ratio above 64, low-complexity data) or when the device cannot take it for want of a resource (status -1) */
            if(gt >= 0 && pc.total < 0xfff00000ull) { s = inflate_piece_device(b, &pc, gt, &st, td); if(!s && st == -1) { pthread_mutex_lock(&b->mu); const int q = b->quit; pthread_mutex_unlock(&b->mu); if(!q) { st = 0; s = inflate_piece(b, &pc, b->team_threads, &st); } } }
            else s = inflate_piece(b, &pc, b->team_threads, &st);
            free(pc.cbuf); free(pc.blk);
            if(!spec && s && b->map && b->spec_on) { s->spec = 1; s->spec_fail = 0; s->file_beg = pc.map_beg; s->file_end = pc.map_end; }      /* (cut by the walk: begins where the piece before it ended by construction; the scanner's frontier moves with it) */
            if(spec) {
                if(s) { s->spec = 1; s->spec_fail = 0; s->file_beg = pc.map_beg; s->file_end = pc.map_end; }
                else if(st == -2) {      /* its bytes did not inflate, or failed their CRC: a damaged file, or a piece that does not begin at a member -- the scanner decides (spec_redo) */
                    s = slab_get_ex(b, MDK_SLAB_HEADROOM + 64, SEQ_FORCE);
                    if(s) { s->spec = 1; s->spec_fail = 1; s->file_beg = pc.map_beg; s->file_end = pc.map_end; st = 0; pthread_mutex_lock(&b->mu); b->err[0] = 0; pthread_mutex_unlock(&b->mu); }
                }
            }
            /* the piece's pages of the file mapping are done with (inflated, or copied to the device's staging block): unmapped here, piece by piece
             * and on many threads, they are not left for the kernel to walk on one core when the process ends (they stay in the page cache) */
            if(b->map && pc.map_end > pc.map_beg) { const size_t a = (pc.map_beg + 4095) & ~(size_t)4095, e = pc.map_end & ~(size_t)4095; if(e > a) (void)madvise((void *)(b->map + a), e - a, MADV_DONTNEED); }
        }
        if(s && gt < 0) { md_dev *dv = __atomic_load_n(&b->dev, __ATOMIC_ACQUIRE); if(dv) md_host_register(dv, s->buf); }          /* the device is up: the slab this team has just filled is made known to the runtime here, not by the thread that uploads from it */
        t1 = now_s(); t_host += t1 - t0; t0 = t1;      /* (a device team: slab wait + copy + device, told apart in td) */
        if(s) {
            n_pieces++;
            if(deliver(b, s, pc.seq)) break;
            t_deliver += now_s() - t0;
            /* test hook (MDK_DEVICE_INFLATE_ONLY=1, `extract` only): the host teams leave after the piece that holds the BAM header, so that
             * every other piece is inflated on the device however small the file is */
            if(gt < 0 && b->host_leaves) { int hd; pthread_mutex_lock(&b->mu); hd = b->header_done; pthread_mutex_unlock(&b->mu); if(hd) break; }
            continue;
        }
        /* the end of the file, or an error */
        pthread_mutex_lock(&b->mu);
        if(st < 0) { if(b->inf_done >= 0) b->inf_done = st; }
        else if(!b->io_end) { b->io_end = 1; reap_pool(b); }
        pthread_cond_broadcast(&b->cv_q);
        pthread_mutex_unlock(&b->mu);
        break;
    }
    { const int k = gt >= 0; pthread_mutex_lock(&b->mu); b->tt_next[k] += t_next; b->tt_host[k] += t_host; b->tt_deliver[k] += t_deliver; b->tt_slab[k] += td[0]; b->tt_copy[k] += td[1]; b->tt_dev[k] += td[2]; b->tt_pieces[k] += n_pieces; pthread_mutex_unlock(&b->mu); }
    if(gt >= 0 && reap_wanted(b)) { md_host_free(b->gpu_stage[gt]); b->gpu_stage[gt] = NULL; b->gpu_stage_cap[gt] = 0; }      /* its last piece has crossed the link (md_piece_wait): the staging block goes now, not at exit */
    free(hbuf);
    free(ta);
    return NULL;
}
static void inflaters_start(mdk_bam *b) {
    int i;
    pthread_mutex_lock(&b->mu); b->next_seq = b->pop_seq = 0; b->io_status = 0; b->io_end = 0; pthread_mutex_unlock(&b->mu);      /* (no team is running; a thread giving a slab back looks at io_end: mdk_slab_unref) */
    b->spec_on = b->map && getenv("MDK_SPEC_FRAMING") && !getenv("MDK_SERIAL_FRAMING"); b->spec_active = 0; b->spec_pos = b->spec_start = b->spec_verified = b->map_pos;      /* (map_pos: the file's start, or the member a seek went to -- a boundary known exactly) */
    for(i = 0; i < b->n_teams; i++) { team_arg *ta = malloc(sizeof(*ta)); if(!ta) break; ta->b = b; ta->gpu_team = -1; ta->idx = i; if(pthread_create(&b->inf_th[i], NULL, inflater_main, ta)) { free(ta); break; } }
    if(i == 0) { b->io_status = -1; b->inf_done = -1; snprintf(b->err, sizeof(b->err), "cannot create an inflate thread"); }
    b->n_teams = i;
    b->inf_started = 1;
    if(!b->reap_started && reap_wanted(b) && pthread_create(&b->reap_th, NULL, reaper_main, b) == 0) b->reap_started = 1;
    if(b->dev && b->n_gpu_teams) { int k; for(k = 0; k < b->n_gpu_teams; k++) { team_arg *ta = malloc(sizeof(*ta)); if(!ta) break; ta->b = b; ta->gpu_team = k; ta->idx = k; if(pthread_create(&b->gpu_th[k], NULL, inflater_main, ta)) { free(ta); break; } } b->n_gpu_teams = k; b->gpu_started = 1; }
}
static void inflaters_stop(mdk_bam *b) {
    int i;
    if(!b->inf_started) return;
    pthread_mutex_lock(&b->mu); b->quit = 1; pthread_cond_broadcast(&b->cv_pool); pthread_cond_broadcast(&b->cv_q); pthread_mutex_unlock(&b->mu);
    for(i = 0; i < b->n_teams; i++) pthread_join(b->inf_th[i], NULL);
    if(b->gpu_started) { for(i = 0; i < b->n_gpu_teams; i++) pthread_join(b->gpu_th[i], NULL); b->gpu_started = 0; }
    b->inf_started = 0;
}
int mdk_bam_attach_device(mdk_bam *b, struct md_dev *dev, int n_teams) {
    int k;
    if(!b || !dev || b->dev) return -1;
    if(n_teams < 1) n_teams = 1;
    if(n_teams > MDK_GPU_TEAMS_MAX) n_teams = MDK_GPU_TEAMS_MAX;
    pthread_mutex_lock(&b->life_mu);                              /* (the reader thread may be inside a seek, which stops and restarts every team) */
    __atomic_store_n(&b->dev, dev, __ATOMIC_RELEASE);             /* (the host teams, already running, look at it without a lock: inflater_main) */
    /* A device piece is a whole number of the device's ROUNDS of members: k_inflate keeps a fixed number of wavefronts resident, each inflating one member at a
     * time, so 5,135 members on 2,560 wavefronts took three members' time where 5,120 take two (a 96 MB piece: 3.8 ms instead of 2.6, gpurun_out r06l).
     * MDK_GPU_PIECE_ROUNDS=n (default 2); a piece size given in bytes (MDK_GPU_PIECE_MB, tests) stands as it is. */
    if(!getenv("MDK_GPU_PIECE_MB")) {
        const int per = md_piece_members_per_round(dev); int rounds = getenv("MDK_GPU_PIECE_ROUNDS") ? atoi(getenv("MDK_GPU_PIECE_ROUNDS")) : 2;
        if(rounds < 1) rounds = 1;
        if(rounds > 8) rounds = 8;
        pthread_mutex_lock(&b->io_mu);
        if(per > 0) { b->gpu_piece_members = per * rounds; b->gpu_piece_bytes = (size_t)b->gpu_piece_members * 28672u; if(b->gpu_piece_bytes > (240u << 20)) b->gpu_piece_bytes = 240u << 20; }
        pthread_mutex_unlock(&b->io_mu);
    }
    b->n_gpu_teams = n_teams; b->max_dalloc = n_teams + (getenv("MDK_DSLAB_EXTRA") && atoi(getenv("MDK_DSLAB_EXTRA")) >= 1 ? atoi(getenv("MDK_DSLAB_EXTRA")) : 8);      /* (a chunk read in place keeps its piece until its results are in: md_dev_upload_raw_inplace) */
    if(b->inf_started) {
        for(k = 0; k < n_teams; k++) { team_arg *ta = malloc(sizeof(*ta)); if(!ta) break; ta->b = b; ta->gpu_team = k; ta->idx = k; if(pthread_create(&b->gpu_th[k], NULL, inflater_main, ta)) { free(ta); break; } }
        b->n_gpu_teams = k; b->gpu_started = 1;
    }
    pthread_mutex_unlock(&b->life_mu);
    return 0;
}
static void slab_destroy(mdk_slab *s) { if(!s) return; if(s->piece) md_piece_destroy(s->piece); md_host_free(s->buf); free(s->sum); free(s->off32); free(s->mem); free(s); }
void mdk_bam_detach_device(mdk_bam *b) {
    int i;
    if(!b || !b->dev) return;
    pthread_mutex_lock(&b->life_mu);
    /* the device teams end; slabs they made that are still queued or held are destroyed with the reader (mdk_bam_close) or here */
    pthread_mutex_lock(&b->mu); b->quit = 1; if(!b->inf_done) b->inf_done = 1; pthread_cond_broadcast(&b->cv_pool); pthread_cond_broadcast(&b->cv_q); pthread_mutex_unlock(&b->mu);
    if(b->gpu_started) { for(i = 0; i < b->n_gpu_teams; i++) pthread_join(b->gpu_th[i], NULL); b->gpu_started = 0; }
    for(i = 0; i < b->n_teams && b->inf_started; i++) pthread_join(b->inf_th[i], NULL);
    b->inf_started = 0;
    pthread_mutex_lock(&b->mu);
    for(i = 0; i < MDK_READY; i++) if(b->ready[i] && b->ready[i]->piece) { slab_destroy(b->ready[i]); b->ready[i] = NULL; b->n_ready--; }
    for(i = 0; i < b->n_dpool; i++) slab_destroy(b->dpool[i]);
    b->n_dpool = 0;
    if(b->cur && b->cur->piece) { slab_destroy(b->cur); b->cur = NULL; }
    pthread_mutex_unlock(&b->mu);
    for(i = 0; i < MDK_GPU_TEAMS_MAX; i++) { md_host_free(b->gpu_stage[i]); b->gpu_stage[i] = NULL; b->gpu_stage_cap[i] = 0; }
    __atomic_store_n(&b->dev, (md_dev *)NULL, __ATOMIC_RELEASE); b->n_gpu_teams = 0;
    pthread_mutex_unlock(&b->life_mu);
}

/* scanner side: next inflated slab (blocking); NULL at end of data or on error (b->inf_done < 0) */
static mdk_slab *slab_next(mdk_bam *b) {
    mdk_slab *s = NULL; double t0 = io_now();
    pthread_mutex_lock(&b->mu);
    pthread_cond_broadcast(&b->cv_pool);        /* the team that holds piece pop_seq may take a slab beyond the cap */
    for(;;) {
        mdk_slab **slot = &b->ready[b->pop_seq % MDK_READY];
        if(*slot) { s = *slot; *slot = NULL; b->n_ready--; b->pop_seq++; break; }
        if(b->inf_done < 0 || b->quit) break;
        if(b->io_end) {                /* the file has ended: done once every piece handed out has been taken */
            uint64_t handed; pthread_mutex_unlock(&b->mu); pthread_mutex_lock(&b->io_mu); handed = b->next_seq; pthread_mutex_unlock(&b->io_mu); pthread_mutex_lock(&b->mu);
            if(b->ready[b->pop_seq % MDK_READY]) continue;
            if(b->pop_seq >= handed) { if(!b->inf_done) b->inf_done = 1; break; }
        }
        pthread_cond_wait(&b->cv_q, &b->mu);
    }
    pthread_cond_broadcast(&b->cv_pool);          /* a place is free / the scanner is about to starve: let the inflaters go on */
    pthread_mutex_unlock(&b->mu);
    b->t_inflate += io_now() - t0;
    return s;
}

/* make at least n bytes available at the scan position, moving to the next slab when the current one runs out
 * (the unfinished tail is completed in the next slab's headroom); 1 ok, 0 clean end of data, <0 error.
 * A slab inflated on the device becomes current with off == end: the caller sees that through mdk_bam_at_device. */
/* a piece that does not fit the one before it: everything the teams hold is thrown away and the rest of the file is framed under the lock, from the last boundary verified */
static void spec_redo(mdk_bam *b) {
    int i;
    pthread_mutex_lock(&b->life_mu);
    inflaters_stop(b);
    pthread_mutex_lock(&b->mu);
    for(i = 0; i < MDK_READY; i++) if(b->ready[i]) { mdk_slab *q = b->ready[i]; b->ready[i] = NULL; q->refs = 1; pthread_mutex_unlock(&b->mu); mdk_slab_unref(b, q); pthread_mutex_lock(&b->mu); }
    b->n_ready = 0; b->quit = 0; b->inf_done = 0; b->clen = 0; b->file_eof = 0; b->err[0] = 0;
    pthread_mutex_unlock(&b->mu);
    b->map_pos = b->spec_verified; b->n_spec_redo++; b->spec_off = 1;      /* (no team is running) */
    inflaters_start(b);
    pthread_mutex_unlock(&b->life_mu);
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk host] a piece cut without the lock did not fit at byte %zu of the file: framing under the lock from there\n", b->spec_verified);
}
static int slab_all_ok(const mdk_slab *s) { int i; for(i = 0; i < s->n_mem; i++) if(!s->mem[i].ok) return 0; return 1; }
static int need(mdk_bam *b, size_t n) {
    for(;;) {
        mdk_slab *s; size_t left;
        if(b->cur && b->cur->piece) { if(b->mem_i < b->cur->n_mem) return 2; }          /* standing in a device slab: nothing to read through */
        else if(b->cur && b->cur->end - b->off >= n) return 1;
        s = slab_next(b); left = (b->cur && !b->cur->piece) ? b->cur->end - b->off : 0;
        if(s && s->spec) {      /* a piece cut without the lock is good if it begins where the piece before it ended (claim_range) */
            if(s->spec_fail || s->file_beg != b->spec_verified) { mdk_slab_unref(b, s); spec_redo(b); continue; }
            b->spec_verified = s->file_end;
        }
        if(!s) {
            if(b->inf_done < 0) return b->inf_done;
            if(left == 0) return 0;
            snprintf(b->err, sizeof(b->err), "truncated BAM record at end of file"); return -2;
        }
        if(s->piece && (left || !slab_all_ok(s))) { mdk_slab *h = slab_materialize(b, s); mdk_slab_unref(b, s); if(!h) return -2; s = h; }
        if(s->piece) {
            if(b->cur) mdk_slab_unref(b, b->cur);
            b->cur = s; b->off = 0; b->mem_i = 0; b->sum_i = b->sum_end = 0;
            continue;
        }
        if(left > s->beg) { snprintf(b->err, sizeof(b->err), "BAM record larger than %u bytes", MDK_SLAB_HEADROOM); mdk_slab_unref(b, s); return -2; }
        if(left) { memcpy(s->buf + s->beg - left, b->cur->buf + b->off, left); s->beg -= left; }
        if(b->cur) mdk_slab_unref(b, b->cur);
        b->cur = s; b->off = s->beg; b->mem_i = 0; b->sum_i = b->sum_end = 0;
    }
}
int mdk_bam_at_device(mdk_bam *b, mdk_slab **s, int *mi) {
    int rc;
    if(b->sum_i < b->sum_end) return 0;                        /* inside a host member's summaries */
    rc = need(b, 1);
    if(rc < 0) return rc;
    if(rc != 2) return 0;
    *s = b->cur; *mi = b->mem_i;
    return 1;
}
void mdk_bam_dev_advance(mdk_bam *b) { if(b->cur && b->cur->piece && b->mem_i < b->cur->n_mem) { b->n_records += b->cur->mem[b->mem_i].n_sum; b->n_fast += b->cur->mem[b->mem_i].n_sum; b->mem_i++; } }

/* the file could be opened but not read as a BAM: say why (a damaged BGZF member, a failed CRC32 check) before the caller's "Couldn't open" */
static mdk_bam *open_fail(mdk_bam *b, const char *fn) { if(b->err[0]) fprintf(stderr, "[mdk] %s: %s\n", fn, b->err); mdk_bam_close(b); return NULL; }
mdk_bam *mdk_bam_open(const char *fn, int nthreads) {
    mdk_bam *b = xcalloc(1, sizeof(*b)); int rc; uint32_t i; size_t o;
    if(!b) return NULL;
    b->f = fopen(fn, "rb");
    if(!b->f) { free(b); return NULL; }
    if(!getenv("MDK_NO_MMAP")) {
        struct stat st;
        if(fstat(fileno(b->f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(b->f), 0);
            if(m != MAP_FAILED) { b->map = m; b->map_len = (size_t)st.st_size; b->map_pos = 0; (void)madvise(m, b->map_len, MADV_SEQUENTIAL); }
            if(getenv("MDK_RESERVE_HINT")) {      /* (off by default: the allocations it spares the streaming phase -- 10-30 ms calls next to the pieces' copies, profiles/r06pf_slow_calls.txt -- do not show in the wall clock, and 15 GiB made ahead cost the 512 Mb run 0-5 % in three of four interleaved comparisons, profiles/r06_e2e_ab.txt) */       /* what the device will hold at once: up to 16 pieces being inflated or read in place (0.64 GB each), the chunks' slots and the contigs in use */
                uint64_t pieces = (uint64_t)st.st_size / (96ull << 20) + 2; if(pieces > 16) pieces = 16;
                md_dev_reserve_hint(pieces * (640ull << 20) + (6ull << 30));
            }
        }
    }
    b->nthreads = nthreads < 1 ? 1 : nthreads;
    /* how far the host teams may run ahead of the consumers, in slabs (~27 MB inflated each: one piece of CCHUNK compressed bytes).  A dozen:
     * every slab is registered with the runtime the first time it is uploaded from (1 ms), and what a process has registered costs the
     * kernel ~0.13 s per GB when the process ends (profiles/r04d_*: 60 slabs made `extract` of a 2 GB BAM 0.45 s slower than 30 did).  The
     * teams that find no slab wait; the device inflates what they do not get to. */
    b->max_alloc = b->nthreads >= 8 ? 12 : b->nthreads + 4;
    if(getenv("MDK_SLAB_CAP")) b->max_alloc = atoi(getenv("MDK_SLAB_CAP")) > 1 ? atoi(getenv("MDK_SLAB_CAP")) : 2;
    pthread_mutex_init(&b->mu, NULL); pthread_mutex_init(&b->io_mu, NULL); pthread_mutex_init(&b->life_mu, NULL); pthread_cond_init(&b->cv_q, NULL); pthread_cond_init(&b->cv_pool, NULL); pthread_cond_init(&b->cv_reap, NULL); pthread_cond_init(&b->cv_reaped, NULL);
    b->n_teams = b->nthreads >= 32 ? 4 : b->nthreads >= 8 ? 2 : 1;
    if(getenv("MDK_DEVICE_INFLATE_ONLY")) { b->host_leaves = 1; b->n_teams = 1; }
    b->gpu_piece_bytes = GCHUNK;
    if(getenv("MDK_GPU_PIECE_MB") && atof(getenv("MDK_GPU_PIECE_MB")) >= 0.25 && atof(getenv("MDK_GPU_PIECE_MB")) <= 256) b->gpu_piece_bytes = (size_t)(atof(getenv("MDK_GPU_PIECE_MB")) * 1048576.0);      /* test hook: many small device pieces */
    if(getenv("MDK_INFLATE_TEAMS")) { b->n_teams = atoi(getenv("MDK_INFLATE_TEAMS")); if(b->n_teams < 1) b->n_teams = 1; if(b->n_teams > 8) b->n_teams = 8; }
    b->team_threads = (b->nthreads + b->n_teams - 1) / b->n_teams;
    (void)crc_wanted();                                           /* (decided here, once, before the threads that ask) */
    inflaters_start(b);
    if((rc = need(b, 12)) <= 0 || memcmp(b->cur->buf + b->off, "BAM\1", 4)) return open_fail(b, fn);
    b->l_text = le32(b->cur->buf + b->off + 4);
    if(need(b, 12 + (size_t)b->l_text) <= 0) return open_fail(b, fn);
    b->text = xmalloc((size_t)b->l_text + 1); memcpy(b->text, b->cur->buf + b->off + 8, b->l_text); b->text[b->l_text] = 0;
    b->n_targets = (int32_t)le32(b->cur->buf + b->off + 8 + b->l_text);
    b->off += 12 + (size_t)b->l_text;
    b->target_name = xcalloc((size_t)b->n_targets + 1, sizeof(char *)); b->target_len = xcalloc((size_t)b->n_targets + 1, sizeof(uint32_t));
    for(i = 0; i < (uint32_t)b->n_targets; i++) {
        uint32_t ln;
        if(need(b, 4) <= 0) return open_fail(b, fn);
        ln = le32(b->cur->buf + b->off);
        if(need(b, 8 + (size_t)ln) <= 0) return open_fail(b, fn);
        o = b->off;
        b->target_name[i] = xmalloc((size_t)ln + 1); memcpy(b->target_name[i], b->cur->buf + o + 4, ln); b->target_name[i][ln] = 0;
        b->target_len[i] = le32(b->cur->buf + o + 4 + ln);
        b->off += 8 + (size_t)ln;
    }
    pthread_mutex_lock(&b->mu); b->header_done = 1; pthread_mutex_unlock(&b->mu);
    return b;
}

void mdk_bam_close(mdk_bam *b) {
    int i;
    if(!b) return;
    inflaters_stop(b);
    reaper_stop(b);
    for(i = 0; i < b->n_reap; i++) slab_destroy(b->reap[i]);
    free(b->reap);
    if(b->cur) slab_destroy(b->cur);
    for(i = 0; i < MDK_READY; i++) if(b->ready[i]) slab_destroy(b->ready[i]);
    for(i = 0; i < b->n_pool; i++) slab_destroy(b->pool[i]);
    for(i = 0; i < b->n_dpool; i++) slab_destroy(b->dpool[i]);
    for(i = 0; i < MDK_GPU_TEAMS_MAX; i++) md_host_free(b->gpu_stage[i]);
    free(b->pool); free(b->dpool);
    if(b->f) fclose(b->f);
    if(b->target_name) for(i = 0; i < b->n_targets; i++) free(b->target_name[i]);
    free(b->target_name); free(b->target_len); free(b->text); if(b->map) munmap((void *)b->map, b->map_len); else free(b->cbuf);
    pthread_mutex_destroy(&b->mu); pthread_mutex_destroy(&b->io_mu); pthread_mutex_destroy(&b->life_mu); pthread_cond_destroy(&b->cv_q); pthread_cond_destroy(&b->cv_pool); pthread_cond_destroy(&b->cv_reap); pthread_cond_destroy(&b->cv_reaped);
    free(b);
}

void mdk_bam_abort(mdk_bam *b) {
    if(!b) return;
    pthread_mutex_lock(&b->mu); b->quit = 1; if(!b->inf_done) b->inf_done = 1; pthread_cond_broadcast(&b->cv_q); pthread_cond_broadcast(&b->cv_pool); pthread_mutex_unlock(&b->mu);
}

mdk_slab *mdk_bam_cur_slab(mdk_bam *b, size_t *off) { *off = b->off; return b->cur; }

int mdk_rec_parse(const uint8_t *r, uint32_t len, mdk_rec *o) {
    if(len < 32) return -1;
    o->raw = r; o->raw_len = len;
    o->tid = (int32_t)le32(r); o->pos = (int32_t)le32(r + 4); o->l_qname = r[8]; o->mapq = r[9];
    o->n_cigar = le16(r + 12); o->flag = le16(r + 14); o->l_qseq = (int32_t)le32(r + 16);
    o->mtid = (int32_t)le32(r + 20); o->mpos = (int32_t)le32(r + 24);
    o->qname = (const char *)(r + 32);
    o->cigar = r + 32 + o->l_qname; o->seq = o->cigar + 4u * o->n_cigar;
    if(o->l_qseq < 0) return -1;
    o->qual = o->seq + ((size_t)o->l_qseq + 1) / 2; o->aux = o->qual + o->l_qseq;
    if(o->aux > r + len) return -1;
    o->aux_len = (int32_t)((r + len) - o->aux);
    return 0;
}

int mdk_bam_peek(mdk_bam *b, mdk_rec *r) {
    int rc = need(b, 4); uint32_t bs;
    if(rc <= 0 || rc == 2) return rc;            /* 2: the scanner stands in a slab inflated on the device (mdk_bam_at_device) */
    bs = le32(b->cur->buf + b->off);
    rc = need(b, 4 + (size_t)bs);
    if(rc <= 0) { if(rc == 0) { snprintf(b->err, sizeof(b->err), "truncated BAM record at end of file"); return -2; } return rc; }
    if(mdk_rec_parse(b->cur->buf + b->off + 4, bs, r) != 0) { snprintf(b->err, sizeof(b->err), "malformed BAM record"); return -2; }
    return 1;
}
void mdk_bam_advance(mdk_bam *b, const mdk_rec *r) { b->off += 4 + (size_t)r->raw_len; b->n_records++; }

int mdk_bam_peek_sum(mdk_bam *b, mdk_rsum *o, const uint8_t **raw) {
    mdk_rec r; int rc, k;
    for(;;) {
        if(b->sum_i < b->sum_end) {                       /* inside an ok member: the inflating thread has been here already */
            *o = b->cur->sum[b->sum_i]; *raw = b->cur->buf + o->off + 4;
            return 1;
        }
        rc = need(b, 4);
        if(rc <= 0 || rc == 2) return rc;
        {   /* does an ok member begin exactly here? */
            mdk_slab *s = b->cur;
            while(b->mem_i < s->n_mem && s->mem[b->mem_i].off < b->off) b->mem_i++;
            if(b->mem_i < s->n_mem && s->mem[b->mem_i].off == b->off && s->mem[b->mem_i].ok) {
                const mdk_member *m = &s->mem[b->mem_i++];
                if(m->n_sum) { b->sum_i = m->sum0; b->sum_end = (size_t)m->sum0 + m->n_sum; }
                continue;                                 /* an empty member: the next one starts at the same place */
            }
        }
        break;
    }
    rc = mdk_bam_peek(b, &r);
    if(rc <= 0) return rc;
    {
        int32_t rl = 0;
        for(k = 0; k < r.n_cigar; k++) { uint32_t v = le32(r.cigar + 4 * k), op = v & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(v >> 4); }
        o->off = (uint32_t)(b->off); o->len = r.raw_len; o->tid = r.tid; o->pos = r.pos; o->endp = r.pos + (rl > 0 ? rl : 1);
    }
    *raw = r.raw;
    return 1;
}
int mdk_bam_member_run(mdk_bam *b, const mdk_rsum **v, size_t *n, const mdk_member **m) {
    if(b->sum_i >= b->sum_end || !b->cur || b->mem_i < 1) return 0;
    *v = b->cur->sum + b->sum_i; *n = b->sum_end - b->sum_i; *m = &b->cur->mem[b->mem_i - 1];
    return 1;
}
void mdk_bam_advance_run(mdk_bam *b, size_t k) {
    const mdk_rsum *last;
    if(!k) return;
    last = &b->cur->sum[b->sum_i + k - 1];
    b->sum_i += k; b->n_fast += k; b->n_records += k;
    b->off = (size_t)last->off + 4 + (size_t)last->len;
}
void mdk_bam_advance_sum(mdk_bam *b, const mdk_rsum *r) {
    if(b->sum_i < b->sum_end) { b->sum_i++; b->n_fast++; } else b->n_slow++;
    b->off = (size_t)r->off + 4 + (size_t)r->len; b->n_records++;
}

/* ---- BAI + seeking ---- */
mdk_bai *mdk_bai_load(const char *bam_fn) {
    char fn[4096]; FILE *f; uint8_t *d; size_t sz, o = 8; mdk_bai *x; int32_t r;
    snprintf(fn, sizeof(fn), "%s.bai", bam_fn); f = fopen(fn, "rb");
    if(!f) { size_t l = strlen(bam_fn); if(l > 4 && l + 1 < sizeof(fn)) { memcpy(fn, bam_fn, l - 4); strcpy(fn + l - 4, ".bai"); f = fopen(fn, "rb"); } }
    if(!f) return NULL;
    fseek(f, 0, SEEK_END); sz = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    d = malloc(sz + 8);
    if(!d || fread(d, 1, sz, f) != sz || sz < 8 || memcmp(d, "BAI\1", 4)) { fclose(f); free(d); return NULL; }
    fclose(f);
    x = xcalloc(1, sizeof(*x)); x->n_ref = (int32_t)le32(d + 4);
    x->n_intv = xcalloc((size_t)x->n_ref + 1, 4); x->ioff = xcalloc((size_t)x->n_ref + 1, sizeof(uint64_t *)); x->first = xcalloc((size_t)x->n_ref + 1, 8);
    for(r = 0; r < x->n_ref; r++) {
        int32_t nb, b, ni, i; uint64_t first = 0;
        if(o + 4 > sz) goto bad;
        nb = (int32_t)le32(d + o); o += 4;
        for(b = 0; b < nb; b++) {
            uint32_t bin; int32_t nc, c;
            if(o + 8 > sz) goto bad;
            bin = le32(d + o); nc = (int32_t)le32(d + o + 4); o += 8;
            if(o + 16u * (size_t)nc > sz) goto bad;
            if(bin != 37450) for(c = 0; c < nc; c++) { uint64_t cb = (uint64_t)le32(d + o + 16 * c) | ((uint64_t)le32(d + o + 16 * c + 4) << 32); if(!first || cb < first) first = cb; }
            o += 16u * (size_t)nc;
        }
        if(o + 4 > sz) goto bad;
        ni = (int32_t)le32(d + o); o += 4;
        if(o + 8u * (size_t)ni > sz) goto bad;
        x->n_intv[r] = ni; x->ioff[r] = xmalloc(8u * (size_t)(ni + 1)); x->first[r] = first;
        for(i = 0; i < ni; i++) x->ioff[r][i] = (uint64_t)le32(d + o + 8 * i) | ((uint64_t)le32(d + o + 8 * i + 4) << 32);
        o += 8u * (size_t)ni;
    }
    free(d);
    return x;
bad:
    free(d); mdk_bai_free(x);
    return NULL;
}
void mdk_bai_free(mdk_bai *x) { int32_t r; if(!x) return; for(r = 0; r < x->n_ref; r++) free(x->ioff[r]); free(x->ioff); free(x->n_intv); free(x->first); free(x); }
uint64_t mdk_bai_start(const mdk_bai *x, int32_t tid, int64_t beg) {
    int64_t w;
    if(!x || tid < 0 || tid >= x->n_ref || !x->first[tid]) return 0;
    w = beg >> 14;
    if(w >= x->n_intv[tid]) return 0;                 /* nothing overlaps this window or any later one */
    for(; w >= 0; w--) if(x->ioff[tid][w]) return x->ioff[tid][w];
    return x->first[tid];
}

int mdk_bam_seek(mdk_bam *b, uint64_t voffset) {
    int i;
    pthread_mutex_lock(&b->life_mu);
    inflaters_stop(b);
    pthread_mutex_lock(&b->mu);
    for(i = 0; i < MDK_READY; i++) if(b->ready[i]) { mdk_slab *q = b->ready[i]; b->ready[i] = NULL; q->refs = 1; pthread_mutex_unlock(&b->mu); mdk_slab_unref(b, q); pthread_mutex_lock(&b->mu); }
    b->n_ready = 0; b->quit = 0; b->inf_done = 0; b->clen = 0; b->file_eof = 0; b->seeked = 1;
    pthread_mutex_unlock(&b->mu);
    if(b->cur) { mdk_slab_unref(b, b->cur); b->cur = NULL; }
    if(b->map) { if((size_t)(voffset >> 16) > b->map_len) { snprintf(b->err, sizeof(b->err), "seek failed"); pthread_mutex_unlock(&b->life_mu); return -2; } b->map_pos = (size_t)(voffset >> 16); __atomic_store_n(&b->pop_next, b->map_pos, __ATOMIC_RELAXED); }
    else if(fseeko(b->f, (off_t)(voffset >> 16), SEEK_SET)) { snprintf(b->err, sizeof(b->err), "seek failed"); pthread_mutex_unlock(&b->life_mu); return -2; }
    inflaters_start(b);
    pthread_mutex_unlock(&b->life_mu);
    {
        int rc = need(b, (size_t)(voffset & 0xffff) + 1);
        if(rc < 0) return rc;
        if(rc == 0) return 0;                          /* nothing there */
        if(rc == 2) return 1;                          /* a slab inflated on the device is taken from the first record of its first member: the
                                                          records in front of the offset end before the window the caller asked for, and it passes over them */
        b->off += (size_t)(voffset & 0xffff);
    }
    return 1;
}

/* (the FASTA loader: mdk_fasta.c) */

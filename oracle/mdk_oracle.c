/*
 * mdk_oracle.c -- CPU ORACLE for the `MethylDackel extract` hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py may build, link or run it.
 * The shipped path (methyldackel_amd/) never includes, links or executes anything here.
 *
 * What it is: a plain-C, single-threaded, deliberately literal restatement of the
 * reference algorithm (MethylDackel 0.6.1, /root/reference) for `extract`:
 *   - option surface / validation / return codes      extract.c:706-1069,1343-1514
 *   - chunk scheduler + adjustBounds                   extract.c:325-378, common.c:466-493
 *   - read admission (filter_func) + trimming          common.c:137-208,407-463
 *   - mappability filter (BBM input)                   common.c:210-335, extract.c:1236-1339
 *   - conversion-efficiency filter                     common.c:338-404
 *   - mate-overlap quality resolution                  overlaps.c:27-147
 *   - per-column pileup loop, variant filter, merge    extract.c:399-510
 *   - text emitters                                    extract.c:39-99,207-222,562-569
 * plus a restatement of the behaviour of the third-party engine the reference sits on
 * (htslib >= 1.11, NOT vendored in the reference and NOT installed in this image): BGZF/BAM
 * decode, region iteration (sam_itr_queryi), the bam_plp/bam_mplp pileup buffer with its
 * constructor/destructor callbacks, bam_aux_get/bam_aux2i, faidx_fetch_seq, hts_parse_reg.
 * Those are restated from the htslib API contract and its published algorithm.
 *
 * PARITY STATUS: "weakly pinned".  The reference cannot be compiled here (htslib and
 * libBigWig are absent, no network), so the only pins are the reference's own test
 * expectations (tests/test.py: 15 CLI runs asserting output LINE COUNTS on the fixture
 * BAMs that are copied, as data, under tests/golden/).  tests/test_oracle_reference_vectors.py
 * replays all 15.  14 agree; case 8 (--nOT 50,50,40,40 -> reference asserts 12 lines) yields
 * 11 lines here, and by hand-execution of common.c:174-208 + overlaps.c:54-119 -- documented in
 * DESIGN.md.  No byte-level golden output exists in the reference.
 *
 * Not supported (reference gets them only via libraries absent here): CRAM input, bigWig (-M).
 * BED (-l/--keepStrand): bed.c restated below (parseBED, spanOverlapsBED, posOverlapsBED, readStrandOverlapsBED).
 * `mbias`: MBias.c:16-573 and svg.c:8-454 restated below (extractMBias, mbias_main, makeSVGs, makeTXT, getThresholds).
 * `perRead`: perRead.c:16-464 restated below (processRead, perReadMetrics, perRead_main).
 * PARITY FOR BED, MBIAS AND PERREAD IS UNPINNED: tests/test.py never runs `-l`, `mbias` or `perRead`, so nothing of the reference's own
 * pins these; they rest on this restatement alone.
 *
 * Style note: this file follows the reference's control flow one step at a time (a real pileup
 * buffer swept column by column, reads copied into it, qualities rewritten in place).  The
 * product does none of that -- it is a read-parallel scatter on the GPU -- which is what makes
 * comparing the two meaningful.
 */
#define _GNU_SOURCE
#include <assert.h>
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
#include <zlib.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <dlfcn.h>
#include <time.h>

#define ORACLE_VERSION "0.6.1"
/* Differential-search knob (tests/t8_differential.py): MDK_ORACLE_PERTURB=<n> makes ONE rule deviate
 * from the reference's code, to find out which single deviation would reproduce an expectation the code contradicts.
 * 0 (the default) is the reference's behaviour; nothing else is ever used by a parity test. */
static int g_pt = 0; int g_pt_abs[16];
enum { PT_ABS_RIGHT_M1 = 1, PT_ABS_LEFT_M1, PT_ABS_RIGHT_P1, PT_ABS_LEFT_P1, PT_ABS_SWAP_R1R2, PT_ABS_AS_RELATIVE, PT_ABS_5PRIME, PT_ABS_QUAL_ONLY,
       PT_ABS_N_ONLY, PT_TRIM_AFTER_PAIRING, PT_SWAP_A_B, PT_TIE_FIRST, PT_SKIP_N_IN_OVERLAP, PT_NO_OVERLAP, PT_ABS_SKIP_READ1, PT_ABS_SKIP_READ2, PT_ADMIT_QCFAIL, PT_N };
#define RUNOFFSET 99
#define BBM_VERSION 1

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                                */
/* ------------------------------------------------------------------------------------------ */
static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if(!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); } return p; }
static void *xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if(!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); } return p; }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

typedef struct { char *s; size_t l, m; } kstr;
static void kputs_(kstr *k, const char *s) {
    size_t n = strlen(s);
    if(k->l + n + 1 > k->m) { k->m = (k->l + n + 1) * 2; k->s = xrealloc(k->s, k->m); }
    memcpy(k->s + k->l, s, n + 1); k->l += n;
}

/* ------------------------------------------------------------------------------------------ */
/* BAM file, fully inflated in memory (BGZF: concatenated gzip members, SAM spec 4.1)           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t tid, pos, l_qseq, mtid, mpos;
    uint16_t flag, n_cigar;
    uint8_t mapq, l_qname;
    const char *qname;
    const uint8_t *cigar, *seq, *qual, *aux;   /* cigar is unaligned little-endian u32[] */
    int32_t aux_len;
    int32_t rlen;                              /* raw reference length of the CIGAR */
} brec;

typedef struct {
    uint8_t *data; size_t len;
    int32_t n_targets; char **target_name; uint32_t *target_len;
    brec *rec; size_t n_rec;
    size_t *tid_lo, *tid_hi;                   /* record index range per tid (sorted input) */
    int32_t max_rlen;
} bamfile;

static uint32_t cig_op(const uint8_t *c, int i) { return rd32(c + 4 * i) & 0xf; }
static uint32_t cig_len(const uint8_t *c, int i) { return rd32(c + 4 * i) >> 4; }

/* bam_cigar2rlen: M,D,N,=,X consume the reference */
static int32_t cigar2rlen(const uint8_t *c, int n) {
    int32_t l = 0; int i;
    for(i = 0; i < n; i++) { uint32_t op = cig_op(c, i); if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += cig_len(c, i); }
    return l;
}
/* bam_endpos(): pos + rlen, with rlen==0 treated as 1 */
static int32_t rec_endpos(const brec *r) { return r->pos + (r->rlen > 0 ? r->rlen : 1); }

/* -@ N: the reference gives every worker its own file handle and region iterator (extract.c:283-295), so BGZF inflate
 * and record decoding scale with the workers.  This oracle keeps the whole file in memory instead, so the same work is
 * spread over the N threads up front: members are inflated block-parallel, records are decoded range-parallel. */
static int g_load_threads = 1;
static double wall_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
/* MDK_ORACLE_PROFILE=1: wall clock per phase on stderr; a phase one thread does alone counts as serial (SERIAL), the others are spread over the -@ N threads */
static double g_t0, g_t_serial, g_t_all;
#define PHASE_(name, serial) do { double t_ = wall_s(); g_t_all += t_ - g_t0; if(serial) g_t_serial += t_ - g_t0; if(getenv("MDK_ORACLE_PROFILE")) fprintf(stderr, "[oracle] %-22s %.3f s%s\n", name, t_ - g_t0, serial ? "  (one thread)" : ""); g_t0 = t_; } while(0)
#define PHASE(name) PHASE_(name, 0)
#define PHASE_SERIAL(name) PHASE_(name, 1)
typedef struct { const uint8_t *raw; const size_t *moff, *mout; const uint32_t *mlen, *misz, *mhdr; size_t nm; uint8_t *out; int k, n, bad;
                 size_t *roff; size_t n_roff, cap_roff; uint32_t *mcount; } inflate_job;      /* roff/mcount: the records each member holds when it starts on a record boundary */
/* htslib never lets a record straddle two BGZF members (bam_write1 flushes first), so a member normally starts on a record
 * boundary: the thread that inflated it walks it at once.  mcount[i] = records found, or UINT32_MAX when the walk does not
 * end exactly at the member's last byte (then the whole file is walked serially instead). */
static void walk_member(inflate_job *j, size_t i) {
    const uint8_t *d = j->out + j->mout[i]; size_t L = j->misz[i], o = 0; uint32_t n = 0;
    while(o + 4 <= L) {
        uint32_t bs = rd32(d + o);
        if(bs < 32 || o + 4 + (size_t)bs > L) break;
        if(j->n_roff == j->cap_roff) { j->cap_roff = j->cap_roff ? j->cap_roff * 2 : 65536; j->roff = xrealloc(j->roff, j->cap_roff * sizeof(size_t)); }
        j->roff[j->n_roff++] = j->mout[i] + o; n++;
        o += 4 + (size_t)bs;
    }
    if(o != L) { j->n_roff -= n; n = UINT32_MAX; }
    j->mcount[i] = n;
}
/* htslib inflates BGZF members with libdeflate when it is built with it (two to three times faster than zlib); the image has
 * the runtime library without its header, so it is bound by name here too -- the CPU baseline should not be slower than the
 * reference would be.  MDK_ZLIB_INFLATE=1 or a missing library leave zlib. */
typedef struct { void *(*alloc)(void); int (*run)(void *, const void *, size_t, void *, size_t, size_t *); void (*release)(void *);  uint32_t (*crc)(uint32_t, const void *, size_t); } ldeflate_t;
static ldeflate_t g_ld; static int g_ld_state;
static void ldeflate_init(void) {
    void *so = getenv("MDK_ZLIB_INFLATE") ? NULL : dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    g_ld_state = -1;
    if(so) {
        g_ld.alloc = (void *(*)(void))dlsym(so, "libdeflate_alloc_decompressor");
        g_ld.run = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(so, "libdeflate_deflate_decompress");
        g_ld.release = (void (*)(void *))dlsym(so, "libdeflate_free_decompressor");
        g_ld.crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(so, "libdeflate_crc32");
        if(g_ld.alloc && g_ld.run && g_ld.release) g_ld_state = 1;
    }
}
static void *inflate_main(void *arg) {
    inflate_job *j = arg; size_t i; z_stream zs; void *ld = g_ld_state == 1 ? g_ld.alloc() : NULL;
    for(i = (size_t)j->k; i < j->nm; i += (size_t)j->n) {
        if(!j->misz[i]) { if(j->mcount) j->mcount[i] = 0; continue; }        /* an empty member (the EOF marker) holds no record */
        if(ld) {
            size_t got = 0;
            if(g_ld.run(ld, j->raw + j->moff[i] + j->mhdr[i], j->mlen[i] - j->mhdr[i] - 8, j->out + j->mout[i], j->misz[i], &got) != 0 || got != j->misz[i]) { j->bad = 1; break; }
            /* htslib's bgzf_read_block checks every block's CRC32 against its trailer (with libdeflate's crc32 when it is built with it) */
            if((g_ld.crc ? g_ld.crc(0, j->out + j->mout[i], j->misz[i]) : (uint32_t)crc32(0L, j->out + j->mout[i], j->misz[i])) != rd32(j->raw + j->moff[i] + j->mlen[i] - 8)) { j->bad = 1; break; }
            if(j->mcount) walk_member(j, i);
            continue;
        }
        memset(&zs, 0, sizeof(zs));
        zs.next_in = (uint8_t *)j->raw + j->moff[i] + j->mhdr[i]; zs.avail_in = j->mlen[i] - j->mhdr[i] - 8;
        zs.next_out = j->out + j->mout[i]; zs.avail_out = j->misz[i];
        if(inflateInit2(&zs, -15) != Z_OK) { j->bad = 1; return NULL; }
        if(inflate(&zs, Z_FINISH) != Z_STREAM_END) { inflateEnd(&zs); j->bad = 1; return NULL; }
        inflateEnd(&zs);
        if((uint32_t)crc32(0L, j->out + j->mout[i], j->misz[i]) != rd32(j->raw + j->moff[i] + j->mlen[i] - 8)) { j->bad = 1; return NULL; }
        if(j->mcount) walk_member(j, i);
    }
    if(ld) g_ld.release(ld);
    return NULL;
}
typedef struct { bamfile *bf; const size_t *roff; size_t lo, hi; int32_t max_rlen; int bad; } decode_job;
static void *decode_main(void *arg) {
    decode_job *j = arg; bamfile *bf = j->bf; size_t i;
    for(i = j->lo; i < j->hi; i++) {
        size_t o = j->roff[i]; uint32_t bs = rd32(bf->data + o); const uint8_t *r = bf->data + o + 4; brec *b = &bf->rec[i];
        b->tid = (int32_t)rd32(r); b->pos = (int32_t)rd32(r + 4);
        b->l_qname = r[8]; b->mapq = r[9];
        b->n_cigar = rd16(r + 12); b->flag = rd16(r + 14);
        b->l_qseq = (int32_t)rd32(r + 16); b->mtid = (int32_t)rd32(r + 20); b->mpos = (int32_t)rd32(r + 24);
        b->qname = (const char *)(r + 32);
        b->cigar = r + 32 + b->l_qname;
        b->seq = b->cigar + 4 * b->n_cigar;
        b->qual = b->seq + (b->l_qseq + 1) / 2;
        b->aux = b->qual + b->l_qseq;
        b->aux_len = (int32_t)((r + bs) - b->aux);
        if(b->aux_len < 0) { j->bad = 1; return NULL; }
        b->rlen = cigar2rlen(b->cigar, b->n_cigar);
        if(b->rlen > j->max_rlen) j->max_rlen = b->rlen;
    }
    return NULL;
}
/* the file is mapped, not read: every worker of the reference reads through its own handle (extract.c:283-295), nobody reads the whole file on one
 * thread first.  The mapping's pages are touched by the -@ N threads, each its share, so that the serial walk over the members' headers finds them */
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
typedef struct { const uint8_t *p; size_t len; } touch_job;
static void *touch_main(void *arg) {
    touch_job *j = arg; size_t o; volatile uint8_t sink = 0;
    if(j->len && madvise((void *)j->p, j->len, MADV_POPULATE_READ) != 0) for(o = 0; o < j->len; o += 4096) sink ^= j->p[o];
    (void)sink;
    return NULL;
}
typedef struct { bamfile *bf; size_t lo, hi; size_t *tlo, *thi; int bad; } range_job;      /* tlo/thi[tid]: first record / one past the last record of the contig inside [lo,hi) */
static void *range_main(void *arg) {
    range_job *j = arg; bamfile *bf = j->bf; size_t i; int32_t t;
    for(t = 0; t < bf->n_targets; t++) { j->tlo[t] = (size_t)-1; j->thi[t] = 0; }
    for(i = j->lo; i < j->hi; i++) {
        const brec *b = &bf->rec[i];
        if(b->tid < 0 || b->tid >= bf->n_targets) continue;
        if(j->tlo[b->tid] == (size_t)-1) j->tlo[b->tid] = i;
        else if(j->thi[b->tid] != i) { j->bad = 1; return NULL; }                       /* contigs interleaved */
        if(i > j->tlo[b->tid] && bf->rec[i - 1].pos > b->pos) { j->bad = 2; return NULL; }
        j->thi[b->tid] = i + 1;
    }
    return NULL;
}
typedef struct { size_t *dst; const size_t *first; const uint32_t *mcount; inflate_job *ijob; size_t m0, nm; int k, nt; } gather_job;
static void *gather_main(void *arg) {      /* thread k's members are i = k, k + nt, ...: their record offsets, noted in that order, go to their places in file order */
    gather_job *g = arg; size_t i, cur = 0;
    for(i = (size_t)g->k; i < g->nm; i += (size_t)g->nt) {
        if(g->mcount[i] == UINT32_MAX) continue;
        if(i >= g->m0) memcpy(g->dst + g->first[i], g->ijob[g->k].roff + cur, g->mcount[i] * sizeof(size_t));
        cur += g->mcount[i];
    }
    return NULL;
}
static int bam_load(const char *fn, bamfile *bf) {
    FILE *f = fopen(fn, "rb");
    uint8_t *raw; size_t rawlen, o = 0, cap, i, nm = 0, mcap = 1024, total = 0, *moff, *mout, *roff; uint32_t *mlen, *misz, *mhdr, *mcount = NULL; inflate_job *ijob = NULL;
    int nt = g_load_threads < 1 ? 1 : g_load_threads, k, bad = 0, mapped = 0;
    memset(bf, 0, sizeof(*bf));
    if(!f) return -1;
    g_t0 = wall_s();
    { struct stat st; void *m;
      if(fstat(fileno(f), &st) != 0 || st.st_size <= 0) { fclose(f); return -1; }
      rawlen = (size_t)st.st_size;
      m = mmap(NULL, rawlen, PROT_READ, MAP_PRIVATE, fileno(f), 0);
      if(m == MAP_FAILED) { raw = xmalloc(rawlen); if(fread(raw, 1, rawlen, f) != rawlen) { fclose(f); free(raw); return -1; } }       /* (not a mappable file: read it) */
      else {
          pthread_t *th = xmalloc(sizeof(pthread_t) * nt); touch_job *tj = xmalloc(sizeof(touch_job) * nt); const size_t per = ((rawlen / (size_t)nt) + 4095) & ~(size_t)4095;
          raw = m; mapped = 1;
          for(k = 0; k < nt; k++) { size_t a = per * (size_t)k, e = a + per; if(a > rawlen) a = rawlen; if(e > rawlen || k == nt - 1) e = rawlen; tj[k].p = raw + a; tj[k].len = e - a; if(k) pthread_create(&th[k], NULL, touch_main, &tj[k]); }
          touch_main(&tj[0]);
          for(k = 1; k < nt; k++) pthread_join(th[k], NULL);
          free(th); free(tj);
      }
      fclose(f); }
    PHASE("map file");
    /* member table (BGZF: concatenated gzip members with a 'BC' extra subfield holding the member size) */
    moff = xmalloc(mcap * sizeof(size_t)); mout = xmalloc(mcap * sizeof(size_t)); mlen = xmalloc(mcap * 4); misz = xmalloc(mcap * 4); mhdr = xmalloc(mcap * 4);
    while(o + 18 <= rawlen) {
        uint16_t xlen, bsize = 0; size_t x; int have = 0;
        if(raw[o] != 0x1f || raw[o + 1] != 0x8b || raw[o + 2] != 8 || !(raw[o + 3] & 4)) { if(mapped) munmap(raw, rawlen); else free(raw); return -2; }
        xlen = rd16(raw + o + 10);
        for(x = o + 12; x + 4 <= o + 12 + xlen;) {      /* find the 'BC' extra subfield */
            uint16_t slen = rd16(raw + x + 2);
            if(raw[x] == 'B' && raw[x + 1] == 'C' && slen == 2) { bsize = rd16(raw + x + 4); have = 1; }
            x += 4 + slen;
        }
        if(!have || o + bsize + 1 > rawlen) { if(mapped) munmap(raw, rawlen); else free(raw); return -2; }
        if(nm == mcap) { mcap *= 2; moff = xrealloc(moff, mcap * sizeof(size_t)); mout = xrealloc(mout, mcap * sizeof(size_t)); mlen = xrealloc(mlen, mcap * 4); misz = xrealloc(misz, mcap * 4); mhdr = xrealloc(mhdr, mcap * 4); }
        moff[nm] = o; mlen[nm] = (uint32_t)bsize + 1; mhdr[nm] = 12u + xlen; misz[nm] = rd32(raw + o + bsize + 1 - 4); mout[nm] = total; total += misz[nm]; nm++;
        o += (size_t)bsize + 1;
    }
    PHASE_SERIAL("member table");
    bf->data = xmalloc(total + 1);
    bf->len = total;
    if(g_ld_state == 0) ldeflate_init();
    {
        pthread_t *th = xmalloc(sizeof(pthread_t) * nt); inflate_job *job = xmalloc(sizeof(inflate_job) * nt);
        mcount = xmalloc((nm + 1) * sizeof(uint32_t));
        for(k = 0; k < nt; k++) { inflate_job j = {raw, moff, mout, mlen, misz, mhdr, nm, bf->data, k, nt, 0, NULL, 0, 0, mcount}; job[k] = j; if(k) pthread_create(&th[k], NULL, inflate_main, &job[k]); }
        inflate_main(&job[0]);
        for(k = 1; k < nt; k++) pthread_join(th[k], NULL);
        for(k = 0; k < nt; k++) bad |= job[k].bad;
        free(th); ijob = job;
    }
    if(mapped) munmap(raw, rawlen); else free(raw);
    free(moff); free(mlen); free(mhdr);
    PHASE("inflate");
    if(bad) return -2;
    /* header */
    if(bf->len < 12 || memcmp(bf->data, "BAM\1", 4)) return -3;
    o = 8 + rd32(bf->data + 4);
    bf->n_targets = (int32_t)rd32(bf->data + o); o += 4;
    bf->target_name = xmalloc(sizeof(char *) * bf->n_targets);
    bf->target_len = xmalloc(sizeof(uint32_t) * bf->n_targets);
    for(i = 0; i < (size_t)bf->n_targets; i++) {
        uint32_t ln = rd32(bf->data + o);
        bf->target_name[i] = (char *)(bf->data + o + 4);
        bf->target_len[i] = rd32(bf->data + o + 4 + ln);
        o += 8 + ln;
    }
    /* records: where they start, then the fields (range-parallel).  The inflating threads have walked every member that
     * starts on a record boundary; the member the header ends in is walked from the first record here, and a file whose
     * records do straddle members falls back to one serial walk over the block_size words. */
    cap = 1024; roff = xmalloc(cap * sizeof(size_t));
    {
        size_t im = 0, ocur = o, end_im, tot = 0; int ok = 1; size_t *cur = calloc((size_t)nt, sizeof(size_t));
        while(im + 1 < nm && mout[im + 1] <= o) im++;                    /* member holding the first record's first byte */
        end_im = mout[im] + misz[im];
        while(ocur + 4 <= end_im) {                                      /* rest of that member */
            uint32_t bs = rd32(bf->data + ocur);
            if(bs < 32 || ocur + 4 + (size_t)bs > end_im) break;
            if(bf->n_rec == cap) { cap *= 2; roff = xrealloc(roff, cap * sizeof(size_t)); }
            roff[bf->n_rec++] = ocur; ocur += 4 + (size_t)bs;
        }
        if(ocur != end_im && !(im + 1 == nm && ocur == bf->len)) ok = 0;
        for(i = 0; i <= im && i < nm; i++) if(mcount[i] != UINT32_MAX) cur[i % (size_t)nt] += mcount[i];      /* what the threads noted for the header members is not used */
        for(i = im + 1; i < nm && ok; i++) { if(mcount[i] == UINT32_MAX) ok = 0; else tot += mcount[i]; }
        if(ok) {
            size_t *first = xmalloc((nm + 1) * sizeof(size_t)), at = bf->n_rec; pthread_t *th = xmalloc(sizeof(pthread_t) * nt); gather_job *gj = xmalloc(sizeof(gather_job) * nt);
            if(bf->n_rec + tot + 1 > cap) { cap = bf->n_rec + tot + 1; roff = xrealloc(roff, cap * sizeof(size_t)); }
            for(i = 0; i < nm; i++) { first[i] = at; if(i > im) at += mcount[i]; }
            for(k = 0; k < nt; k++) { gather_job g = {roff, first, mcount, ijob, im + 1, nm, k, nt}; gj[k] = g; if(k) pthread_create(&th[k], NULL, gather_main, &gj[k]); }
            gather_main(&gj[0]);
            for(k = 1; k < nt; k++) pthread_join(th[k], NULL);
            bf->n_rec = at;
            free(first); free(th); free(gj);
        } else {
            bf->n_rec = 0;
            while(o + 4 <= bf->len) {
                uint32_t bs = rd32(bf->data + o);
                if(o + 4 + bs > bf->len || bs < 32) { free(roff); return -3; }
                if(bf->n_rec == cap) { cap *= 2; roff = xrealloc(roff, cap * sizeof(size_t)); }
                roff[bf->n_rec++] = o;
                o += 4 + (size_t)bs;
            }
        }
        free(cur);
        for(k = 0; k < nt; k++) free(ijob[k].roff);
        free(ijob); free(mcount); free(mout); free(misz);
    }
    PHASE("record walk");
    bf->rec = xmalloc((bf->n_rec + 1) * sizeof(brec));
    {
        pthread_t *th = xmalloc(sizeof(pthread_t) * nt); decode_job *job = xmalloc(sizeof(decode_job) * nt);
        for(k = 0; k < nt; k++) { decode_job j = {bf, roff, bf->n_rec * (size_t)k / nt, bf->n_rec * (size_t)(k + 1) / nt, 0, 0}; job[k] = j; if(k) pthread_create(&th[k], NULL, decode_main, &job[k]); }
        decode_main(&job[0]);
        for(k = 1; k < nt; k++) pthread_join(th[k], NULL);
        for(k = 0; k < nt; k++) { bad |= job[k].bad; if(job[k].max_rlen > bf->max_rlen) bf->max_rlen = job[k].max_rlen; }
        free(th); free(job);
    }
    free(roff);
    PHASE("record decode");
    if(bad) return -3;
    /* per-tid ranges; the pileup needs coordinate-sorted input (htslib errors out otherwise) */
    bf->tid_lo = xmalloc(sizeof(size_t) * (bf->n_targets + 1));
    (void)0;
    bf->tid_hi = xmalloc(sizeof(size_t) * (bf->n_targets + 1));
    for(i = 0; i < (size_t)bf->n_targets; i++) bf->tid_lo[i] = bf->tid_hi[i] = 0;
    {   /* every thread its stretch of the records, then the stretches are joined in order */
        pthread_t *th = xmalloc(sizeof(pthread_t) * nt); range_job *rj = xmalloc(sizeof(range_job) * nt); int32_t t;
        for(k = 0; k < nt; k++) { range_job j = {bf, bf->n_rec * (size_t)k / nt, bf->n_rec * (size_t)(k + 1) / nt, xmalloc(sizeof(size_t) * (bf->n_targets + 1)), xmalloc(sizeof(size_t) * (bf->n_targets + 1)), 0}; rj[k] = j; if(k) pthread_create(&th[k], NULL, range_main, &rj[k]); }
        range_main(&rj[0]);
        for(k = 1; k < nt; k++) pthread_join(th[k], NULL);
        for(k = 0; k < nt && !bad; k++) {
            if(rj[k].bad) { bad = rj[k].bad; break; }
            if(rj[k].lo < rj[k].hi && rj[k].lo > 0) { const brec *a = &bf->rec[rj[k].lo - 1], *b = &bf->rec[rj[k].lo]; if(a->tid == b->tid && a->tid >= 0 && a->tid < bf->n_targets && a->pos > b->pos) { bad = 2; break; } }
            for(t = 0; t < bf->n_targets; t++) {
                if(rj[k].tlo[t] == (size_t)-1) continue;
                if(bf->tid_hi[t] == 0 && bf->tid_lo[t] == 0) bf->tid_lo[t] = rj[k].tlo[t];
                else if(bf->tid_hi[t] != rj[k].tlo[t]) { bad = 1; break; }
                bf->tid_hi[t] = rj[k].thi[t];
            }
        }
        for(k = 0; k < nt; k++) { free(rj[k].tlo); free(rj[k].thi); }
        free(th); free(rj);
        if(bad == 1) { fprintf(stderr, "oracle: BAM is not coordinate sorted (contigs interleaved)\n"); return -4; }
        if(bad == 2) { fprintf(stderr, "oracle: BAM is not coordinate sorted\n"); return -4; }
    }
    PHASE("contig ranges");
    return 0;
}

/* region iterator == sam_itr_queryi(idx, tid, beg, end) + sam_itr_next (extract.c:379, common.c:413):
 * file order, same tid, pos < end, bam_endpos > beg. */
typedef struct { const bamfile *bf; int32_t tid, beg, end; size_t cur, hi; } regitr;
static void regitr_init(regitr *it, const bamfile *bf, int32_t tid, int32_t beg, int32_t end) {
    size_t lo = bf->tid_lo[tid], hi = bf->tid_hi[tid]; int64_t want = (int64_t)beg - bf->max_rlen - 1;
    it->bf = bf; it->tid = tid; it->beg = beg; it->end = end; it->hi = hi;
    /* first record with pos >= beg - max_rlen - 1 (binary search; sorted within tid) */
    { size_t a = lo, b = hi; while(a < b) { size_t m = a + (b - a) / 2; if((int64_t)bf->rec[m].pos < want) a = m + 1; else b = m; } it->cur = a; }
}
static const brec *regitr_next(regitr *it) {
    while(it->cur < it->hi) {
        const brec *r = &it->bf->rec[it->cur];
        if(r->pos >= it->end) { it->cur = it->hi; return NULL; }
        it->cur++;
        if(rec_endpos(r) > it->beg) return r;
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* FASTA (faidx semantics: name up to first whitespace, letters kept verbatim incl. case)       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int n; char **name; char **seq; int64_t *len; } fasta;
static int fasta_load(const char *fn, fasta *fa) {
    FILE *f = fopen(fn, "rb"); char *line = NULL; size_t cap = 0; ssize_t n; int64_t m = 0;
    memset(fa, 0, sizeof(*fa));
    if(!f) return -1;
    while((n = getline(&line, &cap, f)) >= 0) {
        if(line[0] == '>') {
            char *e = line + 1; while(*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') e++;
            *e = 0;
            fa->name = xrealloc(fa->name, sizeof(char *) * (fa->n + 1));
            fa->seq = xrealloc(fa->seq, sizeof(char *) * (fa->n + 1));
            fa->len = xrealloc(fa->len, sizeof(int64_t) * (fa->n + 1));
            fa->name[fa->n] = strdup(line + 1); fa->seq[fa->n] = NULL; fa->len[fa->n] = 0; fa->n++; m = 0;
        } else if(fa->n) {
            ssize_t i; int k = fa->n - 1;
            for(i = 0; i < n; i++) {
                unsigned char c = (unsigned char)line[i];
                if(c <= ' ' || c > '~') continue;                    /* isgraph() as in faidx */
                if(fa->len[k] + 1 > m) { m = m ? m * 2 : 1024; fa->seq[k] = xrealloc(fa->seq[k], m); }
                fa->seq[k][fa->len[k]++] = (char)c;
            }
        }
    }
    free(line); fclose(f);
    return 0;
}
typedef struct { const char *fn; fasta *fa; int rc; } fasta_job;
static void *fasta_bg_main(void *arg) { fasta_job *j = arg; j->rc = fasta_load(j->fn, j->fa); return NULL; }
/* faidx_fetch_seq(fai, name, beg, end_inclusive, &len): clamped to the contig; len=-2 unknown name */
static char *fetch_seq(const fasta *fa, const char *name, int64_t beg, int64_t end, int *len) {
    int i; int64_t L; char *s;
    for(i = 0; i < fa->n; i++) if(!strcmp(fa->name[i], name)) break;
    if(i == fa->n) { *len = -2; return NULL; }
    L = fa->len[i];
    if(end < beg) beg = end;
    if(beg < 0) beg = 0; else if(L <= beg) beg = L;
    if(end < 0) end = 0; else if(L <= end) end = L - 1;
    end += 1;                                       /* half-open now */
    if(end < beg) end = beg;
    s = xmalloc((size_t)(end - beg) + 1);
    memcpy(s, fa->seq[i] + beg, (size_t)(end - beg)); s[end - beg] = 0;
    *len = (int)(end - beg);
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* Config (MethylDackel.h:90-126)                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int keepCpG, keepCHG, keepCHH;
    int minMapq, minPhred, keepDupes, minDepth;
    int keepDiscordant, keepSingleton, ignoreFlags, requireFlags;
    int merge, methylKit, minOppositeDepth;
    int ignoreNH;
    double maxVariantFrac;
    int fraction, counts, logit;
    int cytosine_report;
    FILE *output_fp[3];
    char *reg;
    float minConversionEfficiency;
    char *BBMName;
    char **chromNames; uint32_t chromCount; uint32_t *chromLengths;
    char filterMappability;
    float mappabilityCutoff;
    int minMappableBases;
    char **bw_data;
    int bounds[16], absoluteBounds[16];
    int nThreads;
    unsigned long chunkSize;
    struct bedRegions_ *bed;
} Config;

/* per-chunk context == mplp_data (MethylDackel.h:139-149) */
typedef struct { Config *config; const bamfile *bf; regitr iter; int lseq; char *seq; uint32_t offset; int32_t bedIdx; } mplp_data;

/* working copy of a record: what htslib hands around as bam1_t (seq/qual are mutable) */
typedef struct {
    const brec *r;
    uint8_t *seq, *qual;       /* private copies */
} bam1;
#define seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)

/* ------------------------------------------------------------------------------------------ */
/* aux access: bam_aux_get returns a pointer to the TYPE byte (common.c:85-87 relies on it)     */
/* ------------------------------------------------------------------------------------------ */
static const uint8_t *aux_get(const brec *r, const char tag[2]) {
    const uint8_t *s = r->aux, *e = r->aux + r->aux_len;
    while(e - s >= 3) {
        const uint8_t *t = s + 2; uint8_t ty = *t; const uint8_t *v = t + 1; size_t sz;
        int hit = (s[0] == (uint8_t)tag[0] && s[1] == (uint8_t)tag[1]);
        switch(ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': { const uint8_t *z = memchr(v, 0, (size_t)(e - v)); if(!z) return NULL; sz = (size_t)(z - v) + 1; break; }
        case 'B': { uint32_t n; size_t es; if(e - v < 5) return NULL; n = rd32(v + 1);
                    switch(v[0]) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: return NULL; }
                    sz = 5 + es * (size_t)n; break; }
        default: return NULL;
        }
        if((size_t)(e - v) < sz) return NULL;
        if(hit) return t;
        s = v + sz;
    }
    return NULL;
}
static int64_t aux2i(const uint8_t *t) {   /* bam_aux2i: integer types only, else 0 */
    switch(*t) {
    case 'c': return (int8_t)t[1];
    case 'C': return t[1];
    case 's': return (int16_t)rd16(t + 1);
    case 'S': return rd16(t + 1);
    case 'i': return (int32_t)rd32(t + 1);
    case 'I': return rd32(t + 1);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* common.c restated                                                                            */
/* ------------------------------------------------------------------------------------------ */
/* ---------------------------------------------------------------- bed.c ---------------------------------------- */
typedef struct { int32_t tid, start, end; int8_t strand; } bedRegion;          /* MethylDackel.h:24-39 */
typedef struct bedRegions_ { bedRegion *region; int32_t n, m; } bedRegions;    /* MethylDackel.h:41-52 */

static int64_t compareRegions(int32_t tid0, int32_t start0, int32_t end0, int32_t tid1, int32_t start1, int32_t end1) {   /* bed.c:11-16 */
    if(tid0 != tid1) return ((int64_t)tid0) - ((int64_t)tid1);
    if(start0 < start1 && end0 >= start1) return 0;
    if(start0 >= start1 && start0 < end1) return 0;
    return ((int64_t)start0) - ((int64_t)start1);
}
static int spanOverlapsBED(int32_t tid, int32_t start, int32_t end, bedRegions *regs, int32_t *idx) {   /* bed.c:22-41 */
    bedRegion *reg = regs->region; int64_t rv = -1; int i;
    if(compareRegions(reg[*idx].tid, reg[*idx].start, reg[*idx].end - 1, tid, start, end) == 0) return 1;
    else {
        for(i = *idx; i < regs->n; i++) {
            rv = compareRegions(reg[i].tid, reg[i].start, reg[i].end - 1, tid, start, end);
            if(rv >= 0) { *idx = i; rv = (rv >= 1) ? 0 : 1; break; }
        }
        if(rv < 0) rv = -1;
    }
    return (int)rv;
}
static int posOverlapsBED(int32_t tid, int32_t pos, bedRegions *regions, int32_t idx) {   /* bed.c:46-53 */
    if(idx >= regions->n) return 0;
    if(tid != regions->region[idx].tid) return (regions->region[idx].tid < tid) ? -1 : 0;
    if(pos >= regions->region[idx].end) return -1;
    if(pos < regions->region[idx].start) return 0;
    return 1;
}
static int getStrand(const brec *b);
static int readStrandOverlapsBED(const brec *b, bedRegion region) {   /* bed.c:56-64 */
    int s = getStrand(b);
    if(region.strand) {
        if(region.strand == 1 && (s == 1 || s == 3)) return 1;
        if(region.strand == 2 && (s == 2 || s == 4)) return 1;
        return 0;
    }
    return 1;
}
static int sortBED_func(const void *a, const void *b) {   /* bed.c:66-80 */
    const bedRegion *pa = a, *pb = b;
    if(pa->tid < pb->tid) return -1;
    if(pa->tid > pb->tid) return 1;
    if(pa->start < pb->start) return -1;
    if(pa->start > pb->start) return 1;
    if(pa->end < pb->end) return -1;
    if(pa->end > pb->end) return 1;
    if(pa->strand < pb->strand) return -1;
    if(pa->strand > pb->strand) return 1;
    return 0;
}
/* parseBED (bed.c:90-237).  The reference reads lines through htslib's kstream (gz or plain, '\n'-separated, one
 * trailing '\r' dropped when the line is longer than one character) and stops at the first EMPTY line because its
 * loop condition is "length > 0".  A line that ends before its start or end column makes the reference read stale
 * bytes of the kstring buffer (undefined); here the line buffer is followed by NULs, so such a line is "malformed". */
static bedRegions *parseBED(const char *fn, const bamfile *hdr, int keepStrand) {
    gzFile fp; char *data = NULL; size_t n = 0, m = 0, o = 0; int32_t lnum = 0, i; bedRegions *regions;
    char *line = NULL;
    if((fp = gzopen(fn, "r")) == NULL) { fprintf(stderr, "Couldn't open %s for reading.\n", fn); return NULL; }
    for(;;) {
        int got;
        if(m - n < 65536) { m = m ? 2 * m : 1 << 20; data = xrealloc(data, m); }
        got = gzread(fp, data + n, 65536);
        if(got <= 0) break;
        n += (size_t)got;
    }
    gzclose(fp);
    regions = xmalloc(sizeof(*regions)); regions->n = 0; regions->m = 1000; regions->region = xmalloc(sizeof(bedRegion) * (size_t)regions->m);
    while(o < n) {      /* ks_getuntil(ks, KS_SEP_LINE, ...) > 0 */
        size_t e = o, l; char *p1, *p2; bedRegion *r;
        while(e < n && data[e] != '\n') e++;
        l = e - o;
        if(l > 1 && data[e - 1] == '\r') l--;
        line = xrealloc(line, l + 4); memcpy(line, data + o, l); memset(line + l, 0, 4);
        o = e + 1;
        if(l == 0) break;
        lnum++;
        p1 = line; p2 = p1;
        if(*p1 == '\0') continue;
        if(*p1 == '#') continue;
        if(regions->m - regions->n < 100) { regions->m += 1000; regions->region = xrealloc(regions->region, sizeof(bedRegion) * (size_t)regions->m); }
        r = &regions->region[regions->n];
        r->tid = -1; r->start = -1; r->end = -1; r->strand = 0;
        while(*p2 && !isspace((unsigned char)*p2)) p2++;
        if(*p2 != '\0') *p2 = '\0';
        for(i = 0; i < hdr->n_targets; i++) if(strcmp(p1, hdr->target_name[i]) == 0) { r->tid = i; break; }
        if(r->tid == -1) { if(strcmp(p1, "track") == 0) continue; if(strcmp(p1, "browser") == 0) continue; }
        if(r->tid == -1) { fprintf(stderr, "Couldn't properly parse line number %i in %s.\n", lnum, fn); goto err; }
        p1 = p2 + 1;
        if(sscanf(p1, "%" SCNd32, &r->start) != 1 || r->start == -1) { fprintf(stderr, "Line %" PRId32 " of %s is malformed.\n", lnum, fn); goto err; }
        p2++;
        while(*p2 && !isspace((unsigned char)*p2)) p2++;
        if(*p2 != '\0') *p2 = '\0';
        p1 = p2 + 1;
        if(sscanf(p1, "%" SCNd32, &r->end) != 1 || r->end == -1) { fprintf(stderr, "Line %" PRId32 " of %s is malformed.\n", lnum, fn); goto err; }
        if(r->start >= r->end) { fprintf(stderr, "The position on line %" PRId32 " of %s is incorrect (%" PRId32 " >= %" PRId32 ".\n", lnum, fn, r->start, r->end); goto err; }
        if(r->start < 0) r->start = 0;
        if(r->end > (int64_t)hdr->target_len[r->tid] + 1) r->end = (int32_t)(hdr->target_len[r->tid] + 1);
        regions->n++;
        if((size_t)(p2 - line) >= l) continue;
        p2++;
        if(keepStrand != 1) continue;
        while(*p2 && !isspace((unsigned char)*p2)) p2++;
        while(*p2 && isspace((unsigned char)*p2)) p2++;          /* 4th column */
        if(*p2 == '\0') continue;
        while(*p2 && !isspace((unsigned char)*p2)) p2++;
        if(*p2 == '\0') continue;
        while(*p2 && isspace((unsigned char)*p2)) p2++;          /* 5th column */
        if(*p2 == '\0') continue;
        while(*p2 && !isspace((unsigned char)*p2)) p2++;
        if(*p2 == '\0') continue;
        while(*p2 && isspace((unsigned char)*p2)) p2++;          /* strand */
        if(*p2 == '\0') continue;
        if(*p2 == '+') regions->region[regions->n - 1].strand = 1;
        else if(*p2 == '-') regions->region[regions->n - 1].strand = 2;
    }
    free(line); free(data);
    qsort(regions->region, (size_t)regions->n, sizeof(bedRegion), sortBED_func);
    fprintf(stderr, "Parsed %" PRId32 " regions in %s\n", regions->n, fn);
    return regions;
err:
    free(line); free(data); free(regions->region); free(regions);
    return NULL;
}

static int isCpG(char *seq, int pos, int seqlen) {            /* common.c:49-61 */
    if(pos >= seqlen) return 0;
    if(seq[pos] == 'C' || seq[pos] == 'c') {
        if(pos + 1 == seqlen) return 0;
        if(seq[pos + 1] == 'G' || seq[pos + 1] == 'g') return 1;
        return 0;
    } else if(seq[pos] == 'G' || seq[pos] == 'g') {
        if(pos == 0) return 0;
        if(seq[pos - 1] == 'C' || seq[pos - 1] == 'c') return -1;
        return 0;
    }
    return 0;
}
static int isCHG(char *seq, int pos, int seqlen) {            /* common.c:63-75 */
    if(pos >= seqlen) return 0;
    if(seq[pos] == 'C' || seq[pos] == 'c') {
        if(pos + 2 >= seqlen) return 0;
        if(seq[pos + 2] == 'G' || seq[pos + 2] == 'g') return 1;
        return 0;
    } else if(seq[pos] == 'G' || seq[pos] == 'g') {
        if(pos <= 1) return 0;
        if(seq[pos - 2] == 'C' || seq[pos - 2] == 'c') return -1;
        return 0;
    }
    return 0;
}
static int isCHH(char *seq, int pos, int seqlen) {            /* common.c:77-82 */
    if(pos >= seqlen) return 0;
    if(seq[pos] == 'C' || seq[pos] == 'c') return 1;
    else if(seq[pos] == 'G' || seq[pos] == 'g') return -1;
    return 0;
}

static int getStrand(const brec *b) {                          /* common.c:84-116 */
    const uint8_t *XG = aux_get(b, "XG");
    if(XG != NULL && XG[1] != 'C' && XG[1] != 'G') XG = NULL;
    if(XG == NULL) {
        if(b->flag & 0x1) {
            if((b->flag & 0x50) == 0x50) return 2;
            else if(b->flag & 0x40) return 1;
            else if((b->flag & 0x90) == 0x90) return 1;
            else if(b->flag & 0x80) return 2;
            return 0;
        } else {
            if(b->flag & 0x10) return 2;
            return 1;
        }
    } else {
        if(XG[1] == 'C') {
            if((b->flag & 0x51) == 0x41) return 1;
            else if((b->flag & 0x51) == 0x51) return 3;
            else if((b->flag & 0x91) == 0x81) return 3;
            else if((b->flag & 0x91) == 0x91) return 1;
            else if(b->flag & 0x10) return 3;
            else return 1;
        } else {
            if((b->flag & 0x51) == 0x41) return 4;
            else if((b->flag & 0x51) == 0x51) return 2;
            else if((b->flag & 0x91) == 0x81) return 2;
            else if((b->flag & 0x91) == 0x91) return 4;
            else if(b->flag & 0x10) return 2;
            else return 4;
        }
    }
}

/* common.c:118-134 (and its twin getMethylState, 338-354) */
static int methState(Config *config, const bam1 *b, int qpos) {
    uint8_t base;
    int strand = getStrand(b->r);
    if(strand == 0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
    /* a record whose CIGAR consumes more query bases than it stores (SEQ '*', l_qseq 0) makes the reference read past its
     * sequence and quality arrays; such a base is given no letter and quality 0 here */
    if(qpos >= b->r->l_qseq) return 0;
    base = seqi(b->seq, qpos);
    if(b->qual[qpos] < config->minPhred) return 0;
    if(base == 2 && (strand == 1 || strand == 3)) return 1;
    else if(base == 8 && (strand == 1 || strand == 3)) return -1;
    else if(base == 4 && (strand == 2 || strand == 4)) return 1;
    else if(base == 1 && (strand == 2 || strand == 4)) return -1;
    return 0;
}

static void maskBase(bam1 *b, int i) { b->qual[i] = 0; if(i & 1) b->seq[i >> 1] |= 0xf; else b->seq[i >> 1] |= 0xf0; }

static void trimAlignment(bam1 *b, int bounds[16]) {           /* common.c:137-172 */
    int strand = getStrand(b->r) - 1, i, lb, rb, l = b->r->l_qseq;
    if(strand < 0) return;   /* reference indexes bounds[-4..] here (UB); such reads abort later anyway */
    if(b->r->flag & 0x80) { lb = bounds[4 * strand + 2]; rb = bounds[4 * strand + 3]; }
    else { lb = bounds[4 * strand]; rb = bounds[4 * strand + 1]; }
    lb = (lb < l) ? lb : l;
    if(lb) for(i = 0; i < lb; i++) maskBase(b, i);
    if(rb) for(i = rb; i < l; i++) maskBase(b, i);
}
static void trimAbsoluteAlignment(bam1 *b, int bounds[16]) {   /* common.c:174-208 */
    int strand = getStrand(b->r) - 1, i, lb, rb, l = b->r->l_qseq;
    if(strand < 0) return;
    if(b->r->flag & 0x80) { lb = bounds[4 * strand + 2]; rb = bounds[4 * strand + 3]; }
    else { lb = bounds[4 * strand]; rb = bounds[4 * strand + 1]; }
    if(g_pt) {                                            /* differential search only; never taken by a parity test */
        if(g_pt == PT_ABS_SWAP_R1R2) { if(b->r->flag & 0x80) { lb = bounds[4 * strand]; rb = bounds[4 * strand + 1]; } else { lb = bounds[4 * strand + 2]; rb = bounds[4 * strand + 3]; } }
        if(g_pt == PT_ABS_SKIP_READ1 && !(b->r->flag & 0x80)) return;
        if(g_pt == PT_ABS_SKIP_READ2 && (b->r->flag & 0x80)) return;
        if(g_pt == PT_ABS_RIGHT_M1 && rb > 0) rb--;
        if(g_pt == PT_ABS_LEFT_M1 && lb > 0) lb--;
        if(g_pt == PT_ABS_RIGHT_P1 && rb > 0) rb++;
        if(g_pt == PT_ABS_LEFT_P1 && lb > 0) lb++;
        if(g_pt == PT_ABS_5PRIME && (b->r->flag & 0x10)) { int t = lb; lb = rb; rb = t; }
        if(g_pt == PT_ABS_AS_RELATIVE) { lb = (lb < l) ? lb : l; for(i = 0; i < lb; i++) maskBase(b, i); if(rb) for(i = rb; i < l; i++) maskBase(b, i); return; }
        if(g_pt == PT_ABS_QUAL_ONLY || g_pt == PT_ABS_N_ONLY) {
            lb = (lb < l) ? lb : l; rb = (rb < l) ? rb : l;
            for(i = 0; i < l; i++) if(i < lb || i >= l - rb) { if(g_pt == PT_ABS_QUAL_ONLY) b->qual[i] = 0; else { if(i & 1) b->seq[i >> 1] |= 0xf; else b->seq[i >> 1] |= 0xf0; } }
            return;
        }
    }
    lb = (lb < l) ? lb : l;
    rb = (rb < l) ? rb : l;
    if(lb) for(i = 0; i < lb; i++) maskBase(b, i);
    if(rb) for(i = 0; i < rb; i++) maskBase(b, l - 1 - i);
}

/* common.c:210-275 */
static unsigned char *getMappabilityValue(Config *config, const char *chrom_n, uint32_t start, uint32_t end) {
    char chromFound = 0; uint32_t chrom = (uint32_t)-1; int i;
    unsigned char *data; int index, offset, arrlen = 0;
    for(i = 0; i < (int)config->chromCount; i++) if(!strcmp(config->chromNames[i], chrom_n)) { chrom = i; chromFound = 1; break; }
    data = xmalloc((size_t)(end - start));
    index = (int)(start / 8); offset = (int)(start % 8);
    if(chromFound) { arrlen = config->chromLengths[chrom] / 8; if(config->chromLengths[chrom] % 8 > 0) arrlen++; }
    for(i = 0; i < (int)(end - start); i++) {
        unsigned char byte, mask;
        if(chromFound) { if(index >= arrlen) byte = 0; else byte = (unsigned char)config->bw_data[chrom][index]; }
        else byte = 0;
        mask = (unsigned char)(1 << offset);
        data[i] = (unsigned char)((byte & mask) >> offset);
        if(offset == 7) { index++; offset = 0; } else offset++;
    }
    return data;
}
/* common.c:277-335 */
static char check_mappability(mplp_data *ldata, const brec *b) {
    int read1_start, read1_end, read2_start, read2_end, i, num_mappable_reads = 0;
    signed char num_mappable_bases = 0;       /* `char` on x86-64 */
    unsigned char *vals;
    if((b->flag & 0x40) || ((b->flag & 0x10) && (b->flag & 0x80))) {
        read1_start = b->pos; read1_end = b->pos + b->l_qseq;
        read2_start = b->mpos; read2_end = b->mpos + b->l_qseq;
    } else {
        read2_start = b->pos; read2_end = b->pos + b->l_qseq;
        read1_start = b->mpos; read1_end = b->mpos + b->l_qseq;
    }
    vals = getMappabilityValue(ldata->config, ldata->bf->target_name[b->tid], (uint32_t)read1_start, (uint32_t)read1_end);
    for(i = 0; i < read1_end - read1_start; i++) {
        if(vals[i] > 0) num_mappable_bases = (signed char)(num_mappable_bases + 1);
        if(num_mappable_bases >= ldata->config->minMappableBases) { num_mappable_reads++; break; }
    }
    free(vals);
    vals = getMappabilityValue(ldata->config, ldata->bf->target_name[b->tid], (uint32_t)read2_start, (uint32_t)read2_end);
    num_mappable_bases = 0;
    for(i = 0; i < read2_end - read2_start; i++) {
        if(vals[i] > 0) num_mappable_bases = (signed char)(num_mappable_bases + 1);
        if(num_mappable_bases >= ldata->config->minMappableBases) { num_mappable_reads++; break; }
    }
    free(vals);
    return (char)num_mappable_reads;
}

static float computeEfficiency(unsigned int nMethyl, unsigned int nUMethyl) {   /* common.c:356-359 */
    if(nMethyl + nUMethyl == 0) return 1.0;
    return nUMethyl / ((float)(nMethyl + nUMethyl));
}
/* common.c:361-404.  NB the reference reads ldata->seq at a negative index when the read starts
 * left of the chunk window (undefined behaviour); here such positions are treated as "no context". */
static float computeConversionEfficiency(bam1 *b, mplp_data *ldata) {
    unsigned int nMethyl = 0, nUMethyl = 0;
    uint32_t i, j, seqEnd = ldata->offset + ldata->lseq, op, opLen;
    int state, pos = b->r->pos, seqPos = 0;
    for(i = 0; i < b->r->n_cigar; i++) {
        op = cig_op(b->r->cigar, i); opLen = cig_len(b->r->cigar, i);
        switch(op) {
        case 0: case 7: case 8:
            for(j = 0; j < opLen; j++, seqPos++) {
                int64_t wi = (int64_t)pos + j - ldata->offset;
                if((uint32_t)(pos + j) >= seqEnd) return computeEfficiency(nMethyl, nUMethyl);
                if(wi < 0) continue;
                if(isCpG(ldata->seq, (int)wi, ldata->lseq)) continue;
                else if(isCHG(ldata->seq, (int)wi, ldata->lseq) || isCHH(ldata->seq, (int)wi, ldata->lseq)) {
                    state = methState(ldata->config, b, seqPos);
                    if(state > 0) nMethyl++; else if(state < 0) nUMethyl++;
                }
            }
            break;
        case 1: case 4: seqPos += opLen; break;
        case 2: case 3: pos += opLen; break;
        }
    }
    return computeEfficiency(nMethyl, nUMethyl);
}

/* filter_func (common.c:407-463): pulls the next ADMITTED record of the region into *b.
 * returns 0 on success, -1 at end.  b->seq/b->qual are (re)filled private copies. */
static int filter_func(mplp_data *ldata, bam1 *b) {
    const brec *r; Config *c = ldata->config;
    while(1) {
        r = regitr_next(&ldata->iter);
        if(!r) return -1;
        if(r->tid == -1 || (r->flag & 0x4)) continue;
        if(r->mapq < c->minMapq) continue;
        if((r->flag & c->ignoreFlags) && !(g_pt == PT_ADMIT_QCFAIL && !(r->flag & c->ignoreFlags & ~0x200))) continue;
        if(c->requireFlags && (r->flag & c->requireFlags) != c->requireFlags) continue;
        if(!c->keepDupes && (r->flag & 0x400)) continue;
        if(!c->ignoreNH) {
            const uint8_t *p = aux_get(r, "NH");
            if(p != NULL) { int NH = (int)aux2i(p); if(NH > 1) continue; }
        }
        if(c->filterMappability && check_mappability(ldata, r) == 0) continue;
        if(!c->keepSingleton && (r->flag & 0x9) == 0x9) continue;
        if(!c->keepDiscordant && (r->flag & 0x3) == 0x1) continue;
        /* common.c:431 sets 0x2 on the private copy; nothing downstream in extract looks at it */
        if(c->bed) {       /* common.c:432-439: prefilter on the read's span, strand independent */
            int overlap = spanOverlapsBED(r->tid, r->pos, rec_endpos(r), c->bed, &ldata->bedIdx);
            if(overlap == 0) continue;
            if(overlap < 0) return -1;
        }
        b->r = r;
        b->seq = xrealloc(b->seq, (size_t)(r->l_qseq + 1) / 2 + 1);
        b->qual = xrealloc(b->qual, (size_t)r->l_qseq + 1);
        memcpy(b->seq, r->seq, (size_t)(r->l_qseq + 1) / 2);
        memcpy(b->qual, r->qual, (size_t)r->l_qseq);
        if(c->minConversionEfficiency > 0.0) {
            if(computeConversionEfficiency(b, ldata) < c->minConversionEfficiency) continue;
        }
        trimAlignment(b, c->bounds);
        if(g_pt != PT_TRIM_AFTER_PAIRING || !(r->flag & 0x1)) trimAbsoluteAlignment(b, c->absoluteBounds);
        else memcpy(g_pt_abs, c->absoluteBounds, sizeof(g_pt_abs));      /* diagnostic: paired reads are trimmed after the overlap rule instead */
        return 0;
    }
}

/* adjustBounds (common.c:466-493) */
static void adjustBounds(const bamfile *bf, const fasta *fa, uint32_t *localTid, uint32_t *localPos, uint32_t *localEnd) {
    uint32_t start, end, tmp; int seqlen; char *seq;
    end = *localEnd + 1;
    if(*localEnd > 0) start = *localEnd - 1; else start = 0;
    seq = fetch_seq(fa, bf->target_name[*localTid], (int)start, (int)end, &seqlen);
    if(seqlen > 1) {
        if(seqlen > 2 && (seq[0] & 0x5F) == 'C' && (seq[2] & 0x5F) == 'G') *localEnd += 2;
        else if((seq[1] & 0x5F) == 'G') *localEnd += 1;
    }
    free(seq);
    if(*localPos > *localEnd) { tmp = *localPos; *localPos = *localEnd; *localEnd = tmp; }
}

/* ------------------------------------------------------------------------------------------ */
/* overlaps.c restated.  The qname dictionary is a tiny chained hash keyed on the qname string  */
/* ------------------------------------------------------------------------------------------ */
struct lbnode;
typedef struct oent { const char *key; struct lbnode *val; struct oent *next; } oent;
typedef struct { oent **b; size_t nb; } ohash_t;
static size_t ohash_fn(const char *s) { size_t h = 1469598103934665603ULL; for(; *s; s++) h = (h ^ (uint8_t)*s) * 1099511628211ULL; return h; }
static ohash_t *initOlapHash(void) { ohash_t *h = xmalloc(sizeof(*h)); h->nb = 1 << 16; h->b = calloc(h->nb, sizeof(oent *)); return h; }
static void destroyOlapHash(ohash_t *h) { size_t i; for(i = 0; i < h->nb; i++) { oent *e = h->b[i]; while(e) { oent *n = e->next; free(e); e = n; } } free(h->b); free(h); }
static oent **ohash_find(ohash_t *h, const char *k) { oent **p = &h->b[ohash_fn(k) & (h->nb - 1)]; while(*p && strcmp((*p)->key, k)) p = &(*p)->next; return p; }

typedef struct lbnode {       /* one buffered read of the pileup (htslib lbnode_t) */
    bam1 b; int32_t beg, end; struct lbnode *next;
} lbnode;

static int32_t *calculate_positions(const brec *read) {        /* overlaps.c:27-52 */
    int32_t *positions = xmalloc(sizeof(int32_t) * (size_t)(read->l_qseq + 1));
    int i, j, offset = 0, op, op_len; int32_t previous_position = read->pos;
    for(i = 0; i < read->n_cigar; i++) {
        op = cig_op(read->cigar, i); op_len = cig_len(read->cigar, i);
        for(j = 0; j < op_len; j++) {
            if(op == 0 || op == 7 || op == 8) { if(offset < read->l_qseq) positions[offset] = previous_position; previous_position++; offset++; }
            else if(op == 1 || op == 4) { if(offset < read->l_qseq) positions[offset] = -1; offset++; }
            else if(op == 2 || op == 3) previous_position++;
            else if(op == 5) { }
            else fprintf(stderr, "[calculate_positions] We encountered a CIGAR operation that we're not ready to deal with in %s\n", read->qname);
        }
    }
    /* a CIGAR that covers fewer query bases than l_qseq leaves the tail uninitialised in the
     * reference; treat as unaligned */
    for(; offset < read->l_qseq; offset++) positions[offset] = -1;
    return positions;
}

static void cust_tweak_overlap_quality(bam1 *a, bam1 *b) {     /* overlaps.c:54-119 */
    int ia = 0, ib = 0; int32_t na = a->r->l_qseq, nb = b->r->l_qseq;
    int32_t *posa = calculate_positions(a->r), *posb = calculate_positions(b->r);
    uint8_t *a_qual = a->qual, *b_qual = b->qual, *a_seq = a->seq, *b_seq = b->seq;
    int sa = getStrand(a->r), sb = getStrand(b->r);
    if(((sa - sb) & 1) == 1) goto quit;
    if(g_pt == PT_NO_OVERLAP) goto quit;
    while(ia < na && posa[ia] < 0) ia++;
    while(ib < nb && posb[ib] < 0) ib++;
    if(ia == na || ib == nb) goto quit;
    if(posa[ia] < posb[ib]) { while(ia < na && posa[ia] < posb[ib]) ia++; }
    else { while(ib < nb && posb[ib] < posa[ia]) ib++; }
    if(ia == na || ib == nb) goto quit;
    while(ia < na && ib < nb) {
        if(posa[ia] < posb[ib] || posa[ia] < 0) { ia++; continue; }
        if(posb[ib] < posa[ia] || posb[ib] < 0) { ib++; continue; }
        if(g_pt == PT_SKIP_N_IN_OVERLAP && (seqi(a_seq, ia) == 15 || seqi(b_seq, ib) == 15)) { ia++; ib++; continue; }
        if(g_pt == PT_TIE_FIRST && seqi(a_seq, ia) == seqi(b_seq, ib) && a_qual[ia] == b_qual[ib]) { a_qual[ia] = (uint8_t)(int)(a_qual[ia] + 0.2 * a_qual[ia]); b_qual[ib] = 0; ia++; ib++; continue; }
        if(seqi(a_seq, ia) != seqi(b_seq, ib)) {
            if(a_qual[ia] > b_qual[ib] && seqi(a_seq, ia) != 15) { a_qual[ia] -= b_qual[ib]; b_qual[ib] = 0; }
            else if(b_qual[ib] > a_qual[ia] && seqi(b_seq, ib) != 15) { b_qual[ib] -= a_qual[ia]; a_qual[ia] = 0; }
            else { a_qual[ia] = 0; b_qual[ib] = 0; }
        } else {
            /* `a_qual[ia] += 0.2*a_qual[ia]` : uint8 <- double; >=256 wraps mod 256 with gcc/x86-64 */
            if(a_qual[ia] > b_qual[ib]) { a_qual[ia] = (uint8_t)(int)(a_qual[ia] + 0.2 * a_qual[ia]); b_qual[ib] = 0; }
            else { b_qual[ib] = (uint8_t)(int)(b_qual[ib] + 0.2 * b_qual[ib]); a_qual[ia] = 0; }
        }
        ia++; ib++;
    }
quit:
    free(posa); free(posb);
}

static void custom_overlap_constructor(ohash_t *oh, lbnode *nb) {   /* overlaps.c:121-139 */
    oent **p = ohash_find(oh, nb->b.r->qname);
    if(!(nb->b.r->flag & 0x1) || ((nb->b.r->flag & 12) > 0)) return;
    if(*p == NULL) { oent *e = xmalloc(sizeof(*e)); e->key = nb->b.r->qname; e->val = nb; e->next = NULL; *p = e; }
    else {
        oent *e = *p;
        if(g_pt == PT_SWAP_A_B) cust_tweak_overlap_quality(&nb->b, &e->val->b); else cust_tweak_overlap_quality(&e->val->b, &nb->b);
        if(g_pt == PT_TRIM_AFTER_PAIRING) { int sv = g_pt; g_pt = 0; trimAbsoluteAlignment(&e->val->b, g_pt_abs); trimAbsoluteAlignment(&nb->b, g_pt_abs); g_pt = sv; }
        *p = e->next; free(e);
    }
}
static void custom_overlap_destructor(ohash_t *oh, lbnode *nb) {    /* overlaps.c:141-147 */
    oent **p = ohash_find(oh, nb->b.r->qname);
    if(*p) { oent *e = *p; *p = e->next; free(e); }
}

/* ------------------------------------------------------------------------------------------ */
/* htslib pileup buffer restated (bam_plp_push / bam_plp64_next / bam_plp64_auto, one input,    */
/* maxcnt = INT_MAX, no htslib-side overlap detection: extract.c:394-399).                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const bam1 *b; int32_t qpos; int is_del, is_refskip; } pileup1;
typedef struct {
    mplp_data *data; ohash_t *oh;
    lbnode *head, *tail;           /* tail is the spare node, as in htslib */
    int32_t tid, pos, max_tid, max_pos; int is_eof;
    pileup1 *plp; int max_plp;
    bam1 scratch;
} plp_t;

static lbnode *node_new(void) { lbnode *n = calloc(1, sizeof(*n)); if(!n) exit(2); return n; }
static void node_free(lbnode *n) { free(n->b.seq); free(n->b.qual); free(n); }

static void plp_init(plp_t *it, mplp_data *data, ohash_t *oh) {
    memset(it, 0, sizeof(*it)); it->data = data; it->oh = oh;
    it->head = it->tail = node_new(); it->max_tid = it->max_pos = -1;
}
static void plp_destroy(plp_t *it) {
    lbnode *p = it->head;
    while(p != it->tail) { lbnode *n = p->next; if(it->oh) custom_overlap_destructor(it->oh, p); node_free(p); p = n; }
    node_free(it->tail); free(it->plp); free(it->scratch.seq); free(it->scratch.qual);
}
/* resolve_cigar2: where does column `pos` fall in this read? */
static void resolve_cigar(pileup1 *p, const brec *r, int32_t pos) {
    int32_t x = r->pos, y = 0; int k;
    p->qpos = 0; p->is_del = p->is_refskip = 0;
    for(k = 0; k < r->n_cigar; k++) {
        int op = cig_op(r->cigar, k); int32_t l = (int32_t)cig_len(r->cigar, k);
        if(op == 0 || op == 7 || op == 8) { if(pos < x + l) { p->qpos = y + (pos - x); return; } x += l; y += l; }
        else if(op == 2 || op == 3) { if(pos < x + l) { p->qpos = y; p->is_del = 1; p->is_refskip = (op == 3); return; } x += l; }
        else if(op == 1 || op == 4) y += l;
    }
    p->is_del = 1;   /* unreachable for beg <= pos < end */
}
static void plp_push(plp_t *it, const bam1 *b) {
    if(b) {
        lbnode *t = it->tail;
        /* bam_copy1 */
        t->b.r = b->r;
        t->b.seq = xrealloc(t->b.seq, (size_t)(b->r->l_qseq + 1) / 2 + 1);
        t->b.qual = xrealloc(t->b.qual, (size_t)b->r->l_qseq + 1);
        memcpy(t->b.seq, b->seq, (size_t)(b->r->l_qseq + 1) / 2);
        memcpy(t->b.qual, b->qual, (size_t)b->r->l_qseq);
        t->beg = b->r->pos; t->end = b->r->pos + b->r->rlen;      /* raw rlen, not bam_endpos */
        it->max_tid = b->r->tid; it->max_pos = t->beg;
        if(t->end > it->pos || b->r->tid > it->tid) {
            lbnode *next = node_new();
            if(it->oh) custom_overlap_constructor(it->oh, t);     /* mbias installs no constructor/destructor (MBias.c:158-161) */
            t->next = next; it->tail = next;
        }
    } else it->is_eof = 1;
}
static pileup1 *plp_next(plp_t *it, int *_tid, int32_t *_pos, int *_n_plp) {
    *_n_plp = 0;
    if(it->is_eof && it->head == it->tail) return NULL;
    while(it->is_eof || it->max_tid > it->tid || (it->max_tid == it->tid && it->max_pos > it->pos)) {
        int n_plp = 0; lbnode **pptr = &it->head;
        while(*pptr != it->tail) {
            lbnode *p = *pptr;
            if(p->b.r->tid < it->tid || (p->b.r->tid == it->tid && p->end <= it->pos)) {
                if(it->oh) custom_overlap_destructor(it->oh, p);
                *pptr = p->next; node_free(p);
            } else {
                if(p->b.r->tid == it->tid && p->beg <= it->pos) {
                    if(n_plp == it->max_plp) { it->max_plp = it->max_plp ? it->max_plp << 1 : 256; it->plp = xrealloc(it->plp, sizeof(pileup1) * it->max_plp); }
                    it->plp[n_plp].b = &p->b;
                    resolve_cigar(&it->plp[n_plp], p->b.r, it->pos);
                    n_plp++;
                }
                pptr = &(*pptr)->next;
            }
        }
        *_n_plp = n_plp; *_tid = it->tid; *_pos = it->pos;
        if(it->head != it->tail && it->tid < it->head->b.r->tid) { it->tid = it->head->b.r->tid; it->pos = it->head->beg; }
        else if(it->head != it->tail && it->pos < it->head->beg) it->pos = it->head->beg;
        else ++it->pos;
        if(n_plp) return it->plp;
        if(it->is_eof && it->head == it->tail) break;
    }
    return NULL;
}
static pileup1 *plp_auto(plp_t *it, int *_tid, int32_t *_pos, int *_n_plp) {
    pileup1 *plp;
    if((plp = plp_next(it, _tid, _pos, _n_plp)) != NULL) return plp;
    *_n_plp = 0;
    if(it->is_eof) return NULL;
    while(filter_func(it->data, &it->scratch) >= 0) {
        plp_push(it, &it->scratch);
        if((plp = plp_next(it, _tid, _pos, _n_plp)) != NULL) return plp;
    }
    plp_push(it, NULL);
    if((plp = plp_next(it, _tid, _pos, _n_plp)) != NULL) return plp;
    return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* extract.c restated                                                                           */
/* ------------------------------------------------------------------------------------------ */
static double logit(double p) { return log(p) - log(1 - p); }
struct lastCall { int32_t tid, pos; uint32_t nmethyl, nunmethyl; };
static const char *TriNucleotideContexts[25] = {"CAA", "CAC", "CAG", "CAT", "CAN", "CCA", "CCC", "CCG", "CCT", "CCN",
    "CGA", "CGC", "CGG", "CGT", "CGN", "CTA", "CTC", "CTG", "CTT", "CTN", "CNA", "CNC", "CNG", "CNT", "CNN"};

static void writeCall(kstr *ks, Config *config, const char *chrom, int32_t pos, int32_t width, uint32_t nmethyl, uint32_t nunmethyl, char base, const char *context, const char *tnc) {
    char str[10000]; char strand = (base == 'C' || base == 'c') ? 'F' : 'R';
    if((int64_t)(uint32_t)(nmethyl + nunmethyl) < config->minDepth && !config->cytosine_report) return;
    if(!config->fraction && !config->logit && !config->counts && !config->methylKit && !config->cytosine_report) {
        snprintf(str, 10000, "%s\t%i\t%i\t%i\t%" PRIu32 "\t%" PRIu32 "\n", chrom, pos, pos + width, (int)(100.0 * ((double)nmethyl) / (nmethyl + nunmethyl)), nmethyl, nunmethyl);
    } else if(config->fraction) {
        snprintf(str, 10000, "%s\t%i\t%i\t%f\n", chrom, pos, pos + width, ((double)nmethyl) / (nmethyl + nunmethyl));
    } else if(config->counts) {
        snprintf(str, 10000, "%s\t%i\t%i\t%i\n", chrom, pos, pos + width, nmethyl + nunmethyl);
    } else if(config->logit) {
        snprintf(str, 10000, "%s\t%i\t%i\t%f\n", chrom, pos, pos + width, logit(((double)nmethyl) / (nmethyl + nunmethyl)));
    } else if(config->methylKit) {
        snprintf(str, 10000, "%s.%i\t%s\t%i\t%c\t%i\t%6.2f\t%6.2f\n", chrom, pos + 1, chrom, pos + 1, strand, nmethyl + nunmethyl,
                 100.0 * ((double)nmethyl) / (nmethyl + nunmethyl), 100.0 * ((double)nunmethyl) / (nmethyl + nunmethyl));
    } else {
        strand = (base == 'C' || base == 'c') ? '+' : '-';
        snprintf(str, 10000, "%s\t%i\t%c\t%" PRIu32 "\t%" PRIu32 "\tC%s\t%s\n", chrom, pos + 1, strand, nmethyl, nunmethyl, context, tnc);
    }
    kputs_(ks, str);
}
static char revcomp(char b) {
    switch(b) { case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A'; default: return 'N'; }
}
static int tri_idx(char base) { switch(base) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }
static int getTriNucContext(char *seq, uint32_t offset, int seqlen, int direction) {   /* extract.c:120-180 */
    int rv = 0; char base;
    if((direction > 0 && (int64_t)offset + 2 >= seqlen) || (direction < 0 && offset <= 1)) rv = 4;
    else { base = seq[(int64_t)offset + 2 * direction]; if(direction < 0) base = revcomp(base); rv = tri_idx(base); }
    if((direction > 0 && (int64_t)offset + 1 >= seqlen) || (direction < 0 && offset == 0)) rv += 20;
    else { base = seq[(int64_t)offset + direction]; if(direction < 0) base = revcomp(base); rv += 5 * tri_idx(base); }
    return rv;
}
static void writeBlank(kstr **ks, Config *config, const char *chrom, int32_t pos, uint32_t localPos2, uint32_t *lastPos, char *seq, int seqlen) {
    int triNucContext = 0, direction = 0; char context[3] = "HG";
    if(pos == -1) return;
    for(; (int64_t)*lastPos < pos; (*lastPos)++) {
        if((direction = isCpG(seq, *lastPos - localPos2, seqlen)) != 0) { if(!config->keepCpG) continue; context[0] = 'G'; context[1] = 0; }
        else if((direction = isCHG(seq, *lastPos - localPos2, seqlen)) != 0) { if(!config->keepCHG) continue; context[0] = 'H'; context[1] = 'G'; }
        else if((direction = isCHH(seq, *lastPos - localPos2, seqlen)) != 0) { if(!config->keepCHH) continue; context[0] = 'H'; context[1] = 'H'; }
        else continue;
        triNucContext = getTriNucContext(seq, *lastPos - localPos2, seqlen, direction);
        writeCall(ks[0], config, chrom, *lastPos, 1, 0, 0, (direction > 0) ? 'C' : 'G', context, TriNucleotideContexts[triNucContext]);
    }
}
static void processLast(kstr *ks, Config *config, struct lastCall *last, const bamfile *hdr, int32_t tid, int32_t pos, int width, uint32_t nmethyl, uint32_t nunmethyl, char base) {
    if(last->tid == tid && last->pos == pos) {
        nmethyl += last->nmethyl; nunmethyl += last->nunmethyl;
        writeCall(ks, config, hdr->target_name[tid], pos, width, nmethyl, nunmethyl, base, NULL, NULL);
        last->tid = -1;
    } else {
        if(last->tid != -1) writeCall(ks, config, hdr->target_name[last->tid], last->pos, width, last->nmethyl, last->nunmethyl, base, NULL, NULL);
        last->tid = tid; last->pos = pos; last->nmethyl = nmethyl; last->nunmethyl = nunmethyl;
    }
}
static int isVariant(Config *config, const pileup1 *plp, uint32_t *coverage, int strand) {   /* extract.c:225-239 */
    uint8_t base;
    if(plp->qpos >= plp->b->r->l_qseq) return 0;          /* as in methState: no stored base */
    base = seqi(plp->b->seq, plp->qpos);
    if(plp->b->qual[plp->qpos] < config->minPhred) return 0;
    *coverage += 1;
    if(strand & 1) { if(base != 4 && base != 15) return 1; else return 0; }
    else { if(base != 2 && base != 15) return 1; else return 0; }
}

/* globals of main.c:7-15 */
static uint32_t globalTid, globalPos, globalEnd, bin_, outputBin;
static uint64_t globalnVariantPositions;
static pthread_mutex_t positionMutex = PTHREAD_MUTEX_INITIALIZER, outputMutex = PTHREAD_MUTEX_INITIALIZER;   /* main.c:14-15 */
static pthread_cond_t outputCv = PTHREAD_COND_INITIALIZER;      /* the reference spins on outputMutex (extract.c:514-520); waiting on a condition gives the same order */
static void awaitTurn(uint32_t localBin) { pthread_mutex_lock(&outputMutex); while(outputBin != localBin) pthread_cond_wait(&outputCv, &outputMutex); }
static void passTurn(void) { outputBin++; pthread_cond_broadcast(&outputCv); pthread_mutex_unlock(&outputMutex); }
typedef struct { Config *config; const bamfile *bf; const fasta *fa; } extractArgs;
static FILE *dump_fp;    /* MDK_ORACLE_DUMP: per-column raw counters, used by the kernel-level parity tests */

static void *extractCalls(void *arg_) {   /* extract.c:247-560; one call per worker thread (extract.c:1479-1486) */
    extractArgs *arg = arg_; Config *config = arg->config; const bamfile *bf = arg->bf; const fasta *fa = arg->fa;
    int tid = 0, i, seqlen, type, rv, n_plp, strand, direction, tnc;
    int32_t pos = 0;
    uint32_t nmethyl = 0, nunmethyl = 0, nOff = 0, nVariant = 0;
    uint32_t localPos = 0, localEnd = 0, localTid = 0, localPos2 = 0, lastPos = 0, localBin = 0;
    uint64_t nVariantPositions = 0;
    pileup1 *plp; char *seq = NULL, base = 'A'; char context[3] = "HG";
    struct lastCall lastCpG_, lastCHG_, *lastCpG = NULL, *lastCHG = NULL;
    kstr os_[3], *os[3]; mplp_data data; int32_t bedIdx = 0; int o;
    memset(os_, 0, sizeof(os_)); os[0] = &os_[0]; os[1] = &os_[1]; os[2] = &os_[2];
    for(i = 0; i < 3; i++) { os_[i].m = 1024; os_[i].s = xmalloc(1024); os_[i].s[0] = 0; }
    if(config->merge) {
        if(config->keepCpG) { lastCpG = &lastCpG_; lastCpG->tid = -1; }
        if(config->keepCHG) { lastCHG = &lastCHG_; lastCHG->tid = -1; }
    }
    memset(&data, 0, sizeof(data)); data.config = config; data.bf = bf;

    while(1) {
        plp_t iter; ohash_t *oh;
        pthread_mutex_lock(&positionMutex);                    /* extract.c:325-350: claim the next chunk */
        localBin = bin_++;
        localTid = globalTid; localPos = globalPos;
        localEnd = (uint32_t)(localPos + config->chunkSize);
        if(localTid >= (uint32_t)bf->n_targets) { pthread_mutex_unlock(&positionMutex); break; }
        if(globalEnd && localEnd > globalEnd) localEnd = globalEnd;
        adjustBounds(bf, fa, &localTid, &localPos, &localEnd);
        globalPos = localEnd;
        if(globalEnd > 0 && globalPos >= globalEnd) globalTid = (uint32_t)-1;
        if(localTid < (uint32_t)bf->n_targets && globalTid != (uint32_t)-1) {
            if(globalPos >= bf->target_len[localTid]) { localEnd = bf->target_len[localTid]; globalTid++; globalPos = 0; }
        }
        pthread_mutex_unlock(&positionMutex);
        if(config->bed) {   /* extract.c:352-369: skip chunks that touch no BED region (the bin is marked as written) */
            if(spanOverlapsBED((int32_t)localTid, (int32_t)localPos, (int32_t)localEnd, config->bed, &bedIdx) != 1) { awaitTurn(localBin); passTurn(); continue; }
        }
        localPos2 = 0; if(localPos > 1) localPos2 = localPos - 2;
        lastPos = localPos;
        if(localTid >= (uint32_t)bf->n_targets) break;
        if(globalEnd && localPos >= globalEnd) break;
        regitr_init(&data.iter, bf, (int32_t)localTid, (int32_t)localPos, (int32_t)localEnd);
        seq = fetch_seq(fa, bf->target_name[localTid], (int)localPos2, (int)(localEnd + 10), &seqlen);
        if(seqlen < 0) {
            fprintf(stderr, "faidx_fetch_seq returned %i while trying to fetch the sequence for tid %s:%" PRIu32 "-%" PRIu32 "!\n", seqlen, bf->target_name[localTid], localPos2, localEnd);
            fprintf(stderr, "Note that the output will be truncated!\n");
            awaitTurn(localBin); passTurn();        /* the reference `continue`s without releasing its bin (extract.c:382-387): with -@ > 1 every later chunk then waits forever; not reproduced */
            continue;
        }
        data.seq = seq; data.offset = localPos2; data.lseq = seqlen;
        oh = initOlapHash();
        plp_init(&iter, &data, oh);

        while((plp = plp_auto(&iter, &tid, &pos, &n_plp)) != NULL) {
            if((uint32_t)pos < localPos || (uint32_t)pos >= localEnd) continue;
            if(config->bed) {   /* extract.c:402-405 */
                while((o = posOverlapsBED(tid, pos, config->bed, bedIdx)) == -1) bedIdx++;
                if(o == 0) continue;
            }
            if((direction = isCpG(seq, pos - localPos2, seqlen))) { if(!config->keepCpG) continue; type = 0; }
            else if((direction = isCHG(seq, pos - localPos2, seqlen))) { if(!config->keepCHG) continue; type = 1; }
            else if((direction = isCHH(seq, pos - localPos2, seqlen))) { if(!config->keepCHH) continue; type = 2; }
            else continue;

            nmethyl = nunmethyl = nVariant = nOff = 0;
            base = seq[pos - localPos2];
            for(i = 0; i < n_plp; i++) {
                if(plp[i].is_del) continue;
                if(plp[i].is_refskip) continue;
                if(config->bed) if(!readStrandOverlapsBED(plp[i].b->r, config->bed->region[bedIdx])) continue;   /* extract.c:425 */
                strand = getStrand(plp[i].b->r);
                if(strand & 1) { if(base != 'C' && base != 'c') { nVariant += isVariant(config, plp + i, &nOff, strand); continue; } }
                else { if(base != 'G' && base != 'g') { nVariant += isVariant(config, plp + i, &nOff, strand); continue; } }
                rv = methState(config, plp[i].b, plp[i].qpos);
                if(rv > 0) nmethyl++; else if(rv < 0) nunmethyl++;
            }
            if(dump_fp && (nmethyl + nunmethyl > 0 || nOff > 0))
                fprintf(dump_fp, "%d\t%d\t%d\t%d\t%u\t%u\t%u\t%u\n", tid, pos, type, (base == 'G' || base == 'g'), nmethyl, nunmethyl, nOff, nVariant);

            if(config->minOppositeDepth > 0 && nOff >= (uint32_t)config->minOppositeDepth && ((double)nVariant) / ((double)nOff) >= config->maxVariantFrac) {
                nVariantPositions++;
                if(config->merge) {
                    if(type == 0 && lastCpG->tid == tid && lastCpG->pos == pos - 1 && (base == 'G' || base == 'g')) { lastCpG->nmethyl = 0; lastCpG->nunmethyl = 0; }
                    else if(type == 1 && lastCHG->tid == tid && lastCHG->pos == pos - 2 && (base == 'G' || base == 'g')) { lastCHG->nmethyl = 0; lastCHG->nunmethyl = 0; }
                }
                continue;
            }
            if(nmethyl + nunmethyl == 0 && config->cytosine_report == 0) continue;
            if(!config->merge || type == 2) {
                if(config->cytosine_report) {
                    writeBlank(os, config, bf->target_name[localTid], pos, localPos2, &lastPos, seq, seqlen);
                    if(type == 0) { context[0] = 'G'; context[1] = 0; }
                    else if(type == 1) { context[0] = 'H'; context[1] = 'G'; }
                    else { context[0] = 'H'; context[1] = 'H'; }
                    tnc = getTriNucContext(seq, pos - localPos2, seqlen, direction);
                    writeCall(os[0], config, bf->target_name[tid], pos, 1, nmethyl, nunmethyl, base, context, TriNucleotideContexts[tnc]);
                } else writeCall(os[type], config, bf->target_name[tid], pos, 1, nmethyl, nunmethyl, base, NULL, NULL);
            } else {
                if(type == 0) { if(base == 'G' || base == 'g') pos--; processLast(os[0], config, lastCpG, bf, tid, pos, 2, nmethyl, nunmethyl, base); }
                else { if(base == 'G' || base == 'g') pos -= 2; processLast(os[1], config, lastCHG, bf, tid, pos, 3, nmethyl, nunmethyl, base); }
            }
            lastPos = pos + 1;
        }
        plp_destroy(&iter);

        nmethyl = 0; nunmethyl = 0;
        if(config->merge) {
            if(config->keepCpG && lastCpG->tid != -1) { processLast(os[0], config, lastCpG, bf, tid, pos, 2, nmethyl, nunmethyl, base); lastCpG->tid = -1; }
            if(config->keepCHG && lastCHG->tid != -1) { processLast(os[1], config, lastCHG, bf, tid, pos, 3, nmethyl, nunmethyl, base); lastCHG->tid = -1; }
        } else if(config->cytosine_report) {
            writeBlank(os, config, bf->target_name[localTid], localEnd, localPos2, &lastPos, seq, seqlen);
        }
        free(seq);
        /* ordered flush (extract.c:514-535) */
        awaitTurn(localBin);
        if(config->cytosine_report) { if(os[0]->l) { fputs(os[0]->s, config->output_fp[0]); os[0]->l = 0; os[0]->s[0] = 0; } }
        else {
            if(config->keepCpG && os[0]->l) { fputs(os[0]->s, config->output_fp[0]); os[0]->l = 0; os[0]->s[0] = 0; }
            if(config->keepCHG && os[1]->l) { fputs(os[1]->s, config->output_fp[1]); os[1]->l = 0; os[1]->s[0] = 0; }
            if(config->keepCHH && os[2]->l) { fputs(os[2]->s, config->output_fp[2]); os[2]->l = 0; os[2]->s[0] = 0; }
        }
        passTurn();
        destroyOlapHash(oh);
    }
    for(i = 0; i < 3; i++) free(os_[i].s);
    if(nVariantPositions > 0) { pthread_mutex_lock(&outputMutex); globalnVariantPositions += nVariantPositions; pthread_mutex_unlock(&outputMutex); }
    return NULL;
}

static void printHeader(FILE *of, const char *context, char *opref, Config config) {   /* extract.c:562-569 */
    fprintf(of, "track type=\"bedGraph\" description=\"%s %s", opref, context);
    if(config.merge) fprintf(of, " merged");
    if(config.fraction) fprintf(of, " methylation fractions\"\n");
    else if(config.counts) fprintf(of, " methylation counts\"\n");
    else if(config.logit) fprintf(of, " logit transformed methylation fractions\"\n");
    else fprintf(of, " methylation levels\"\n");
}

static void parseBounds(char *s2, int *vals, int mult) {       /* common.c:11-43 */
    char *p, *s = strdup(s2), *end; int i, v; long tempV;
    p = strtok(s, ",");
    if(!p) { fprintf(stderr, "Invalid bounds string, %s\n", s2); free(s); return; }
    errno = 0;      /* NOT in the reference, which tests whatever errno was left by earlier calls; cleared here so that the result is defined */
    tempV = strtol(p, &end, 10);
    if((errno == ERANGE && (tempV == LONG_MAX || tempV == LONG_MIN)) || (errno != 0 && tempV == 0) || end == p) v = -1;
    else if(tempV > INT_MAX || tempV < LONG_MIN) v = -1; else v = (int)tempV;
    if(v >= 0) vals[4 * mult] = v; else { fprintf(stderr, "Invalid bounds string, %s\n", s2); free(s); return; }
    for(i = 1; i < 4; i++) {
        p = strtok(NULL, ",");
        if(!p) { fprintf(stderr, "Invalid bounds string, %s\n", s2); free(s); return; }   /* reference segfaults here */
        errno = 0;
        tempV = strtol(p, &end, 10);
        if((errno == ERANGE && (tempV == LONG_MAX || tempV == LONG_MIN)) || (errno != 0 && tempV == 0) || end == p) v = -1;
        else if(tempV > INT_MAX || tempV < LONG_MIN) v = -1; else v = (int)tempV;
        if(v >= 0) vals[4 * mult + i] = v; else { fprintf(stderr, "Invalid bounds string, %s\n", s2); free(s); return; }
    }
    free(s);
}

/* hts_parse_reg (htslib): "chr", "chr:beg", "chr:beg-", "chr:beg-end", "chr:-end"; commas allowed */
static const char *parse_reg(const char *s, int *beg, int *end) {
    const char *colon = strrchr(s, ':'), *p; int64_t b = 0, e = 0; int nd;
    if(!colon) { *beg = 0; *end = INT_MAX; return s + strlen(s); }
    p = colon + 1;
    if(*p == '-') { /* chr:-100 == chr:1-100 */
        p++; nd = 0; while((*p >= '0' && *p <= '9') || *p == ',') { if(*p != ',') { e = e * 10 + (*p - '0'); nd++; } p++; }
        if(*p || !nd) return NULL;
        *beg = 0; *end = e > INT_MAX ? INT_MAX : (int)e; return colon;
    }
    nd = 0; while((*p >= '0' && *p <= '9') || *p == ',') { if(*p != ',') { b = b * 10 + (*p - '0'); nd++; } p++; }
    b -= 1;
    if(b < 0) { if(nd && *p == '-') return NULL; *beg = 0; *end = INT_MAX; if(*p) return NULL; return colon; }
    if(*p == 0) e = INT_MAX;
    else if(*p == '-') { p++; while((*p >= '0' && *p <= '9') || *p == ',') { if(*p != ',') e = e * 10 + (*p - '0'); p++; } if(*p) return NULL; }
    else return NULL;
    if(e == 0) e = INT_MAX;
    if(e > INT_MAX) e = INT_MAX;
    if(b >= e) return NULL;
    *beg = (int)b; *end = (int)e; return colon;
}

static int bbm_error(void) { printf("fatal: malformed BBM file\n"); return -9; }

static void extract_usage(void) {
    fprintf(stderr, "\nUsage: MethylDackel extract [OPTIONS] <ref.fa> <sorted_alignments.bam>\n");
    fprintf(stderr, "(mdk_oracle: CPU oracle of the reference's option surface; see the reference for the option text)\n");
}

static int extract_main(int argc, char *argv[]) {              /* extract.c:706-1514 */
    char *opref = NULL, *oname, *p; int c, i; Config config; bamfile bf; fasta fa; fasta_job fa_job; pthread_t fa_th; int fa_bg = 0;
    FILE *BBM_ptr = NULL; char *BWName = NULL; int outputBB = 0, noBAM = 0; char *bedName = NULL; int keepStrand = 0;
    const char *FastaName, *BAMName;
    static struct option lopts[] = {
        {"opref", 1, NULL, 'o'}, {"fraction", 0, NULL, 'f'}, {"counts", 0, NULL, 'c'}, {"logit", 0, NULL, 'm'},
        {"minDepth", 1, NULL, 'd'}, {"noCpG", 0, NULL, 1}, {"CHG", 0, NULL, 2}, {"CHH", 0, NULL, 3},
        {"keepDupes", 0, NULL, 4}, {"keepSingleton", 0, NULL, 5}, {"keepDiscordant", 0, NULL, 6},
        {"OT", 1, NULL, 7}, {"OB", 1, NULL, 8}, {"CTOT", 1, NULL, 9}, {"CTOB", 1, NULL, 10},
        {"mergeContext", 0, NULL, 11}, {"methylKit", 0, NULL, 12},
        {"nOT", 1, NULL, 13}, {"nOB", 1, NULL, 14}, {"nCTOT", 1, NULL, 15}, {"nCTOB", 1, NULL, 16},
        {"minOppositeDepth", 1, NULL, 17}, {"maxVariantFrac", 1, NULL, 18}, {"chunkSize", 1, NULL, 19},
        {"keepStrand", 0, NULL, 20}, {"cytosine_report", 0, NULL, 21}, {"minConversionEfficiency", 1, NULL, 22},
        {"ignoreNH", 0, NULL, 23}, {"ignoreFlags", 1, NULL, 'F'}, {"requireFlags", 1, NULL, 'R'},
        {"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'}, {"mappability", 1, NULL, 'M'},
        {"mappabilityThreshold", 1, NULL, 't'}, {"minMappableBases", 1, NULL, 'b'},
        {"outputBBMFile", 1, NULL, 'O'}, {"outputBBMFileName", 1, NULL, 'N'}, {"mappabilityBBM", 1, NULL, 'B'},
        {0, 0, NULL, 0}};

    globalTid = globalPos = globalEnd = bin_ = outputBin = 0; globalnVariantPositions = 0;
    memset(&config, 0, sizeof(config));
    config.mappabilityCutoff = 0.01; config.minMappableBases = 15;
    config.keepCpG = 1; config.minMapq = 10; config.minPhred = 5; config.minDepth = 1;
    config.ignoreFlags = 0xF00; config.nThreads = 1; config.chunkSize = 1000000;

    optind = 1;
    while((c = getopt_long(argc, argv, "hvq:p:r:l:o:D:f:c:m:d:F:R:@:M:t:b:ON:B:", lopts, NULL)) >= 0) {
        switch(c) {
        case 'h': extract_usage(); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", ORACLE_VERSION, "none: mdk_oracle"); return 0;
        case 'o': opref = strdup(optarg); break;
        case 'D': break;
        case 'd': config.minDepth = atoi(optarg); if(config.minDepth < 1) { fprintf(stderr, "Error, the minimum depth must be at least 1!\n"); return 1; } break;
        case 'r': config.reg = optarg; break;
        case 'l': bedName = optarg; break;
        case 1: config.keepCpG = 0; break;
        case 2: config.keepCHG = 1; break;
        case 3: config.keepCHH = 1; break;
        case 4: config.keepDupes = 1; break;
        case 5: config.keepSingleton = 1; break;
        case 6: config.keepDiscordant = 1; break;
        case 7: parseBounds(optarg, config.bounds, 0); break;
        case 8: parseBounds(optarg, config.bounds, 1); break;
        case 9: parseBounds(optarg, config.bounds, 2); break;
        case 10: parseBounds(optarg, config.bounds, 3); break;
        case 11: config.merge = 1; break;
        case 12: config.methylKit = 1; break;
        case 13: parseBounds(optarg, config.absoluteBounds, 0); break;
        case 14: parseBounds(optarg, config.absoluteBounds, 1); break;
        case 15: parseBounds(optarg, config.absoluteBounds, 2); break;
        case 16: parseBounds(optarg, config.absoluteBounds, 3); break;
        case 17: config.minOppositeDepth = atoi(optarg); break;
        case 18: config.maxVariantFrac = atof(optarg); break;
        case 19: config.chunkSize = strtoul(optarg, NULL, 10); if(config.chunkSize < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); return 1; } break;
        case 20: keepStrand = 1; break;
        case 21: config.cytosine_report = 1; break;
        case 22: config.minConversionEfficiency = atof(optarg); break;
        case 23: config.ignoreNH = 1; break;
        case 'M': BWName = optarg; break;
        case 't': config.mappabilityCutoff = atof(optarg); break;
        case 'b': config.minMappableBases = atoi(optarg); break;
        case 'O': outputBB = 1; break;
        case 'N': outputBB = 1; break;
        case 'B': config.BBMName = optarg; break;
        case 'F': config.ignoreFlags = atoi(optarg); break;
        case 'R': config.requireFlags = atoi(optarg); break;
        case 'q': config.minMapq = atoi(optarg); break;
        case 'p': config.minPhred = atoi(optarg); break;
        case 'm': config.logit = 1; break;
        case 'f': config.fraction = 1; break;
        case 'c': config.counts = 1; break;
        case '@': config.nThreads = atoi(optarg); break;
        case '?': default: fprintf(stderr, "Invalid option '%c'\n", c); extract_usage(); return 1;
        }
    }
    if(outputBB && !BWName) { fprintf(stderr, "You must specify a bigWig file when attempting to create a BBM file!\n"); extract_usage(); return -1; }
    if(argc == 1) { extract_usage(); return 0; }
    if(argc - optind < 2) {
        if(outputBB) noBAM = 1;
        else { fprintf(stderr, "You must supply a reference genome in fasta format and an input BAM file!!!\n"); extract_usage(); return -1; }
    }
    if(config.minPhred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", config.minPhred); config.minPhred = 1; }
    if(config.minMapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", config.minMapq); config.minMapq = 0; }
    if(config.keepDupes > 0 && (config.ignoreFlags & 0x400)) config.ignoreFlags -= 0x400;
    if(config.fraction + config.counts + config.logit + config.methylKit + config.cytosine_report > 1) {
        fprintf(stderr, "More than one of --fraction, --counts, --methylKit, --cytosine_report and --logit were specified. These are mutually exclusive.\n");
        extract_usage(); return 1;
    }
    if(config.methylKit + config.merge == 2) { fprintf(stderr, "--mergeContext and --methylKit are mutually exclusive.\n"); extract_usage(); return 1; }
    if(config.cytosine_report + config.merge == 2) { fprintf(stderr, "--mergeContext and --cytosine_report are mutually exclusive.\n"); extract_usage(); return 1; }
    if(config.fraction + config.counts + config.logit > 1) { fprintf(stderr, "You may specify AT MOST one of -c/--counts, -f/--fraction, or -m/--logit.\n"); return -6; }
    if(!(config.keepCpG + config.keepCHG + config.keepCHH)) {
        fprintf(stderr, "You haven't specified any metrics to output!\nEither don't use the --noCpG option or specify --CHG and/or --CHH.\n");
        return -1;
    }
    if(BWName || noBAM) { fprintf(stderr, "mdk_oracle: bigWig input (-M/-O/-N) needs libBigWig, which is not available; use -B <file.bbm>\n"); return -4; }

    FastaName = argv[optind]; BAMName = argv[optind + 1];
    g_load_threads = config.nThreads;
    /* (the reference's workers fetch their windows of the FASTA as they go, extract.c:381; this oracle reads it whole -- on a thread of its own, next to the BAM) */
    fa_job.fn = FastaName; fa_job.fa = &fa; fa_job.rc = 0;
    fa_bg = config.nThreads > 1 && pthread_create(&fa_th, NULL, fasta_bg_main, &fa_job) == 0;
    if((i = bam_load(BAMName, &bf)) != 0) { if(fa_bg) pthread_join(fa_th, NULL); fprintf(stderr, "Couldn't open %s for reading!\n", BAMName); return -4; }
    if(config.BBMName && (BBM_ptr = fopen(config.BBMName, "rb")) == NULL) { fprintf(stderr, "Couldn't open %s for reading!\n", config.BBMName); return -8; }

    if(BBM_ptr) {                                              /* extract.c:1236-1339 */
        signed char readlen; unsigned char bbm_version = 0; int chromID = 0;
        config.filterMappability = 1;
        fprintf(stderr, "loading mappability data from %s\n", config.BBMName);
        readlen = (signed char)fread(&bbm_version, sizeof(char), 1, BBM_ptr);
        if(bbm_version != BBM_VERSION) { fprintf(stderr, "fatal: %s has wrong BBM version or is malformed\n", config.BBMName); return -10; }
        readlen = (signed char)fread(&config.chromCount, sizeof(config.chromCount), 1, BBM_ptr);
        config.chromNames = xmalloc(config.chromCount * sizeof(char *));
        config.chromLengths = xmalloc(config.chromCount * sizeof(uint32_t));
        if(readlen <= 0) return bbm_error();
        config.bw_data = xmalloc(config.chromCount * sizeof(char *));
        while(chromID < (int)config.chromCount) {
            uint16_t nameLen = 0; char nullterm = 1; uint32_t bpos = 0; int arrlen;
            readlen = (signed char)fread(&nameLen, sizeof(uint16_t), 1, BBM_ptr);
            config.chromNames[chromID] = xmalloc((size_t)nameLen + 1);
            for(i = 0; i < nameLen; i++) readlen = (signed char)fread(&(config.chromNames[chromID][i]), sizeof(char), 1, BBM_ptr);
            config.chromNames[chromID][nameLen] = 0;
            readlen = (signed char)fread(&nullterm, sizeof(char), 1, BBM_ptr);
            if(nullterm) return bbm_error();
            readlen = (signed char)fread(&(config.chromLengths[chromID]), sizeof(uint32_t), 1, BBM_ptr);
            arrlen = config.chromLengths[chromID] / 8; if(config.chromLengths[chromID] % 8 > 0) arrlen++;
            config.bw_data[chromID] = xmalloc((size_t)arrlen + 1);
            while(bpos < config.chromLengths[chromID]) {
                int index = bpos / 8; char offset = bpos % 8, aboveCutoff; unsigned char val = 0; uint16_t runlen = 0;
                if(offset == 0) config.bw_data[chromID][index] = 0;
                if(fread(&val, sizeof(val), 1, BBM_ptr) != 1) return bbm_error();   /* reference would loop forever on a truncated file */
                if(val > 100) {
                    if(val == 255) { readlen = (signed char)fread(&runlen, sizeof(uint16_t), 1, BBM_ptr); readlen = (signed char)fread(&val, sizeof(val), 1, BBM_ptr); }
                    else { runlen = val - RUNOFFSET; readlen = (signed char)fread(&val, sizeof(val), 1, BBM_ptr); }
                    aboveCutoff = (char)(val >= config.mappabilityCutoff * 100.0);
                    for(i = 0; i < runlen; i++) {
                        int tempindex = (bpos + i) / 8; char tempoffset = (bpos + i) % 8;
                        if(tempindex >= arrlen) break;         /* reference writes out of bounds on an over-long run */
                        if(tempoffset == 0) config.bw_data[chromID][tempindex] = 0;
                        config.bw_data[chromID][tempindex] = config.bw_data[chromID][tempindex] | (aboveCutoff << tempoffset);
                    }
                    bpos += runlen;
                    if(runlen == 0) return bbm_error();        /* reference would loop forever */
                } else {
                    aboveCutoff = (char)(val >= config.mappabilityCutoff * 100.0);
                    config.bw_data[chromID][index] = config.bw_data[chromID][index] | (aboveCutoff << offset);
                    bpos++;
                }
            }
            chromID++;
        }
        (void)readlen;
        fclose(BBM_ptr);
    }

    if(fa_bg) { pthread_join(fa_th, NULL); if(fa_job.rc) { fprintf(stderr, "Couldn't open the index for %s!\n", FastaName); return -4; } }
    else if(fasta_load(FastaName, &fa) != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", FastaName); return -4; }

    /* output files (extract.c:1343-1439) */
    if(opref == NULL) {
        opref = strdup(argv[optind + 1]);
        p = strrchr(opref, '.'); if(p != NULL) *p = '\0';
        fprintf(stderr, "writing to prefix:'%s'\n", opref);
    }
    oname = xmalloc(strlen(opref) + 32);
    if(config.cytosine_report) {
        sprintf(oname, "%s.cytosine_report.txt", opref);
        config.output_fp[0] = fopen(oname, "w"); config.output_fp[1] = config.output_fp[0]; config.output_fp[2] = config.output_fp[0];
        if(!config.output_fp[0]) return -3;
    }
    {
        static const char *ctx[3] = {"CpG", "CHG", "CHH"}; int keep[3]; keep[0] = config.keepCpG; keep[1] = config.keepCHG; keep[2] = config.keepCHH;
        for(i = 0; i < 3; i++) {
            if(!keep[i] || config.cytosine_report) continue;
            if(config.fraction) sprintf(oname, "%s_%s.meth.bedGraph", opref, ctx[i]);
            else if(config.counts) sprintf(oname, "%s_%s.counts.bedGraph", opref, ctx[i]);
            else if(config.logit) sprintf(oname, "%s_%s.logit.bedGraph", opref, ctx[i]);
            else if(config.methylKit) sprintf(oname, "%s_%s.methylKit", opref, ctx[i]);
            else sprintf(oname, "%s_%s.bedGraph", opref, ctx[i]);
            config.output_fp[i] = fopen(oname, "w");
            if(config.output_fp[i] == NULL) { fprintf(stderr, "Couldn't open the output %s metrics file for writing! Insufficient permissions?\n", ctx[i]); return -3; }
            if(config.methylKit) fprintf(config.output_fp[i], "chrBase\tchr\tbase\tstrand\tcoverage\tfreqC\tfreqT\n");
            else printHeader(config.output_fp[i], ctx[i], opref, config);
        }
    }
    if(config.reg) {                                           /* extract.c:1441-1468 */
        const char *foo; char *bar; int s = 0, e = 0;
        foo = parse_reg(config.reg, &s, &e);
        if(foo == NULL) { fprintf(stderr, "Could not parse the specified region!\n"); return -4; }
        bar = xmalloc((size_t)(foo - config.reg) + 1);
        strncpy(bar, config.reg, (size_t)(foo - config.reg)); bar[foo - config.reg] = 0;
        globalTid = (uint32_t)-1;
        for(i = 0; i < bf.n_targets; i++) if(!strcmp(bf.target_name[i], bar)) { globalTid = (uint32_t)i; break; }
        if(globalTid == (uint32_t)-1) { fprintf(stderr, "%s did not match a known chromosome/contig name!\n", config.reg); return -6; }
        if(s > 0) globalPos = (uint32_t)s;
        if(e > 0) globalEnd = (uint32_t)e;
        if(globalEnd > bf.target_len[globalTid]) globalEnd = bf.target_len[globalTid];
        free(bar);
    }
    if(bedName) {       /* extract.c:1469-1477 */
        config.bed = parseBED(bedName, &bf, keepStrand);
        if(!config.bed) { fprintf(stderr, "There was an error while reading in your BED file!\n"); return 1; }
    }
    if(getenv("MDK_ORACLE_DUMP")) dump_fp = fopen(getenv("MDK_ORACLE_DUMP"), "w");

    {   /* extract.c:1479-1486 */
        PHASE_SERIAL("fasta, outputs");
        extractArgs ea = {&config, &bf, &fa}; int nt = config.nThreads < 1 ? 1 : config.nThreads; pthread_t *threads = calloc((size_t)nt, sizeof(pthread_t));
        if(dump_fp) nt = 1;      /* the per-column dump is written as columns are finished: one worker keeps it ordered */
        for(i = 1; i < nt; i++) pthread_create(threads + i, NULL, extractCalls, &ea);
        extractCalls(&ea);
        for(i = 1; i < nt; i++) pthread_join(threads[i], NULL);
        free(threads);
        PHASE("chunks (all workers)");
        if(getenv("MDK_ORACLE_PROFILE")) fprintf(stderr, "[oracle] serial phases %.3f s of %.3f s = %.1f %%\n", g_t_serial, g_t_all, g_t_all > 0 ? 100.0 * g_t_serial / g_t_all : 0.0);
    }

    if(dump_fp) { fclose(dump_fp); dump_fp = NULL; }
    if(globalnVariantPositions) printf("%" PRIu64 " positions were excluded due to likely being variants.\n", globalnVariantPositions);
    if(config.cytosine_report) fclose(config.output_fp[0]);
    if(config.keepCpG && !config.cytosine_report) fclose(config.output_fp[0]);
    if(config.keepCHG && !config.cytosine_report) fclose(config.output_fp[1]);
    if(config.keepCHH && !config.cytosine_report) fclose(config.output_fp[2]);
    free(opref); free(oname);
    return 0;
}


/* ------------------------------------------------------------------------------------------ */
/* MBias.c + svg.c restated (`mbias`).  The reference's tests hold no expectation for this       */
/* command: parity for mbias is UNPINNED by the reference and rests on this restatement alone.  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t l, m; uint32_t *unmeth1, *unmeth2, *meth1, *meth2; } strandMeth;      /* MethylDackel.h:171-176 */
#define kroundup32_(x) (--(x), (x) |= (x) >> 1, (x) |= (x) >> 2, (x) |= (x) >> 4, (x) |= (x) >> 8, (x) |= (x) >> 16, ++(x))

static strandMeth *growStrandMeth(strandMeth *s, int32_t l) {   /* MBias.c:16-40 */
    int32_t m; int i;
    l++;
    m = l; kroundup32_(m);
    if(m < 32) m = 32;
    s->unmeth1 = xrealloc(s->unmeth1, sizeof(uint32_t) * (size_t)m);
    s->meth1 = xrealloc(s->meth1, sizeof(uint32_t) * (size_t)m);
    s->unmeth2 = xrealloc(s->unmeth2, sizeof(uint32_t) * (size_t)m);
    s->meth2 = xrealloc(s->meth2, sizeof(uint32_t) * (size_t)m);
    for(i = s->m; i < m; i++) { s->unmeth1[i] = 0; s->meth1[i] = 0; s->unmeth2[i] = 0; s->meth2[i] = 0; }
    s->m = m;
    return s;
}
static strandMeth *mergeStrandMeth(strandMeth *target, strandMeth *source) {   /* MBias.c:42-55 */
    int32_t i;
    if(source->l == 0) return target;
    if(target->m < source->l) target = growStrandMeth(target, source->m);
    if(target->l < source->l) target->l = source->l;
    for(i = 0; i < source->l; i++) {
        target->unmeth1[i] += source->unmeth1[i]; target->meth1[i] += source->meth1[i];
        target->unmeth2[i] += source->unmeth2[i]; target->meth2[i] += source->meth2[i];
    }
    return target;
}

/* extractMBias (MBias.c:57-230).  A contig missing from the FASTA makes the reference's worker return NULL, which
 * mbias_main then dereferences (MBias.c:150-155,543-546); here that is a fatal error. */
static strandMeth **extractMBias(Config *config, const bamfile *bf, const fasta *fa) {
    int tid = 0, i, seqlen, rv, n_plp, strand, o = 0; int32_t pos = 0, bedIdx = 0;
    pileup1 *plp; char *seq = NULL, base;
    strandMeth **meths = xmalloc(4 * sizeof(strandMeth *));
    uint32_t localPos = 0, localEnd = 0, localTid = 0; mplp_data data;
    for(i = 0; i < 4; i++) meths[i] = calloc(1, sizeof(strandMeth));
    memset(&data, 0, sizeof(data)); data.config = config; data.bf = bf; data.bedIdx = bedIdx;

    while(1) {
        plp_t iter;
        localTid = globalTid; localPos = globalPos;
        localEnd = (uint32_t)(localPos + config->chunkSize);
        if(localTid >= (uint32_t)bf->n_targets) break;
        if(globalEnd && localEnd > globalEnd) localEnd = globalEnd;
        adjustBounds(bf, fa, &localTid, &localPos, &localEnd);
        globalPos = localEnd;
        if(globalEnd > 0 && globalPos >= globalEnd) globalTid = (uint32_t)-1;
        if(localTid < (uint32_t)bf->n_targets && globalTid != (uint32_t)-1) {
            if(globalPos >= bf->target_len[localTid]) { localEnd = bf->target_len[localTid]; globalTid++; globalPos = 0; }
        }
        if(config->bed) { if(spanOverlapsBED((int32_t)localTid, (int32_t)localPos, (int32_t)localEnd, config->bed, &bedIdx) != 1) continue; }
        if(localTid >= (uint32_t)bf->n_targets) break;
        if(globalEnd && localPos >= globalEnd) break;
        regitr_init(&data.iter, bf, (int32_t)localTid, (int32_t)localPos, (int32_t)localEnd);
        seq = fetch_seq(fa, bf->target_name[localTid], (int)localPos, (int)localEnd, &seqlen);      /* NB no 2-base lead-in, 1 base past the end */
        if(seqlen < 0) {
            fprintf(stderr, "faidx_fetch_seq returned %i while trying to fetch the sequence for tid %s:%" PRIu32 "-%" PRIu32 "!\n", seqlen, bf->target_name[localTid], localPos, localEnd);
            fprintf(stderr, "Note that the output will be truncated!\n");
            exit(3);
        }
        data.seq = seq; data.lseq = seqlen; data.offset = localPos;
        plp_init(&iter, &data, NULL);
        while((plp = plp_auto(&iter, &tid, &pos, &n_plp)) != NULL) {
            if((uint32_t)pos < localPos || (uint32_t)pos >= localEnd) continue;
            if(config->bed) {
                while((o = posOverlapsBED(tid, pos, config->bed, bedIdx)) == -1) bedIdx++;
                if(o == 0) continue;
            }
            if(isCpG(seq, pos - localPos, seqlen)) { if(!config->keepCpG) continue; }
            else if(isCHG(seq, pos - localPos, seqlen)) { if(!config->keepCHG) continue; }
            else if(isCHH(seq, pos - localPos, seqlen)) { if(!config->keepCHH) continue; }
            else continue;
            base = seq[pos - localPos];
            for(i = 0; i < n_plp; i++) {
                if(plp[i].is_del) continue;
                if(plp[i].is_refskip) continue;
                if(config->bed) if(!readStrandOverlapsBED(plp[i].b->r, config->bed->region[bedIdx])) continue;
                strand = getStrand(plp[i].b->r);
                if(strand & 1) { if(base != 'C' && base != 'c') continue; }
                else { if(base != 'G' && base != 'g') continue; }
                rv = methState(config, plp[i].b, plp[i].qpos);
                if(rv != 0) {
                    int32_t qpos = plp[i].qpos; strandMeth *sm;
                    if(qpos >= meths[strand - 1]->m) meths[strand - 1] = growStrandMeth(meths[strand - 1], qpos);
                    sm = meths[strand - 1];
                    if(rv < 0) { if(plp[i].b->r->flag & 0x80) sm->unmeth2[qpos]++; else sm->unmeth1[qpos]++; }
                    else { if(plp[i].b->r->flag & 0x80) sm->meth2[qpos]++; else sm->meth1[qpos]++; }
                    if(qpos + 1 > sm->l) sm->l = qpos + 1;
                }
            }
        }
        free(seq);
        plp_destroy(&iter);
    }
    return meths;
}

/* svg.c:8-27  Agresti-Coull interval */
static double CI(uint32_t um, uint32_t m, int which) {
    double ZZ, Z, N_dot, P_dot, X, N, rv;
    X = (double)m; N = (double)(m + um);
    ZZ = 10.8275661707; Z = 3.2905267315;
    N_dot = N + ZZ;
    P_dot = (1.0 / N_dot) * (X + 0.5 * ZZ);
    if(which) { rv = P_dot + Z * sqrt((P_dot / N_dot) * (1 - P_dot)); if(rv > 1.) rv = 1.0; }
    else { rv = P_dot - Z * sqrt((P_dot / N_dot) * (1 - P_dot)); if(rv < 0.) rv = 0.0; }
    return rv;
}
static double getMaxY(strandMeth *m) {   /* svg.c:29-55 */
    double maximum = 0.0, val; int i;
    for(i = 0; i < m->l; i++) {
        if(m->meth1[i] + m->unmeth1[i]) { val = CI(m->unmeth1[i], m->meth1[i], 1); maximum = (val > maximum) ? val : maximum; }
        if(m->meth2[i] + m->unmeth2[i]) { val = CI(m->unmeth2[i], m->meth2[i], 1); maximum = (val > maximum) ? val : maximum; }
    }
    maximum += 0.03;
    if(5 * (((int)ceil(100 * maximum)) / 5) - (int)ceil(100 * maximum)) maximum = (1 + ((int)ceil(100 * maximum)) / 5) * 0.05;
    else maximum = ((int)ceil(100 * maximum) / 5) * 0.05;
    if(maximum > 0.8) maximum = 1.0;
    assert(maximum > 0.0);
    return maximum;
}
static double getMinY(strandMeth *m) {   /* svg.c:57-79 */
    double minimum = 1.0, val; int i;
    for(i = 0; i < m->l; i++) {
        if(m->meth1[i] + m->unmeth1[i]) { val = CI(m->unmeth1[i], m->meth1[i], 0); minimum = (val < minimum) ? val : minimum; }
        if(m->meth2[i] + m->unmeth2[i]) { val = CI(m->unmeth2[i], m->meth2[i], 0); minimum = (val < minimum) ? val : minimum; }
    }
    minimum -= 0.03;
    minimum = 0.01 * (5 * (((int)(100 * minimum)) / 5));
    if(minimum < 0.2) minimum = 0.0;
    assert(minimum < 1.0);
    return minimum;
}
static int getMinX(strandMeth *m, int which) {   /* svg.c:81-91 */
    int i;
    for(i = 0; i < m->l; i++) {
        if(which == 1) { if(m->unmeth1[i] + m->meth1[i]) return i; }
        else { if(m->unmeth2[i] + m->meth2[i]) return i; }
    }
    return m->l;
}
static int getMaxX(strandMeth *m) {   /* svg.c:93-107; its i>=0 test is always true */
    int i;
    for(i = m->l; i > 0; i--) {
        if(m->unmeth1[i - 1] + m->meth1[i - 1]) break;
        if(m->unmeth2[i - 1] + m->meth2[i - 1]) break;
    }
    if(i % 5) i += 5 - (i % 5);
    return i;
}
static int *getXTicks(int maxX, int *n) {   /* svg.c:109-149: every else-if repeats the first test, so the span is 5 or 10 */
    int *o, maxN = 7, i, span = 5;
    *n = maxX / 5;
    if(*n > maxN) { span = 10; *n = maxX / span; }
    o = xmalloc((size_t)(*n) * sizeof(int));
    for(i = 0; i < *n; i++) o[i] = (i + 1) * span;
    return o;
}
static double *getYTicks(double minY, double maxY, int *n) {   /* svg.c:151-164 */
    double *o, span = maxY - minY; int i;
    *n = (int)(1 + ceil(span / 0.05));
    if(span < 0.05) *n = 2;
    o = xmalloc(sizeof(double) * (size_t)(*n));
    for(i = 0; i < *n; i++) o[i] = 0.05 * i + minY;
    return o;
}
static double remapY(double orig, double minY, double maxY, int buffer, int dim) { return buffer + dim - ((double)dim) * (orig - minY) / (maxY - minY); }   /* svg.c:166-168 */
static double remapX(int orig, int maxX, int buffer, int dim) { return buffer + ((double)dim) * orig / ((double)maxX); }   /* svg.c:170-172 */

/* svg.c:174-203.  The loops run to i == m->l inclusive; in the reference that element exists and is zero (the merged
 * arrays are grown past the per-thread capacity, MBias.c:45), which is what the arrays here guarantee too. */
static void plotCI(FILE *of, int minX, int maxX, strandMeth *m, int which, const char *col, int buffer, int dim, double minY, double maxY) {
    uint32_t *meth, *umeth; int32_t i; double val;
    if(which == 1) { meth = m->meth1; umeth = m->unmeth1; } else { meth = m->meth2; umeth = m->unmeth2; }
    val = CI(umeth[minX], meth[minX], 0);
    fprintf(of, "<path d=\"M %f %f\n", remapX(minX + 1, maxX, buffer, dim), remapY(val, minY, maxY, buffer, dim));
    for(i = minX + 1; i <= m->l; i++) {
        if(meth[i] || umeth[i]) { val = CI(umeth[i], meth[i], 0); fprintf(of, "  L %f %f\n", remapX(i + 1, maxX, buffer, dim), remapY(val, minY, maxY, buffer, dim)); }
    }
    for(i = m->l - 1; i >= 0; i--) {
        if(meth[i] || umeth[i]) { val = CI(umeth[i], meth[i], 1); fprintf(of, "  L %f %f\n", remapX(i + 1, maxX, buffer, dim), remapY(val, minY, maxY, buffer, dim)); }
    }
    fprintf(of, "Z\" fill=\"%s\" fill-opacity=\"0.2\"/>\n", col);
}
static void plotVals(FILE *of, int minX, int maxX, strandMeth *m, int which, const char *col, int buffer, int dim, double minY, double maxY) {   /* svg.c:205-228 */
    uint32_t *meth, *umeth; int32_t i; double val;
    if(which == 1) { meth = m->meth1; umeth = m->unmeth1; } else { meth = m->meth2; umeth = m->unmeth2; }
    assert(minX >= 0);
    val = meth[minX] / ((double)(meth[minX] + umeth[minX]));
    fprintf(of, "<path d=\"M %f %f\n", remapX(minX + 1, maxX, buffer, dim), remapY(val, minY, maxY, buffer, dim));
    for(i = minX + 1; i <= m->l; i++) {
        if(meth[i] || umeth[i]) { val = meth[i] / ((double)(meth[i] + umeth[i])); fprintf(of, "  L %f %f\n", remapX(i + 1, maxX, buffer, dim), remapY(val, minY, maxY, buffer, dim)); }
    }
    fprintf(of, "\" stroke=\"%s\" stroke-width=\"2\" fill-opacity=\"0\"/>\n", col);
}
static void getThresholds(strandMeth *m, int which, int *lthresh, int *rthresh) {   /* svg.c:239-294 */
    uint32_t *meth, *umeth; int i, total = 0, middle = m->l / 2;
    double average = 0.0, minCI = 1.0, maxCI = 0.0, tmp, tmp2;
    if(which == 1) { meth = m->meth1; umeth = m->unmeth1; } else { meth = m->meth2; umeth = m->unmeth2; }
    for(i = (int)(0.2 * m->l); i <= (int)(0.8 * m->l); i++) {
        if(meth[i] || umeth[i]) {
            total++;
            average += ((double)meth[i]) / ((double)(meth[i] + umeth[i]));
            tmp = CI(umeth[i], meth[i], 1); if(minCI > tmp) minCI = tmp;
            tmp = CI(umeth[i], meth[i], 0); if(maxCI < tmp) maxCI = tmp;
        }
    }
    if(total) average /= total;
    else { *lthresh = 0; *rthresh = 0; return; }
    for(i = middle; i >= 0; i--) {
        if(meth[i] || umeth[i]) {
            tmp = ((double)meth[i]) / ((double)(meth[i] + umeth[i]));
            tmp2 = CI(umeth[i], meth[i], 1);
            if(tmp2 < average && tmp < minCI && fabs(tmp - average) > 0.05) break;
            tmp2 = CI(umeth[i], meth[i], 0);
            if(tmp2 > average && tmp > maxCI && fabs(tmp - average) > 0.05) break;
        }
    }
    if(i >= 0) *lthresh = i + 2; else *lthresh = 0;
    for(i = middle + 1; i < m->l; i++) {
        if(meth[i] || umeth[i]) {
            tmp = ((double)meth[i]) / ((double)(meth[i] + umeth[i]));
            tmp2 = CI(umeth[i], meth[i], 1);
            if(tmp2 < average && tmp < minCI && fabs(tmp - average) > 0.05) break;
            tmp2 = CI(umeth[i], meth[i], 0);
            if(tmp2 > average && tmp > maxCI && fabs(tmp - average) > 0.05) break;
        }
    }
    if(i < m->l) *rthresh = i; else *rthresh = 0;
}
static void makeSVGs(char *opref, strandMeth **meths, int which) {   /* svg.c:300-437 */
    double minY = 1.0, maxY = 0.0; int minX1 = -1, minX2 = -1, maxX = 0, hasRead1 = 0, hasRead2 = 0;
    int i, j, buffer = 80, dim = 500, nXTicks, nYTicks;
    char *oname = xmalloc(strlen(opref) + strlen("_CTOT.svg "));
    const char *titles[4] = {"Original Top", "Original Bottom", "Complementary to the Original Top", "Complementary to the Original Bottom"};
    const char *abbrevs[4] = {"OT", "OB", "CTOT", "CTOB"};
    const char *col1 = "rgb(248,118,109)", *col2 = "rgb(0,191,196)";
    FILE *of; double *yTicks; int *xTicks, lthresh1, lthresh2, rthresh1, rthresh2; int alreadyPrinting = 0, doingLabel = 0;
    for(i = 0; i < 4; i++) {
        if(meths[i]->l) {
            minY = getMinY(meths[i]); maxY = getMaxY(meths[i]);
            minX1 = getMinX(meths[i], 1); minX2 = getMinX(meths[i], 2);
            maxX = getMaxX(meths[i]);
            xTicks = getXTicks(maxX, &nXTicks); yTicks = getYTicks(minY, maxY, &nYTicks);
            sprintf(oname, "%s_%s.svg", opref, abbrevs[i]);
            of = fopen(oname, "w");
            if(!of) { fprintf(stderr, "mdk_oracle: cannot write %s\n", oname); exit(3); }
            fprintf(of, "<svg height=\"%i\" width=\"%i\"\n", dim + 2 * buffer, dim + 2 * buffer);
            fprintf(of, "    xmlns=\"http://www.w3.org/2000/svg\"\n");
            fprintf(of, "    xmlns:xlink=\"http://www.w3.org/1999/xlink\"\n");
            fprintf(of, "    xmlns:ev=\"http://www.w3.org/2001/xml-events\">\n");
            fprintf(of, "<title>%s Strand</title>\n", titles[i]);
            fprintf(of, "<rect x=\"0\" y=\"0\" width=\"%i\" height=\"%i\" fill=\"white\" />\n", dim + 2 * buffer, dim + 2 * buffer);
            fprintf(of, "<text x=\"%i\" y=\"%i\" text-anchor=\"middle\">%s Strand</text>\n", buffer + (dim >> 1), 20, titles[i]);
            fprintf(of, "<line x1=\"%i\" y1=\"%i\" x2=\"%i\" y2=\"%i\" stroke=\"black\" />\n", buffer, buffer, buffer, buffer + dim);
            fprintf(of, "<line x1=\"%i\" y1=\"%i\" x2=\"%i\" y2=\"%i\" stroke=\"black\" />\n", buffer, buffer + dim, buffer + dim, buffer + dim);
            fprintf(of, "<text x=\"15\" y=\"%i\" transform=\"rotate(270 15, %i)\" text-anchor=\"middle\" dominant-baseline=\"text-before-edge\">", buffer + (dim >> 1), buffer + (dim >> 1));
            doingLabel = 0;
            if(which & 1) { doingLabel = 1; fprintf(of, "CpG"); }
            if(which & 2) { if(doingLabel) fprintf(of, "/CHG"); else fprintf(of, "CHG"); doingLabel = 1; }
            if(which & 4) { if(doingLabel) fprintf(of, "/CHH"); else fprintf(of, "CHH"); doingLabel = 1; }
            if(doingLabel) fprintf(of, " ");
            fprintf(of, "Methylation %%</text>\n");
            fprintf(of, "<text x=\"%i\" y=\"%i\" text-anchor=\"middle\">Position along mapped read (5'->3' of + strand)</text>\n", buffer + (dim >> 1), buffer + dim + 40);
            fprintf(of, "<line x1=\"%i\" y1=\"%i\" x2=\"%i\" y2=\"%i\" stroke=\"black\" />\n", buffer, buffer + dim, buffer, buffer + dim + 5);
            fprintf(of, "<text x=\"%i\" y=\"%i\" text-anchor=\"middle\">%i</text>\n", buffer, buffer + dim + 20, 0);
            for(j = 0; j < nXTicks; j++) {
                fprintf(of, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 5\" stroke=\"grey\" />\n", remapX(xTicks[j], maxX, buffer, dim), buffer, remapX(xTicks[j], maxX, buffer, dim), buffer + dim);
                fprintf(of, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke=\"black\" />\n", remapX(xTicks[j], maxX, buffer, dim), buffer + dim, remapX(xTicks[j], maxX, buffer, dim), buffer + dim + 5);
                fprintf(of, "<text x=\"%f\" y=\"%i\" text-anchor=\"middle\">%i</text>\n", remapX(xTicks[j], maxX, buffer, dim), buffer + dim + 20, xTicks[j]);
            }
            for(j = 0; j < nYTicks; j++) {
                fprintf(of, "<line x1=\"%i\" y1=\"%f\" x2=\"%i\" y2=\"%f\" stroke=\"black\" />\n", buffer, remapY(yTicks[j], minY, maxY, buffer, dim), buffer - 5, remapY(yTicks[j], minY, maxY, buffer, dim));
                fprintf(of, "<text x=\"%i\" y=\"%f\" text-anchor=\"middle\" dominant-baseline=\"middle\">%4.2f</text>\n", buffer - 25, remapY(yTicks[j], minY, maxY, buffer, dim), yTicks[j]);
            }
            for(j = 0; j < meths[i]->l; j++) {
                if(meths[i]->unmeth1[j] + meths[i]->meth1[j]) hasRead1 = 1;
                if(meths[i]->unmeth2[j] + meths[i]->meth2[j]) hasRead2 = 1;
                if(hasRead1 && hasRead2) break;
            }
            if(hasRead1) plotCI(of, minX1, maxX, meths[i], 1, col1, buffer, dim, minY, maxY);
            if(hasRead2) plotCI(of, minX2, maxX, meths[i], 2, col2, buffer, dim, minY, maxY);
            if(hasRead1) plotVals(of, minX1, maxX, meths[i], 1, col1, buffer, dim, minY, maxY);
            if(hasRead2) plotVals(of, minX2, maxX, meths[i], 2, col2, buffer, dim, minY, maxY);
            getThresholds(meths[i], 1, &lthresh1, &rthresh1);
            getThresholds(meths[i], 2, &lthresh2, &rthresh2);
            if(lthresh1 + lthresh2 + rthresh1 + rthresh2) {
                fprintf(of, "<text x=\"%i\" y=\"%i\" text-anchor=\"end\">--%s %i,%i,%i,%i</text>\n", 2 * buffer + dim - 10, 2 * buffer + dim - 10, abbrevs[i], lthresh1, rthresh1, lthresh2, rthresh2);
                if(lthresh1) fprintf(of, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 1\" stroke=\"%s\" stroke-width=\"1\" />\n", remapX(lthresh1, maxX, buffer, dim), dim + buffer, remapX(lthresh1, maxX, buffer, dim), buffer, col1);
                if(rthresh1) fprintf(of, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 1\" stroke=\"%s\" stroke-width=\"1\" />\n", remapX(rthresh1, maxX, buffer, dim), dim + buffer, remapX(rthresh1, maxX, buffer, dim), buffer, col1);
                if(lthresh2) fprintf(of, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 1\" stroke=\"%s\" stroke-width=\"1\" />\n", remapX(lthresh2, maxX, buffer, dim), dim + buffer, remapX(lthresh2, maxX, buffer, dim), buffer, col2);
                if(rthresh2) fprintf(of, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 1\" stroke=\"%s\" stroke-width=\"1\" />\n", remapX(rthresh2, maxX, buffer, dim), dim + buffer, remapX(rthresh2, maxX, buffer, dim), buffer, col2);
            }
            if(hasRead1) {
                fprintf(of, "<rect x=\"%i\" y=\"%i\" width=\"20\" height=\"20\" fill=\"%s\" />\n", dim + buffer + 10, (dim >> 1) + buffer - 20, col1);
                fprintf(of, "<text x=\"%i\" y=\"%i\" text-anchor=\"start\" dominant-baseline=\"middle\">#1</text>\n", dim + buffer + 35, (dim >> 1) + buffer - 10);
            }
            if(hasRead2) {
                fprintf(of, "<rect x=\"%i\" y=\"%i\" width=\"20\" height=\"20\" fill=\"%s\" />\n", dim + buffer + 10, (dim >> 1) + buffer, col2);
                fprintf(of, "<text x=\"%i\" y=\"%i\" text-anchor=\"start\" dominant-baseline=\"middle\">#2</text>\n", dim + buffer + 35, (dim >> 1) + buffer + 10);
            }
            fprintf(of, "</svg>\n");
            if(!alreadyPrinting) fprintf(stderr, "Suggested inclusion options:");
            fprintf(stderr, " --%s %i,%i,%i,%i", abbrevs[i], lthresh1, rthresh1, lthresh2, rthresh2);
            alreadyPrinting = 1;
            fclose(of); free(xTicks); free(yTicks);
            hasRead1 = 0; hasRead2 = 0;
        }
    }
    if(alreadyPrinting) fprintf(stderr, "\n");
    free(oname);
}
static void makeTXT(strandMeth **m) {   /* svg.c:439-454 */
    const char *abbrevs[4] = {"OT", "OB", "CTOT", "CTOB"}; int i, j;
    printf("Strand\tRead\tPosition\tnMethylated\tnUnmethylated\n");
    for(i = 0; i < 4; i++) {
        if(m[i]->l) {
            for(j = 0; j < m[i]->l; j++) {
                if(m[i]->meth1[j] || m[i]->unmeth1[j]) printf("%s\t1\t%i\t%" PRIu32 "\t%" PRIu32 "\n", abbrevs[i], j + 1, m[i]->meth1[j], m[i]->unmeth1[j]);
                if(m[i]->meth2[j] || m[i]->unmeth2[j]) printf("%s\t2\t%i\t%" PRIu32 "\t%" PRIu32 "\n", abbrevs[i], j + 1, m[i]->meth2[j], m[i]->unmeth2[j]);
            }
        }
    }
}
static void mbias_usage(void) { fprintf(stderr, "\nUsage: mdk_oracle mbias [OPTIONS] <ref.fa> <sorted_alignments.bam> <output.prefix>\n"); }

static int mbias_main(int argc, char *argv[]) {                /* MBias.c:308-573 */
    char *opref = NULL, *bedName = NULL; int c, i, j, SVG = 1, txt = 0, keepStrand = 0;
    strandMeth *meths[4], **threadout = NULL; Config config; bamfile bf; fasta fa;
    static struct option lopts[] = {
        {"noCpG", 0, NULL, 1}, {"CHG", 0, NULL, 2}, {"CHH", 0, NULL, 3}, {"keepDupes", 0, NULL, 4}, {"keepSingleton", 0, NULL, 5},
        {"keepDiscordant", 0, NULL, 6}, {"txt", 0, NULL, 7}, {"noSVG", 0, NULL, 8}, {"nOT", 1, NULL, 9}, {"nOB", 1, NULL, 10},
        {"nCTOT", 1, NULL, 11}, {"nCTOB", 1, NULL, 12}, {"chunkSize", 1, NULL, 13}, {"keepStrand", 0, NULL, 14},
        {"minConversionEfficiency", 1, NULL, 15}, {"ignoreNH", 0, NULL, 16}, {"ignoreFlags", 1, NULL, 'F'}, {"requireFlags", 1, NULL, 'R'},
        {"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'}, {0, 0, NULL, 0}};
    memset(&config, 0, sizeof(config));
    config.keepCpG = 1; config.keepCHG = 0; config.keepCHH = 0;
    config.minMapq = 10; config.minPhred = 5; config.keepDupes = 0;
    config.keepSingleton = 0; config.keepDiscordant = 0;
    config.filterMappability = 0; config.ignoreNH = 0;
    config.reg = NULL; config.bed = NULL;
    config.ignoreFlags = 0xF00; config.requireFlags = 0;
    config.nThreads = 1; config.chunkSize = 1000000; config.minConversionEfficiency = 0.0;
    for(i = 0; i < 16; i++) config.bounds[i] = 0;
    for(i = 0; i < 16; i++) config.absoluteBounds[i] = 0;
    optind = 1;
    while((c = getopt_long(argc, argv, "hvq:p:r:l:D:F:@:", lopts, NULL)) >= 0) {
        switch(c) {
        case 'h': mbias_usage(); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", ORACLE_VERSION, "none: mdk_oracle"); return 0;
        case 'D': break;
        case 'r': config.reg = optarg; break;
        case 'l': bedName = optarg; break;
        case 1: config.keepCpG = 0; break;
        case 2: config.keepCHG = 1; break;
        case 3: config.keepCHH = 1; break;
        case 4: config.keepDupes = 1; break;
        case 5: config.keepSingleton = 1; break;
        case 6: config.keepDiscordant = 1; break;
        case 7: txt = 1; break;
        case 8: SVG = 0; txt = 1; break;
        case 9: parseBounds(optarg, config.absoluteBounds, 0); break;
        case 10: parseBounds(optarg, config.absoluteBounds, 1); break;
        case 11: parseBounds(optarg, config.absoluteBounds, 2); break;
        case 12: parseBounds(optarg, config.absoluteBounds, 3); break;
        case 13:
            config.chunkSize = strtoul(optarg, NULL, 10);
            if(config.chunkSize < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); return 1; }
            break;
        case 14: keepStrand = 1; break;
        case 15: config.minConversionEfficiency = (float)atof(optarg); break;
        case 16: config.ignoreNH = 1; break;
        case 'F': config.ignoreFlags = atoi(optarg); break;
        case 'R': config.requireFlags = atoi(optarg); break;
        case 'q': config.minMapq = atoi(optarg); break;
        case 'p': config.minPhred = atoi(optarg); break;
        case '@': config.nThreads = atoi(optarg); break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); mbias_usage(); return 1;
        }
    }
    if(argc == 1) { mbias_usage(); return 0; }
    if((SVG && argc - optind != 3) || (!SVG && argc - optind < 2)) {
        fprintf(stderr, "You must supply a reference genome in fasta format, an input BAM file, and an output prefix!!!\n");
        mbias_usage(); return -1;
    }
    if(config.minPhred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", config.minPhred); config.minPhred = 1; }
    if(config.minMapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", config.minMapq); config.minMapq = 0; }
    if(!(config.keepCpG + config.keepCHG + config.keepCHH)) {
        fprintf(stderr, "You haven't specified any metrics to output!\nEither don't use the --noCpG option or specify --CHG and/or --CHH.\n");
        return -1;
    }
    if(bam_load(argv[optind + 1], &bf) != 0) { fprintf(stderr, "Couldn't open %s for reading!\n", argv[optind + 1]); return -4; }
    if(fasta_load(argv[optind], &fa) != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", argv[optind]); return -4; }
    if(SVG) opref = argv[optind + 2];
    globalTid = 0; globalPos = 0; globalEnd = 0;
    if(config.reg) {
        const char *foo; char *bar; int s = 0, e = 0;
        foo = parse_reg(config.reg, &s, &e);
        if(foo == NULL) { fprintf(stderr, "Could not parse the specified region!\n"); return -4; }
        bar = xmalloc((size_t)(foo - config.reg) + 1);
        strncpy(bar, config.reg, (size_t)(foo - config.reg)); bar[foo - config.reg] = 0;
        globalTid = (uint32_t)-1;
        for(i = 0; i < bf.n_targets; i++) if(!strcmp(bf.target_name[i], bar)) { globalTid = (uint32_t)i; break; }
        if(globalTid == (uint32_t)-1) { fprintf(stderr, "%s did not match a known chromosome/contig name!\n", config.reg); return -6; }
        if(s > 0) globalPos = (uint32_t)s;
        if(e > 0) globalEnd = (uint32_t)e;
        if(globalEnd > bf.target_len[globalTid]) globalEnd = bf.target_len[globalTid];
        free(bar);
    }
    if(bedName) {
        config.bed = parseBED(bedName, &bf, keepStrand);
        if(!config.bed) { fprintf(stderr, "There was an error while reading in your BED file!\n"); return 1; }
    }
    for(i = 0; i < 4; i++) meths[i] = calloc(1, sizeof(strandMeth));
    threadout = extractMBias(&config, &bf, &fa);           /* one worker */
    for(j = 0; j < 4; j++) {
        meths[j] = mergeStrandMeth(meths[j], threadout[j]);
        free(threadout[j]->meth1); free(threadout[j]->unmeth1); free(threadout[j]->meth2); free(threadout[j]->unmeth2); free(threadout[j]);
    }
    free(threadout);
    if(SVG) makeSVGs(opref, meths, config.keepCpG + 2 * config.keepCHG + 4 * config.keepCHH);
    if(txt) makeTXT(meths);
    for(i = 0; i < 4; i++) { free(meths[i]->meth1); free(meths[i]->unmeth1); free(meths[i]->meth2); free(meths[i]->unmeth2); free(meths[i]); }
    return 0;
}



/* ------------------------------------------------------------------------------------------ */
/* perRead.c restated (`perRead`).  As for mbias, the reference's tests never run it: parity for   */
/* perRead is UNPINNED by the reference and rests on this restatement.                          */
/* ------------------------------------------------------------------------------------------ */
static void addRead(kstr *os, const brec *b, const bamfile *hdr, uint32_t nmethyl, uint32_t nunmethyl) {   /* perRead.c:16-36 */
    char str[10000];
    if(nmethyl + nunmethyl > 0) snprintf(str, 10000, "%s\t%s\t%" PRId64 "\t%f\t%" PRIu32 "\n", b->qname, hdr->target_name[b->tid], (int64_t)b->pos, 100. * ((double)nmethyl) / (nmethyl + nunmethyl), nmethyl + nunmethyl);
    else snprintf(str, 10000, "%s\t%s\t%" PRId64 "\t0.0\t%" PRIu32 "\n", b->qname, hdr->target_name[b->tid], (int64_t)b->pos, nmethyl + nunmethyl);
    kputs_(os, str);
}
static int cigar_type(uint32_t op) { static const int t[16] = {3, 1, 2, 2, 1, 0, 0, 3, 3, 0, 0, 0, 0, 0, 0, 0}; return t[op & 15]; }   /* htslib bam_cigar_type: MIDNSHP=XB */
/* processRead (perRead.c:38-94), step for step: after a low-quality base it moves one base on and evaluates that one
 * without looking at its quality or at the CIGAR again.  Two reads past the record's arrays are possible in the
 * reference and are resolved here by the BAM record layout, which is what the reference would see: the base at index
 * l_qseq is the padding nibble of the last sequence byte (odd l_qseq) or the high nibble of the first quality byte
 * (even l_qseq); a CIGAR index of n_cigar (only reachable on malformed records) ends the walk. */
static void processRead(Config *config, const brec *b, char *seq, uint32_t sequenceStart, int seqLen, uint32_t *nmethyl, uint32_t *nunmethyl) {
    uint32_t readPosition = 0, mappedPosition = (uint32_t)b->pos;
    int cigarOPNumber = 0, cigarOPOffset = 0;
    const uint8_t *readSeq = b->seq, *readQual = b->qual;
    int strand = getStrand(b), cigarOPType, direction, base;
    while(readPosition < (uint32_t)b->l_qseq && cigarOPNumber < b->n_cigar) {
        if(cigarOPOffset >= (int)cig_len(b->cigar, cigarOPNumber)) { cigarOPOffset = 0; cigarOPNumber++; }
        if(cigarOPNumber >= b->n_cigar) break;
        cigarOPType = cigar_type(cig_op(b->cigar, cigarOPNumber));
        if(cigarOPType & 2) {
            if(cigarOPType & 1) {
                if(readQual[readPosition] < config->minPhred) { mappedPosition++; readPosition++; cigarOPOffset++; }
                direction = seq ? isCpG(seq, (int)(mappedPosition - sequenceStart), seqLen) : 0;
                if(direction) {
                    if(readPosition < (uint32_t)b->l_qseq) base = (readSeq[readPosition >> 1] >> ((~readPosition & 1) << 2)) & 0xf;
                    else if(b->l_qseq & 1) base = readSeq[readPosition >> 1] & 0xf;
                    else base = b->l_qseq ? (readQual[0] >> 4) & 0xf : 0;
                    if(direction == 1 && (strand & 1) == 1) { if(base == 2) (*nmethyl)++; else if(base == 8) (*nunmethyl)++; }
                    else if(direction == -1 && (strand & 1) == 0) { if(base == 4) (*nmethyl)++; else if(base == 1) (*nunmethyl)++; }
                }
                mappedPosition++; readPosition++; cigarOPOffset++;
            } else { mappedPosition += cig_len(b->cigar, cigarOPNumber++); cigarOPOffset = 0; continue; }
        } else if(cigarOPType & 1) { readPosition += cig_len(b->cigar, cigarOPNumber++); cigarOPOffset = 0; continue; }
        else { cigarOPOffset = 0; cigarOPNumber++; continue; }
    }
}
static void perReadMetrics(Config *config, const bamfile *bf, const fasta *fa) {   /* perRead.c:96-223, one worker */
    int32_t bedIdx = 0; int seqlen; uint32_t nmethyl = 0, nunmethyl = 0;
    uint32_t localPos = 0, localEnd = 0, localTid = 0, localPos2 = 0; char *seq = NULL; kstr os_; kstr *os = &os_; regitr iter; const brec *b;
    memset(&os_, 0, sizeof(os_)); os_.m = 1024; os_.s = xmalloc(1024); os_.s[0] = 0;
    while(1) {
        bin_++;
        localTid = globalTid; localPos = globalPos;
        localEnd = (uint32_t)(localPos + config->chunkSize);
        if(localTid >= (uint32_t)bf->n_targets) break;
        if(globalEnd && localEnd > globalEnd) localEnd = globalEnd;
        globalPos = localEnd;
        if(globalEnd > 0 && globalPos >= globalEnd) globalTid = (uint32_t)-1;
        if(localTid < (uint32_t)bf->n_targets && globalTid != (uint32_t)-1) {
            if(globalPos >= bf->target_len[localTid]) { localEnd = bf->target_len[localTid]; globalTid++; globalPos = 0; }
        }
        if(config->bed) { if(spanOverlapsBED((int32_t)localTid, (int32_t)localPos, (int32_t)localEnd, config->bed, &bedIdx) != 1) continue; }
        localPos2 = 0; if(localPos > 1) localPos2 = localPos - 2;
        if(localTid >= (uint32_t)bf->n_targets) break;
        if(globalEnd && localPos >= globalEnd) break;
        regitr_init(&iter, bf, (int32_t)localTid, (int32_t)localPos, (int32_t)localEnd);
        seq = fetch_seq(fa, bf->target_name[localTid], (int)localPos2, (int)(localEnd + 10000), &seqlen);
        while((b = regitr_next(&iter)) != NULL) {
            if((uint32_t)b->pos < localPos) continue;
            if((uint32_t)b->pos >= localEnd) break;
            nmethyl = 0; nunmethyl = 0;
            if(config->requireFlags && (config->requireFlags & b->flag) != config->requireFlags) continue;
            if(config->ignoreFlags && (config->ignoreFlags & b->flag) != 0) continue;
            if(b->mapq < config->minMapq) continue;
            processRead(config, b, seq, localPos2, seqlen, &nmethyl, &nunmethyl);
            addRead(os, b, bf, nmethyl, nunmethyl);
        }
        free(seq);
        if(os->l) fputs(os->s, config->output_fp[0]);
        os->l = 0; os->s[0] = 0;
    }
    free(os_.s);
}
static void perRead_usage(void) { fprintf(stderr, "\nUsage: mdk_oracle perRead [OPTIONS] <ref.fa> <input>\n"); }
static int perRead_main(int argc, char *argv[]) {              /* perRead.c:275-464 */
    Config config; FILE *ofile = stdout; int i, keepStrand = 0; char *bedName = NULL; bamfile bf; fasta fa; char c;
    static struct option lopts[] = {{"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'}, {"chunkSize", 1, NULL, 19}, {"keepStrand", 0, NULL, 20},
                                    {"ignoreFlags", 1, NULL, 'F'}, {"requireFlags", 1, NULL, 'R'}, {0, 0, NULL, 0}};
    memset(&config, 0, sizeof(config));
    config.keepCpG = 1; config.minMapq = 10; config.minPhred = 5; config.ignoreFlags = 0; config.requireFlags = 0; config.nThreads = 1; config.chunkSize = 1000000;
    optind = 1;
    while((c = (char)getopt_long(argc, argv, "hvq:p:o:@:r:l:F:R:", lopts, NULL)) >= 0) {
        switch(c) {
        case 'h': perRead_usage(); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", ORACLE_VERSION, "none: mdk_oracle"); return 0;
        case 'o': if((ofile = fopen(optarg, "w")) == NULL) { fprintf(stderr, "Couldn't open %s for writing\n", optarg); return 2; } break;
        case 'q': config.minMapq = atoi(optarg); break;
        case 'p': config.minPhred = atoi(optarg); break;
        case '@': config.nThreads = atoi(optarg); break;
        case 'r': config.reg = optarg; break;
        case 'l': bedName = optarg; break;
        case 'F': config.ignoreFlags = atoi(optarg); break;
        case 'R': config.requireFlags = atoi(optarg); break;
        case 19:
            config.chunkSize = strtoul(optarg, NULL, 10);
            if(config.chunkSize < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); return 1; }
            break;
        case 20: keepStrand = 1; break;
        case 21: config.ignoreNH = 1; break;     /* unreachable: --ignoreNH is in the help text but not in lopts (perRead.c:300-308) */
        default: fprintf(stderr, "Invalid option '%c'\n", c); perRead_usage(); return 1;
        }
    }
    if(argc == 1) { perRead_usage(); return 0; }
    if(argc - optind != 2) { fprintf(stderr, "You must supply a reference genome in fasta format and a BAM or CRAM file\n"); perRead_usage(); return -1; }
    if(config.minPhred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", config.minPhred); config.minPhred = 1; }
    if(config.minMapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", config.minMapq); config.minMapq = 0; }
    if(fasta_load(argv[optind], &fa) != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", argv[optind]); perRead_usage(); return -2; }
    if(bam_load(argv[optind + 1], &bf) != 0) { fprintf(stderr, "Couldn't open %s for reading!\n", argv[optind + 1]); return -4; }
    config.output_fp[0] = ofile;
    globalTid = 0; globalPos = 0; globalEnd = 0; bin_ = 0;
    if(config.reg) {
        const char *foo; char *bar; int s = 0, e = 0;
        foo = parse_reg(config.reg, &s, &e);
        if(foo == NULL) { fprintf(stderr, "Could not parse the specified region!\n"); return -4; }
        bar = xmalloc((size_t)(foo - config.reg) + 1);
        strncpy(bar, config.reg, (size_t)(foo - config.reg)); bar[foo - config.reg] = 0;
        globalTid = (uint32_t)-1;
        for(i = 0; i < bf.n_targets; i++) if(!strcmp(bf.target_name[i], bar)) { globalTid = (uint32_t)i; break; }
        if(globalTid == (uint32_t)-1) { fprintf(stderr, "%s did not match a known chromosome/contig name!\n", config.reg); return -6; }
        if(s > 0) globalPos = (uint32_t)s;
        if(e > 0) globalEnd = (uint32_t)e;
        if(globalEnd > bf.target_len[globalTid]) globalEnd = bf.target_len[globalTid];
        free(bar);
    }
    if(bedName) {
        config.bed = parseBED(bedName, &bf, keepStrand);
        if(!config.bed) { fprintf(stderr, "There was an error while reading in your BED file!\n"); return 1; }
    }
    perReadMetrics(&config, &bf, &fa);
    if(ofile != stdout) fclose(ofile);
    return 0;
}


/* ------------------------------------------------------------------------------------------ */
/* mergeContext.c restated (`mergeContext`, a text-to-text tool; unpinned by the reference's tests) */
/* ------------------------------------------------------------------------------------------ */
struct mcLast { char *chrom; int32_t start, end; uint32_t nmethyl, nunmethyl; };
static void printRecord(FILE *of, char *chr, int32_t start, int32_t end, uint32_t nmethyl, uint32_t nunmethyl) {   /* mergeContext.c:24-28 */
    fprintf(of, "%s\t%" PRId32 "\t%" PRId32 "\t%i\t%" PRIu32 "\t%" PRIu32 "\n", chr, start, end, (int)(100.0 * ((double)nmethyl) / (nmethyl + nunmethyl)), nmethyl, nunmethyl);
}
static void MergeOrPrint(FILE *of, struct mcLast *last, char *chr, int32_t start, int32_t width, uint32_t nmethyl, uint32_t nunmethyl) {   /* mergeContext.c:30-56 */
    int32_t end;
    if(width > 0) end = start + width;
    else { end = start + 1; start = end + width; }
    if(last->chrom && strcmp(last->chrom, chr) == 0 && last->start == start && last->end == end) {
        printRecord(of, chr, start, end, nmethyl + last->nmethyl, nunmethyl + last->nunmethyl);
        free(last->chrom); free(chr); last->chrom = NULL;
    } else {
        if(last->chrom) { printRecord(of, last->chrom, last->start, last->end, last->nmethyl, last->nunmethyl); free(last->chrom); }
        last->chrom = chr; last->start = start; last->end = end; last->nmethyl = nmethyl; last->nunmethyl = nunmethyl;
    }
}
static int getContext(const fasta *fa, char *chr, int32_t pos, int *width) {   /* mergeContext.c:58-97 */
    int len = -1, rv = 2, k; int32_t start, end; char *seq, *base;
    for(k = 0; k < fa->n; k++) if(!strcmp(fa->name[k], chr)) { len = (int)fa->len[k]; break; }      /* faidx_seq_len: -1 if unknown */
    start = (pos > 2) ? pos - 2 : 0;
    end = (pos + 2 < len) ? pos + 2 : len - 1;
    seq = fetch_seq(fa, chr, start, end, &len);
    if(!seq) return 3;
    base = seq + (pos - start);
    if(toupper((unsigned char)*base) == 'C') {
        if(end - pos) {
            if(toupper((unsigned char)*(base + 1)) == 'G') { *width = 2; rv = 0; }
            else if(end - pos == 2) { if(toupper((unsigned char)*(base + 2)) == 'G') { *width = 3; rv = 1; } }
        }
    } else {
        assert(toupper((unsigned char)*base) == 'G');
        if(pos - start) {
            if(toupper((unsigned char)*(base - 1)) == 'C') { *width = -2; rv = 0; }
            else if(pos - start == 2) { if(toupper((unsigned char)*(base - 2)) == 'C') { *width = -3; rv = 1; } }
        }
    }
    free(seq);
    return rv;
}
static void mergeContext(FILE *ifile, const fasta *fa, FILE *ofile) {   /* mergeContext.c:99-160 */
    struct mcLast *lastCpG = calloc(1, sizeof(*lastCpG)), *lastCHG = calloc(1, sizeof(*lastCHG));
    char *p, *p2, *chr, *line = NULL; size_t cap = 0; ssize_t got; int type, width = 0; int32_t start, end; uint32_t nmethyl, nunmethyl;
    while((got = getline(&line, &cap, ifile)) >= 0) {       /* ks_getuntil(KS_SEP_LINE): the line without its terminator */
        if(got && line[got - 1] == '\n') line[--got] = 0;
        if(got > 1 && line[got - 1] == '\r') line[--got] = 0;
        assert(got > 0);
        if(strncmp(line, "track", 5) == 0) continue;
        p = strtok(line, "\t"); chr = strdup(p);
        p = strtok(NULL, "\t"); start = (int32_t)strtoll(p, &p2, 10); assert(p2 != p);
        p = strtok(NULL, "\t"); end = (int32_t)strtoll(p, &p2, 10); assert(p2 != p);
        p = strtok(NULL, "\t");
        p = strtok(NULL, "\t"); nmethyl = (uint32_t)strtoul(p, &p2, 10); assert(p2 != p);
        p = strtok(NULL, "\n"); nunmethyl = (uint32_t)strtoul(p, &p2, 10); assert(p2 != p);
        type = getContext(fa, chr, start, &width);
        if(type == 0) MergeOrPrint(ofile, lastCpG, chr, start, width, nmethyl, nunmethyl);
        else if(type == 1) MergeOrPrint(ofile, lastCHG, chr, start, width, nmethyl, nunmethyl);
        else if(type == 2) { printRecord(ofile, chr, start, end, nmethyl, nunmethyl); free(chr); }
        else { fprintf(stderr, "[mergeContext] Error, %s is an unknown chromosome name!\n", chr); free(chr); break; }
    }
    if(lastCpG->chrom) { printRecord(ofile, lastCpG->chrom, lastCpG->start, lastCpG->end, lastCpG->nmethyl, lastCpG->nunmethyl); free(lastCpG->chrom); }
    if(lastCHG->chrom) { printRecord(ofile, lastCHG->chrom, lastCHG->start, lastCHG->end, lastCHG->nmethyl, lastCHG->nunmethyl); free(lastCHG->chrom); }
    free(line); free(lastCpG); free(lastCHG);
}
static void mergeContext_usage(void) { fprintf(stderr, "\nUsage: mdk_oracle mergeContext [OPTIONS] <ref.fa> <input>\n"); }
static int mergeContext_main(int argc, char *argv[]) {         /* mergeContext.c:179-236 */
    fasta fa; FILE *ifile, *ofile = stdout; char c;
    static struct option lopts[] = {{"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'}, {0, 0, NULL, 0}};
    optind = 1;
    while((c = (char)getopt_long(argc, argv, "hvo:", lopts, NULL)) >= 0) {
        switch(c) {
        case 'h': mergeContext_usage(); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", ORACLE_VERSION, "none: mdk_oracle"); return 0;
        case 'o': if((ofile = fopen(optarg, "w")) == NULL) { fprintf(stderr, "Couldn't open %s for writing\n", optarg); return 2; } break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); mergeContext_usage(); return 1;
        }
    }
    if(argc == 1) { mergeContext_usage(); return 0; }
    if(argc - optind != 2) { fprintf(stderr, "You must supply a reference genome in fasta format and an input bedGraph files\n"); mergeContext_usage(); return -1; }
    if(fasta_load(argv[optind], &fa) != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", argv[optind]); mergeContext_usage(); return -2; }
    if((ifile = fopen(argv[optind + 1], "r")) == NULL) { fprintf(stderr, "Couldn't open %s for reading!\n", argv[optind + 1]); return -3; }
    fprintf(ofile, "track type=\"bedGraph\" description=\"merged Methylation metrics\"\n");
    mergeContext(ifile, &fa, ofile);
    if(ofile != stdout) fclose(ofile);
    fclose(ifile);
    return 0;
}

/* test driver: `mdk_oracle mbias-report <prefix> <which>` plots a --txt table read from stdin (makeSVGs + makeTXT on
 * hand-made histograms; the arrays go through the same per-thread -> merged growth as in mbias_main) */
static int mbias_report_main(int argc, char *argv[]) {
    static const char *abbrevs[4] = {"OT", "OB", "CTOT", "CTOB"};
    strandMeth *meths[4], *src[4]; char line[256], name[16]; int i, r, q; unsigned m, u;
    if(argc != 3) { fprintf(stderr, "usage: mdk_oracle mbias-report <prefix> <which> < table\n"); return 1; }
    for(i = 0; i < 4; i++) { meths[i] = calloc(1, sizeof(strandMeth)); src[i] = calloc(1, sizeof(strandMeth)); }
    while(fgets(line, sizeof(line), stdin)) {
        if(sscanf(line, "%15s %d %d %u %u", name, &r, &q, &m, &u) != 5) continue;
        for(i = 0; i < 4; i++) if(!strcmp(name, abbrevs[i])) break;
        if(i == 4 || q < 1 || (r != 1 && r != 2)) continue;
        q--;
        if(q >= src[i]->m) src[i] = growStrandMeth(src[i], q);
        if(r == 1) { src[i]->meth1[q] = m; src[i]->unmeth1[q] = u; } else { src[i]->meth2[q] = m; src[i]->unmeth2[q] = u; }
        if(q + 1 > src[i]->l) src[i]->l = q + 1;
    }
    for(i = 0; i < 4; i++) meths[i] = mergeStrandMeth(meths[i], src[i]);
    makeSVGs(argv[1], meths, atoi(argv[2]));
    makeTXT(meths);
    return 0;
}

#ifndef MDK_ORACLE_NO_MAIN
int main(int argc, char *argv[]) {                             /* main.c:39-62 */
    if(argc == 1) { fprintf(stderr, "mdk_oracle: CPU oracle for `MethylDackel extract`\nUsage: mdk_oracle extract [options] ref.fa aln.bam\n"); return 0; }
    if(strcmp(argv[1], "-v") == 0 || strcmp(argv[1], "--version") == 0) { printf("%s (using HTSlib version %s)\n", ORACLE_VERSION, "none: mdk_oracle"); return 0; }
    if(getenv("MDK_ORACLE_PERTURB")) { g_pt = atoi(getenv("MDK_ORACLE_PERTURB")); if(g_pt < 0 || g_pt >= PT_N) g_pt = 0; }
    if(strcmp(argv[1], "extract") == 0) return extract_main(argc - 1, argv + 1);
    if(strcmp(argv[1], "mbias") == 0) return mbias_main(argc - 1, argv + 1);
    if(strcmp(argv[1], "perRead") == 0) return perRead_main(argc - 1, argv + 1);
    if(strcmp(argv[1], "mergeContext") == 0) return mergeContext_main(argc - 1, argv + 1);
    if(strcmp(argv[1], "mbias-report") == 0) return mbias_report_main(argc - 1, argv + 1);
    fprintf(stderr, "Unknown command!\n");
    return -1;
}
#endif

/* mdk_io.c -- BGZF/BAM streaming reader (parallel block inflate) and FASTA loader.  See mdk_io.h. */
#define _GNU_SOURCE
#include "mdk_io.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <time.h>
static double io_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

#define CCHUNK (24u << 20)     /* compressed bytes pulled per refill */

static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

typedef struct { const uint8_t *in; uint32_t in_len; uint8_t *out; uint32_t out_len; } blk_t;
typedef struct { blk_t *blk; int n; int next; int failed; pthread_mutex_t mu; } inflate_job;

static void *inflate_worker(void *arg) {
    inflate_job *job = arg; z_stream zs; int inited = 0;
    for(;;) {
        int i;
        pthread_mutex_lock(&job->mu); i = job->next; job->next += 8; pthread_mutex_unlock(&job->mu);
        if(i >= job->n) break;
        for(int k = i; k < i + 8 && k < job->n; k++) {
            blk_t *b = &job->blk[k];
            if(!b->out_len) continue;
            if(!inited) { memset(&zs, 0, sizeof(zs)); if(inflateInit2(&zs, -15) != Z_OK) { job->failed = 1; return NULL; } inited = 1; }
            else inflateReset(&zs);
            zs.next_in = (Bytef *)b->in; zs.avail_in = b->in_len; zs.next_out = b->out; zs.avail_out = b->out_len;
            if(inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) job->failed = 1;
        }
    }
    if(inited) inflateEnd(&zs);
    return NULL;
}

/* pull more compressed data, inflate every complete BGZF member in it, append to ubuf (after compacting) */
static int refill(mdk_bam *b) {
    size_t n, off = 0, total = 0; blk_t *blk = NULL; int nb = 0, mb = 0;
    if(b->uoff) { memmove(b->ubuf, b->ubuf + b->uoff, b->ulen - b->uoff); b->ulen -= b->uoff; b->uoff = 0; }
    if(!b->file_eof) {
        if(b->ccap < b->clen + CCHUNK) { b->ccap = b->clen + CCHUNK; b->cbuf = realloc(b->cbuf, b->ccap); if(!b->cbuf) return -1; }
        n = fread(b->cbuf + b->clen, 1, CCHUNK, b->f);
        b->clen += n;
        if(n < CCHUNK) b->file_eof = 1;
    }
    while(off + 18 <= b->clen) {
        const uint8_t *p = b->cbuf + off; uint16_t xlen; uint32_t bsize = 0, isize; size_t x; int have = 0;
        if(p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { snprintf(b->err, sizeof(b->err), "not a BGZF file (bad gzip member header)"); free(blk); return -2; }
        xlen = le16(p + 10);
        if(off + 12 + xlen > b->clen) break;
        for(x = 12; x + 4 <= 12u + xlen;) { uint16_t sl = le16(p + x + 2); if(p[x] == 'B' && p[x + 1] == 'C' && sl == 2) { bsize = le16(p + x + 4) + 1u; have = 1; } x += 4 + sl; }
        if(!have || bsize < 12u + xlen + 8u) { snprintf(b->err, sizeof(b->err), "BGZF member without a valid BC field"); free(blk); return -2; }
        if(off + bsize > b->clen) break;
        isize = le32(p + bsize - 4);
        if(nb == mb) { mb = mb ? mb * 2 : 1024; blk = realloc(blk, sizeof(blk_t) * mb); if(!blk) return -1; }
        blk[nb].in = p + 12 + xlen; blk[nb].in_len = bsize - 12 - xlen - 8; blk[nb].out = NULL; blk[nb].out_len = isize; nb++;
        total += isize; off += bsize;
    }
    if(b->ucap < b->ulen + total + 8) { b->ucap = b->ulen + total + (total >> 2) + 4096; b->ubuf = realloc(b->ubuf, b->ucap); if(!b->ubuf) { free(blk); return -1; } }
    { size_t o = b->ulen; for(int i = 0; i < nb; i++) { blk[i].out = b->ubuf + o; o += blk[i].out_len; } }
    if(nb) {
        inflate_job job; int nt = b->nthreads, i; pthread_t th[64];
        job.blk = blk; job.n = nb; job.next = 0; job.failed = 0; pthread_mutex_init(&job.mu, NULL);
        if(nt > 64) nt = 64; if(nt > (nb + 7) / 8) nt = (nb + 7) / 8; if(nt < 1) nt = 1;
        if(nt == 1) inflate_worker(&job);
        else { for(i = 0; i < nt; i++) pthread_create(&th[i], NULL, inflate_worker, &job); for(i = 0; i < nt; i++) pthread_join(th[i], NULL); }
        pthread_mutex_destroy(&job.mu);
        if(job.failed) { snprintf(b->err, sizeof(b->err), "BGZF inflate failed (corrupt file?)"); free(blk); return -2; }
    }
    b->ulen += total;
    memmove(b->cbuf, b->cbuf + off, b->clen - off); b->clen -= off;
    free(blk);
    if(nb == 0 && b->file_eof) { if(b->clen) { snprintf(b->err, sizeof(b->err), "truncated BGZF member at end of file"); return -2; } return 1; }
    return 0;
}

/* make at least n bytes available at uoff; 1 ok, 0 clean EOF (no bytes left), <0 error */
static int need(mdk_bam *b, size_t n) {
    while(b->ulen - b->uoff < n) {
        double t0 = io_now();
        int rc = refill(b);
        b->t_inflate += io_now() - t0;
        if(rc < 0) return rc;
        if(rc == 1) {
            if(b->ulen - b->uoff >= n) return 1;
            if(b->ulen == b->uoff) return 0;
            snprintf(b->err, sizeof(b->err), "truncated BAM record at end of file"); return -2;
        }
    }
    return 1;
}

mdk_bam *mdk_bam_open(const char *fn, int nthreads) {
    mdk_bam *b = calloc(1, sizeof(*b)); int rc; uint32_t i; size_t o;
    if(!b) return NULL;
    b->f = fopen(fn, "rb");
    if(!b->f) { free(b); return NULL; }
    b->nthreads = nthreads < 1 ? 1 : nthreads;
    if((rc = need(b, 12)) <= 0 || memcmp(b->ubuf + b->uoff, "BAM\1", 4)) { mdk_bam_close(b); return NULL; }
    b->l_text = le32(b->ubuf + b->uoff + 4);
    if(need(b, 12 + (size_t)b->l_text) <= 0) { mdk_bam_close(b); return NULL; }
    b->text = malloc((size_t)b->l_text + 1); memcpy(b->text, b->ubuf + b->uoff + 8, b->l_text); b->text[b->l_text] = 0;
    b->n_targets = (int32_t)le32(b->ubuf + b->uoff + 8 + b->l_text);
    b->uoff += 12 + (size_t)b->l_text;
    b->target_name = calloc((size_t)b->n_targets + 1, sizeof(char *)); b->target_len = calloc((size_t)b->n_targets + 1, sizeof(uint32_t));
    for(i = 0; i < (uint32_t)b->n_targets; i++) {
        uint32_t ln;
        if(need(b, 4) <= 0) { mdk_bam_close(b); return NULL; }
        ln = le32(b->ubuf + b->uoff);
        if(need(b, 8 + (size_t)ln) <= 0) { mdk_bam_close(b); return NULL; }
        o = b->uoff;
        b->target_name[i] = malloc((size_t)ln + 1); memcpy(b->target_name[i], b->ubuf + o + 4, ln); b->target_name[i][ln] = 0;
        b->target_len[i] = le32(b->ubuf + o + 4 + ln);
        b->uoff += 8 + (size_t)ln;
    }
    return b;
}

void mdk_bam_close(mdk_bam *b) {
    int i;
    if(!b) return;
    if(b->f) fclose(b->f);
    if(b->target_name) for(i = 0; i < b->n_targets; i++) free(b->target_name[i]);
    free(b->target_name); free(b->target_len); free(b->text); free(b->cbuf); free(b->ubuf); free(b);
}

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
    if(rc <= 0) return rc;
    bs = le32(b->ubuf + b->uoff);
    rc = need(b, 4 + (size_t)bs);
    if(rc <= 0) { if(rc == 0) { snprintf(b->err, sizeof(b->err), "truncated BAM record at end of file"); return -2; } return rc; }
    if(mdk_rec_parse(b->ubuf + b->uoff + 4, bs, r) != 0) { snprintf(b->err, sizeof(b->err), "malformed BAM record"); return -2; }
    return 1;
}
void mdk_bam_advance(mdk_bam *b, const mdk_rec *r) { b->uoff += 4 + (size_t)r->raw_len; b->n_records++; }

/* ---- FASTA ---- */
int mdk_fasta_load(const char *fn, mdk_fasta *fa) {
    FILE *f = fopen(fn, "rb"); size_t sz, i, w; char *d; int cur = -1, cap = 0;
    memset(fa, 0, sizeof(*fa));
    if(!f) return -1;
    fseek(f, 0, SEEK_END); sz = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    d = malloc(sz + 2);
    if(!d || fread(d, 1, sz, f) != sz) { fclose(f); free(d); return -1; }
    fclose(f); d[sz] = '\n'; d[sz + 1] = 0;
    fa->pool = d;
    /* in-place compaction: header lines become NUL-terminated names, sequence lines lose their whitespace */
    for(i = 0, w = 0; i < sz;) {
        char *nl = memchr(d + i, '\n', sz + 1 - i); size_t e = (size_t)(nl - d);
        if(d[i] == '>') {
            size_t s = i + 1, t = s;
            while(t < e && d[t] != ' ' && d[t] != '\t' && d[t] != '\r') t++;
            if(cur >= 0) fa->len[cur] = (int64_t)(d + w - fa->seq[cur]);
            if(fa->n == cap) { cap = cap ? cap * 2 : 64; fa->name = realloc(fa->name, sizeof(char *) * cap); fa->seq = realloc(fa->seq, sizeof(char *) * cap); fa->len = realloc(fa->len, sizeof(int64_t) * cap); }
            memmove(d + w, d + s, t - s); fa->name[fa->n] = d + w; w += t - s; d[w++] = 0;
            cur = fa->n++; fa->seq[cur] = d + w; fa->len[cur] = 0;
        } else if(cur >= 0) {
            size_t k;
            for(k = i; k < e; k++) { unsigned char c = (unsigned char)d[k]; if(c > ' ' && c <= '~') d[w++] = (char)c; }
        }
        i = e + 1;
    }
    if(cur >= 0) fa->len[cur] = (int64_t)(d + w - fa->seq[cur]);
    return 0;
}
void mdk_fasta_free(mdk_fasta *fa) { free(fa->pool); free(fa->name); free(fa->seq); free(fa->len); memset(fa, 0, sizeof(*fa)); }
int mdk_fasta_find(const mdk_fasta *fa, const char *name) { int i; for(i = 0; i < fa->n; i++) if(!strcmp(fa->name[i], name)) return i; return -1; }

/*
 * mdk_synth.c -- seeded synthetic WGBS data generator (FASTA + coordinate-sorted BAM [+ BBM]).
 *
 * Test/bench infrastructure: produces the inputs named by BASELINE.json's configs (SURVEY.md
 * section 8d): first-order Markov reference with depleted CpG, soft-masked runs and N runs;
 * directional paired-end bisulfite reads (OT pairs 99/147, OB pairs 83/163) with per-site
 * methylation levels, substitution errors, indels, soft clips, ref-skips, flag/MAPQ/NH noise,
 * optional Bismark-style XG tags (incl. CTOT/CTOB), optional secondary/supplementary records.
 * Everything is driven by xorshift64* streams so a (seed, args) pair is reproducible anywhere.
 *
 * usage: mdk_synth -o PREFIX [-L len[,len...]] [-c coverage] [-l readlen] [-s seed] [-z level]
 *                  [--bismark] [--extras] [--clean] [--bbm] [--single]
 * writes PREFIX.fa, PREFIX.bam (and PREFIX.bbm with --bbm) and prints totals as JSON on stdout.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <inttypes.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t rnd(rng_t *r) { uint64_t x = r->s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; r->s = x; return x * 2685821657736338717ULL; }
static double rndu(rng_t *r) { return (rnd(r) >> 11) * (1.0 / 9007199254740992.0); }
static int rndi(rng_t *r, int n) { return (int)(rndu(r) * n); }
static double rndn(rng_t *r) { double u = rndu(r), v = rndu(r); if(u < 1e-300) u = 1e-300; return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

typedef struct { uint8_t *p; size_t l, m; } buf_t;
static void bput(buf_t *b, const void *d, size_t n) { if(b->l + n > b->m) { b->m = (b->l + n) * 2 + 64; b->p = realloc(b->p, b->m); } memcpy(b->p + b->l, d, n); b->l += n; }
static void b32(buf_t *b, uint32_t v) { uint8_t x[4] = {v, v >> 8, v >> 16, v >> 24}; bput(b, x, 4); }
static void b16(buf_t *b, uint16_t v) { uint8_t x[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; bput(b, x, 2); }
static void b8(buf_t *b, uint8_t v) { bput(b, &v, 1); }

/* ---- BGZF writer ----
 * Members are cut exactly as a sequential writer would cut them (bgzf_write / bgzf_flush below), but only noted; bgzf_close compresses
 * them with a pool of threads and writes them in order.  The file is byte for byte what one thread would have written: a member's
 * compressed bytes depend on nothing but its own content.  Virtual offsets (for the BAI) are therefore known only afterwards:
 * callers note (member index, offset in member) and ask bgzf_voffset once the file is closed. */
typedef struct { uint8_t *raw; uint32_t n; uint8_t *comp; uint32_t clen; uint64_t fpos; } member_t;
/* blocks that live until the process ends come from 64 MB slabs (tens of thousands of 64 KB mallocs and frees across threads made
 * the allocator spend more time in the kernel than zlib spent compressing) */
typedef struct { uint8_t *p; size_t left; } bump_t;
static uint8_t *bump(bump_t *b, size_t n) { uint8_t *q; n = (n + 63) & ~(size_t)63; if(b->left < n) { b->left = n > (64u << 20) ? n : (64u << 20); b->p = malloc(b->left); } q = b->p; b->p += n; b->left -= n; return q; }
typedef struct { FILE *f; uint8_t blk[65280]; int n; int level; member_t *m; size_t nm, mm; size_t next; pthread_mutex_t mu; bump_t raws; } bgzf_t;
static void bgzf_flush(bgzf_t *z) {
    if(!z->n) return;
    if(z->nm == z->mm) { z->mm = z->mm ? z->mm * 2 : 4096; z->m = realloc(z->m, z->mm * sizeof(member_t)); }
    z->m[z->nm].raw = bump(&z->raws, (size_t)z->n); memcpy(z->m[z->nm].raw, z->blk, (size_t)z->n); z->m[z->nm].n = (uint32_t)z->n; z->m[z->nm].comp = NULL; z->nm++;
    z->n = 0;
}
static void bgzf_write(bgzf_t *z, const void *d, size_t n) {
    const uint8_t *p = d;
    while(n) { size_t k = sizeof(z->blk) - z->n; if(k > n) k = n; memcpy(z->blk + z->n, p, k); z->n += k; p += k; n -= k; if(z->n == (int)sizeof(z->blk)) bgzf_flush(z); }
}
static void *bgzf_worker(void *arg) {
    bgzf_t *z = arg; bump_t mine = {NULL, 0}; static __thread uint8_t out[70000 + 64];
    for(;;) {
        size_t i; member_t *m; z_stream zs; uint32_t crc; int clen; uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};
        pthread_mutex_lock(&z->mu); i = z->next++; pthread_mutex_unlock(&z->mu);
        if(i >= z->nm) break;
        m = &z->m[i];
        memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, z->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = m->raw; zs.avail_in = m->n; zs.next_out = out + 18; zs.avail_out = 70000;
        deflate(&zs, Z_FINISH); clen = (int)zs.total_out; deflateEnd(&zs);
        crc = crc32(crc32(0, NULL, 0), m->raw, m->n);
        { int bsize = clen + 25; hdr[16] = bsize & 0xff; hdr[17] = bsize >> 8; }
        memcpy(out, hdr, 18);
        { uint8_t t[8] = {crc, crc >> 8, crc >> 16, crc >> 24, (uint8_t)m->n, (uint8_t)(m->n >> 8), (uint8_t)(m->n >> 16), (uint8_t)(m->n >> 24)}; memcpy(out + 18 + clen, t, 8); }
        m->clen = (uint32_t)(18 + clen + 8); m->comp = bump(&mine, m->clen); memcpy(m->comp, out, m->clen); m->raw = NULL;
    }
    return NULL;
}
/* compress everything, fix the members' file positions, write; the member table stays for bgzf_voffset */
static uint64_t bgzf_finish(bgzf_t *z) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    pthread_t th[64]; int nt = (int)sysconf(_SC_NPROCESSORS_ONLN), k, made = 0; uint64_t fpos = 0; size_t i;
    bgzf_flush(z);
    if(getenv("MDK_SYNTH_THREADS")) nt = atoi(getenv("MDK_SYNTH_THREADS"));
    if(nt > 64) nt = 64;
    if(nt < 1) nt = 1;
    z->next = 0; pthread_mutex_init(&z->mu, NULL);
    for(k = 0; k < nt - 1; k++) { if(pthread_create(&th[made], NULL, bgzf_worker, z)) break; made++; }
    bgzf_worker(z);
    for(k = 0; k < made; k++) pthread_join(th[k], NULL);
    for(i = 0; i < z->nm; i++) { z->m[i].fpos = fpos; fwrite(z->m[i].comp, 1, z->m[i].clen, z->f); fpos += z->m[i].clen; z->m[i].comp = NULL; }
    fwrite(eof, 1, 28, z->f); fclose(z->f);
    return fpos;
}
/* (member index, offset inside it) -> virtual offset; a position at the very end of the last noted member = start of the next */
static uint64_t bgzf_voffset(const bgzf_t *z, uint64_t end_fpos, size_t member, uint32_t within) {
    if(member >= z->nm) return (end_fpos << 16) | within;
    return (z->m[member].fpos << 16) | within;
}

/* ---- reference ---- */
typedef struct { char *name; char *seq; int64_t len; float *beta; /* per C/G-of-CpG methylation level, keyed on the C position */ } contig_t;

static void make_contig(contig_t *c, rng_t *r) {
    int64_t i; char prev = 'A'; const double gc = 0.41, cpg_oe = 0.25;
    c->seq = malloc(c->len + 1); c->beta = calloc(c->len + 1, sizeof(float));
    for(i = 0; i < c->len; i++) {
        double u = rndu(r); char b;
        double pC = gc / 2, pG = gc / 2, pA = (1 - gc) / 2;
        if(prev == 'C') { double ng = pG * cpg_oe, d = pG - ng; pG = ng; pA += d / 2; /* T takes the rest */ }
        if(u < pA) b = 'A'; else if(u < pA + pC) b = 'C'; else if(u < pA + pC + pG) b = 'G'; else b = 'T';
        c->seq[i] = b; prev = b;
    }
    c->seq[c->len] = 0;
    /* N runs: 0.5 % of bases in runs of <= 100 */
    { int64_t target = c->len / 200, done = 0; while(done < target && c->len > 400) { int64_t s = (int64_t)(rndu(r) * (c->len - 100)); int l = 1 + rndi(r, 100); for(i = 0; i < l; i++) c->seq[s + i] = 'N'; done += l; } }
    /* soft-masked (lower-case) runs: ~10 % */
    { int64_t target = c->len / 10, done = 0; while(done < target && c->len > 2000) { int64_t s = (int64_t)(rndu(r) * (c->len - 1000)); int l = 50 + rndi(r, 950); for(i = 0; i < l; i++) if(c->seq[s + i] != 'N') c->seq[s + i] |= 0x20; done += l; } }
    for(i = 0; i + 1 < c->len; i++) if((c->seq[i] & 0x5f) == 'C' && (c->seq[i + 1] & 0x5f) == 'G') c->beta[i] = (rndu(r) < 0.8) ? (float)(0.65 + 0.3 * rndu(r)) : (float)(0.02 + 0.16 * rndu(r));
}

/* ---- reads ---- */
typedef struct { int32_t tid, pos; uint64_t ord; uint8_t *d; uint32_t n; } rec_t;
static int rec_cmp(const void *a, const void *b) {
    const rec_t *x = a, *y = b;
    if(x->tid != y->tid) return x->tid < y->tid ? -1 : 1;
    if(x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if(beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if(beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if(beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if(beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if(beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}
static const uint8_t nt16[256] = {['A'] = 1, ['C'] = 2, ['G'] = 4, ['T'] = 8, ['N'] = 15, ['a'] = 1, ['c'] = 2, ['g'] = 4, ['t'] = 8, ['n'] = 15};

typedef struct { int clean, bismark, extras, single; int readlen; int illumina; } opts_t;       /* illumina: read names and aux fields as long as a sequencer's and a bisulfite aligner's */
typedef struct { uint32_t op[8]; int n; int rspan, qlen; } cig_t;

/* choose a CIGAR for a read of `L` query bases */
static void make_cigar(cig_t *c, int L, rng_t *r, const opts_t *o) {
    double u = rndu(r); int a, b;
    c->n = 0;
    if(o->clean || L < 40) { c->op[c->n++] = (uint32_t)L << 4 | 0; }
    else if(u < 0.01) { b = 1 + rndi(r, 3); a = 10 + rndi(r, L - 20 - b); c->op[c->n++] = a << 4 | 0; c->op[c->n++] = b << 4 | 1; c->op[c->n++] = (L - a - b) << 4 | 0; }
    else if(u < 0.02) { b = 1 + rndi(r, 3); a = 10 + rndi(r, L - 20); c->op[c->n++] = a << 4 | 0; c->op[c->n++] = b << 4 | 2; c->op[c->n++] = (L - a) << 4 | 0; }
    else if(u < 0.045) { b = 5 + rndi(r, 16); c->op[c->n++] = b << 4 | 4; c->op[c->n++] = (L - b) << 4 | 0; }
    else if(u < 0.07) { b = 5 + rndi(r, 16); c->op[c->n++] = (L - b) << 4 | 0; c->op[c->n++] = b << 4 | 4; }
    else if(u < 0.071) { b = 100 + rndi(r, 900); a = 20 + rndi(r, L - 40); c->op[c->n++] = a << 4 | 0; c->op[c->n++] = b << 4 | 3; c->op[c->n++] = (L - a) << 4 | 0; }
    else if(u < 0.072) { a = 10 + rndi(r, L - 30); c->op[c->n++] = a << 4 | 7; c->op[c->n++] = 5 << 4 | 8; c->op[c->n++] = (L - a - 5) << 4 | 0; }   /* = and X ops */
    else if(u < 0.073) { b = 3 + rndi(r, 8); c->op[c->n++] = 7 << 4 | 5; c->op[c->n++] = b << 4 | 4; c->op[c->n++] = (L - b) << 4 | 0; }                 /* hard + soft clip */
    else c->op[c->n++] = (uint32_t)L << 4 | 0;
    c->rspan = c->qlen = 0;
    for(a = 0; a < c->n; a++) { int op = c->op[a] & 15, l = c->op[a] >> 4; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) c->rspan += l; if(op == 0 || op == 1 || op == 4 || op == 7 || op == 8) c->qlen += l; }
}

/* strand: 1 OT, 2 OB, 3 CTOT, 4 CTOB -- which conversion the read carries (odd: C->T, even: G->A) */
static void emit_read(buf_t *out, const contig_t *ct, int tid, int64_t pos, const cig_t *cg, int flag, int mapq, int32_t mpos, int32_t tlen,
                      const char *qname, int strand, uint64_t fragkey, rng_t *r, const opts_t *o, int nh, int mtid) {
    static const int qlev[4] = {2, 12, 23, 37}; int L = cg->qlen, i, k, q = 0; int64_t p = pos;
    uint8_t *seq = calloc((L + 1) / 2 + 1, 1), *qual = malloc(L + 1); size_t start = out->l; uint32_t bs;
    for(k = 0; k < cg->n; k++) {
        int op = cg->op[k] & 15, l = cg->op[k] >> 4;
        for(i = 0; i < l; i++) {
            uint8_t code; double u;
            if(op == 0 || op == 7 || op == 8) {
                char rb = (p >= 0 && p < ct->len) ? ct->seq[p] : 'N'; char ub = rb & 0x5f; char sb = ub;
                /* methylation state is a property of the fragment, so both mates agree on it */
                if((strand & 1) && ub == 'C') {
                    double beta = ((p + 1 < ct->len) && (ct->seq[p + 1] & 0x5f) == 'G') ? ct->beta[p] : 0.01;
                    double h = (mix64(fragkey ^ (uint64_t)p * 0x9E3779B97F4A7C15ULL) >> 11) * (1.0 / 9007199254740992.0);
                    if(h >= beta) sb = 'T';
                } else if(!(strand & 1) && ub == 'G') {
                    double beta = (p > 0 && (ct->seq[p - 1] & 0x5f) == 'C') ? ct->beta[p - 1] : 0.01;
                    double h = (mix64(fragkey ^ (uint64_t)p * 0x9E3779B97F4A7C15ULL) >> 11) * (1.0 / 9007199254740992.0);
                    if(h >= beta) sb = 'A';
                }
                if(!o->clean && rndu(r) < 0.005) sb = "ACGT"[rndi(r, 4)];
                if(op == 8 && sb != 'N') sb = (sb == 'A') ? 'C' : 'A';
                code = nt16[(uint8_t)sb]; p++;
            } else if(op == 1 || op == 4) code = nt16[(uint8_t)"ACGT"[rndi(r, 4)]];
            else { if(op == 2 || op == 3) p++; continue; }
            u = rndu(r);
            qual[q] = (uint8_t)(o->clean ? 37 : qlev[u < 0.02 ? 0 : u < 0.07 ? 1 : u < 0.20 ? 2 : 3]);
            seq[q >> 1] |= (q & 1) ? code : (uint8_t)(code << 4);
            q++;
        }
    }
    b32(out, 0);                                  /* block_size placeholder */
    b32(out, (uint32_t)tid); b32(out, (uint32_t)pos);
    b8(out, (uint8_t)(strlen(qname) + 1)); b8(out, (uint8_t)mapq); b16(out, (uint16_t)reg2bin(pos, pos + (cg->rspan ? cg->rspan : 1)));
    b16(out, (uint16_t)cg->n); b16(out, (uint16_t)flag); b32(out, (uint32_t)L);
    b32(out, (uint32_t)mtid); b32(out, (uint32_t)mpos); b32(out, (uint32_t)tlen);
    bput(out, qname, strlen(qname) + 1);
    for(k = 0; k < cg->n; k++) b32(out, cg->op[k]);
    bput(out, seq, (L + 1) / 2); bput(out, qual, L);
    /* aux */
    bput(out, "NMC", 3); b8(out, (uint8_t)rndi(r, 4));
    if(o->illumina) { int j; bput(out, "ASi", 3); b32(out, (uint32_t)(-(int)rndi(r, 40))); bput(out, "XMZ", 3); for(j = 0; j < L; j++) b8(out, (uint8_t)"..z.h.x.Z"[rndi(r, 9)]); b8(out, 0); }
    if(!o->clean && rndu(r) < 0.3) { bput(out, "MDZ", 3); bput(out, "150", 4); }
    if(o->bismark) { bput(out, "XRZ", 3); bput(out, (flag & 0x80) ? "GA" : "CT", 3); bput(out, "XGZ", 3); bput(out, (strand == 1 || strand == 3) ? "CT" : "GA", 3); }
    if(!o->clean && rndu(r) < 0.02) { bput(out, "XGi", 3); b32(out, (uint32_t)rndi(r, 3)); }      /* non-Bismark XG: must be ignored */
    if(nh == 1) { bput(out, "NHC", 3); b8(out, 1); } else if(nh == 2) { bput(out, "NHi", 3); b32(out, 2); }
    if(!o->clean && rndu(r) < 0.05) { bput(out, "ZBB", 3); b8(out, 'S'); b32(out, 3); b16(out, 1); b16(out, 2); b16(out, 3); }
    bs = (uint32_t)(out->l - start - 4);
    out->p[start] = bs; out->p[start + 1] = bs >> 8; out->p[start + 2] = bs >> 16; out->p[start + 3] = bs >> 24;
    free(seq); free(qual);
}

/* ---- pairs pi0 .. pi1-1 of contig t: records appended to the context's pool (one random stream: the order of the draws is the data) ---- */
typedef struct { const contig_t *ct; const opts_t *o; uint64_t seed; rng_t rr; buf_t pool; size_t *offs; rec_t *recs; size_t nrec, mrec; uint64_t ord, npairs_total, nbases; } genctx;
static void gen_pairs(genctx *G, int t, uint64_t pi0, uint64_t pi1) {
    const int64_t L = G->ct[t].len; uint64_t pi;
    for(pi = pi0; pi < pi1; pi++) {
            int flen = (int)(300 + 50 * rndn(&G->rr)), ob, mapq, nh, fl1, fl2, strand1, strand2, single = G->o->single; int64_t fs; cig_t c1, c2; char qn[64]; uint64_t fragkey;
            int64_t p1, p2; double u; int extraflag = 0, discord = 0, singleton = 0;
            if(flen < G->o->readlen) flen = G->o->readlen; if(flen > 4 * G->o->readlen) flen = 4 * G->o->readlen; if(flen > L) flen = (int)L;
            fs = (int64_t)(rndu(&G->rr) * (L - flen + 1));
            ob = rndu(&G->rr) < 0.5;
            make_cigar(&c1, G->o->readlen, &G->rr, G->o); make_cigar(&c2, G->o->readlen, &G->rr, G->o);
            if(G->o->illumina) snprintf(qn, sizeof(qn), "A00%03d:%d:HXXXXXXXX:%d:%04d:%05" PRIu64 ":%05" PRIu64, 123 + t % 7, 45 + t, 1 + (int)(pi & 3), 1101 + (int)(pi % 977), pi % 32749, pi);
            else snprintf(qn, sizeof(qn), "f%d_%" PRIu64, t, pi);
            fragkey = mix64(G->seed ^ ((uint64_t)t << 48) ^ pi);
            mapq = 40 + rndi(&G->rr, 21); nh = 0; fl1 = 0; fl2 = 0;
            if(!G->o->clean) {
                if(rndu(&G->rr) < 0.02) mapq = rndi(&G->rr, 10);
                u = rndu(&G->rr); if(u < 0.10) nh = 1; else if(u < 0.11) nh = 2;
                u = rndu(&G->rr); if(u < 0.03) extraflag = 0x400; else if(u < 0.04) extraflag = 0x200;
                if(rndu(&G->rr) < 0.01) discord = 1;
                if(rndu(&G->rr) < 0.005) singleton = 1;
            }
            /* left read at fs, right read ends at fs+flen */
            p1 = fs; p2 = fs + flen - c2.rspan; if(p2 < 0) p2 = 0; if(p2 + c2.rspan > L) p2 = L - c2.rspan; if(p1 + c1.rspan > L) p1 = L - c1.rspan;
            if(p2 < p1) p2 = p1;
            if(single) {
                int rev = ob; strand1 = ob ? 2 : 1; fl1 = (rev ? 0x10 : 0) | extraflag;
                if(G->o->bismark && rndu(&G->rr) < 0.1) { strand1 = ob ? 4 : 3; fl1 ^= 0x10; }
                if(G->nrec + 1 > G->mrec) { G->mrec = G->mrec ? G->mrec * 2 : 1 << 16; G->offs = realloc(G->offs, G->mrec * sizeof(size_t)); G->recs = realloc(G->recs, G->mrec * sizeof(rec_t)); }
                G->offs[G->nrec] = G->pool.l; emit_read(&G->pool, &G->ct[t], t, p1, &c1, fl1, mapq, -1, 0, qn, strand1, fragkey, &G->rr, G->o, nh, -1);
                G->recs[G->nrec].tid = t; G->recs[G->nrec].pos = (int32_t)p1; G->recs[G->nrec].ord = G->ord++; G->recs[G->nrec].n = (uint32_t)(G->pool.l - G->offs[G->nrec]); G->nrec++; G->nbases += c1.qlen; G->npairs_total++;
                continue;
            }
            if(!ob) { fl1 = 0x1 | 0x2 | 0x20 | 0x40; fl2 = 0x1 | 0x2 | 0x10 | 0x80; strand1 = strand2 = 1; }       /* 99 / 147 */
            else { fl1 = 0x1 | 0x2 | 0x20 | 0x80; fl2 = 0x1 | 0x2 | 0x10 | 0x40; strand1 = strand2 = 2; }       /* 163 / 83 */
            if(G->o->bismark && rndu(&G->rr) < 0.1) {      /* non-directional: CTOT / CTOB pairs (read#1/#2 roles swapped) */
                fl1 ^= 0xC0; fl2 ^= 0xC0; strand1 = strand2 = ob ? 4 : 3;
            }
            if(discord) { fl1 &= ~0x2; fl2 &= ~0x2; }
            fl1 |= extraflag; fl2 |= extraflag;
            if(G->nrec + 4 > G->mrec) { G->mrec = G->mrec ? G->mrec * 2 : 1 << 16; G->offs = realloc(G->offs, G->mrec * sizeof(size_t)); G->recs = realloc(G->recs, G->mrec * sizeof(rec_t)); }
            if(singleton) {                          /* mate unmapped: keep only the left read */
                fl1 = (fl1 | 0x8) & ~0x2 & ~0x20;
                G->offs[G->nrec] = G->pool.l; emit_read(&G->pool, &G->ct[t], t, p1, &c1, fl1, mapq, (int32_t)p1, 0, qn, strand1, fragkey, &G->rr, G->o, nh, t);
                G->recs[G->nrec].tid = t; G->recs[G->nrec].pos = (int32_t)p1; G->recs[G->nrec].ord = G->ord++; G->recs[G->nrec].n = (uint32_t)(G->pool.l - G->offs[G->nrec]); G->nrec++; G->nbases += c1.qlen; G->npairs_total++;
                continue;
            }
            G->offs[G->nrec] = G->pool.l; emit_read(&G->pool, &G->ct[t], t, p1, &c1, fl1, mapq, (int32_t)p2, (int32_t)(p2 + c2.rspan - p1), qn, strand1, fragkey, &G->rr, G->o, nh, t);
            G->recs[G->nrec].tid = t; G->recs[G->nrec].pos = (int32_t)p1; G->recs[G->nrec].ord = G->ord++; G->recs[G->nrec].n = (uint32_t)(G->pool.l - G->offs[G->nrec]); G->nrec++;
            G->offs[G->nrec] = G->pool.l; emit_read(&G->pool, &G->ct[t], t, p2, &c2, fl2, mapq, (int32_t)p1, -(int32_t)(p2 + c2.rspan - p1), qn, strand2, fragkey, &G->rr, G->o, nh, t);
            G->recs[G->nrec].tid = t; G->recs[G->nrec].pos = (int32_t)p2; G->recs[G->nrec].ord = G->ord++; G->recs[G->nrec].n = (uint32_t)(G->pool.l - G->offs[G->nrec]); G->nrec++;
            G->nbases += c1.qlen + c2.qlen; G->npairs_total++;
            if(G->o->extras && rndu(&G->rr) < 0.02) {       /* a secondary or supplementary record sharing the qname */
                cig_t c3; int64_t p3; int f3 = fl1 | (rndu(&G->rr) < 0.5 ? 0x100 : 0x800);
                make_cigar(&c3, G->o->readlen, &G->rr, G->o); p3 = fs + rndi(&G->rr, flen); if(p3 + c3.rspan > L) p3 = L - c3.rspan; if(p3 < 0) p3 = 0;
                G->offs[G->nrec] = G->pool.l; emit_read(&G->pool, &G->ct[t], t, p3, &c3, f3, mapq, (int32_t)p2, 0, qn, strand1, fragkey, &G->rr, G->o, nh, t);
                G->recs[G->nrec].tid = t; G->recs[G->nrec].pos = (int32_t)p3; G->recs[G->nrec].ord = G->ord++; G->recs[G->nrec].n = (uint32_t)(G->pool.l - G->offs[G->nrec]); G->nrec++;
            }
            }
}


/* ---- -j N: the same kind of data made by N threads (contigs' bases per contig, reads in blocks of pairs with a random stream each, sorting, BGZF members and
 * the index per contig, compression and the file's writing by all) -- for inputs of human size in a bench's time.  The data differ from what the serial
 * path makes of the same seed (other random streams) and are as reproducible: a (seed, args, -j) ... the number of threads does not enter the streams. ---- */
#define PAR_BLOCK 200000u            /* pairs per unit of work */
typedef struct { int t; uint64_t pi0, pi1; genctx G; } par_unit;
typedef struct { bgzf_t z; rec_t *recs; size_t nrec; size_t *rm, *em; uint32_t *rw, *ew; uint64_t **lin_; uint64_t *lin, first, last; size_t nlin; uint64_t fpos0, fpos1; uint64_t nbases, npairs; } par_contig;
typedef struct { contig_t *ct; int nct; const opts_t *o; uint64_t seed; double cov; int split_records;
                 par_unit *unit; size_t n_unit; size_t next; pthread_mutex_t mu; par_contig *pc; int phase; member_t **mem; size_t n_mem; int level; int fd; } par_ctx;
static size_t par_take(par_ctx *P, size_t n) { size_t i; pthread_mutex_lock(&P->mu); i = P->next++; pthread_mutex_unlock(&P->mu); return i < n ? i : (size_t)-1; }
static void par_cut_contig(par_ctx *P, int t) {          /* gather, sort, cut into members (a contig starts a member of its own), note where every record lies */
    par_contig *c = &P->pc[t]; size_t u, n = 0, i; bgzf_t *z = &c->z;
    for(u = 0; u < P->n_unit; u++) if(P->unit[u].t == t) n += P->unit[u].G.nrec;
    c->recs = malloc((n + 1) * sizeof(rec_t)); c->nrec = 0;
    for(u = 0; u < P->n_unit; u++) if(P->unit[u].t == t) { genctx *G = &P->unit[u].G; for(i = 0; i < G->nrec; i++) { rec_t r = G->recs[i]; r.d = G->pool.p + G->offs[i]; r.ord = ((uint64_t)u << 32) | i; c->recs[c->nrec++] = r; } c->nbases += G->nbases; c->npairs += G->npairs_total; free(G->recs); free(G->offs); G->recs = NULL; G->offs = NULL; }
    qsort(c->recs, c->nrec, sizeof(rec_t), rec_cmp);
    c->rm = malloc((c->nrec + 1) * sizeof(size_t)); c->em = malloc((c->nrec + 1) * sizeof(size_t)); c->rw = malloc((c->nrec + 1) * 4); c->ew = malloc((c->nrec + 1) * 4);
    for(i = 0; i < c->nrec; i++) {
        if(z->n + 4 > (int)sizeof(z->blk)) bgzf_flush(z);
        if(!P->split_records && z->n && c->recs[i].n <= sizeof(z->blk) && z->n + c->recs[i].n > sizeof(z->blk)) bgzf_flush(z);
        c->rm[i] = z->nm; c->rw[i] = (uint32_t)z->n;
        bgzf_write(z, c->recs[i].d, c->recs[i].n);
        c->em[i] = z->nm; c->ew[i] = (uint32_t)z->n;
    }
    bgzf_flush(z);
}
static void par_index_contig(par_ctx *P, int t) {        /* the BAI entries of a contig, once its members' places in the file are known */
    par_contig *c = &P->pc[t]; size_t i; const int64_t len = P->ct[t].len;
    c->nlin = (size_t)((len >> 14) + 1); c->lin = calloc(c->nlin, 8);
    for(i = 0; i < c->nrec; i++) {
        const uint64_t vo = bgzf_voffset(&c->z, c->fpos1, c->rm[i], c->rw[i]); const int32_t pos = c->recs[i].pos; int64_t w, w1; uint32_t ncig, k, rl = 0; const uint8_t *r = c->recs[i].d + 4;
        ncig = r[12] | (r[13] << 8);
        for(k = 0; k < ncig; k++) { const uint8_t *cg = r + 32 + r[8] + 4 * k; uint32_t cv = cg[0] | (cg[1] << 8) | (cg[2] << 16) | ((uint32_t)cg[3] << 24), op = cv & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cv >> 4; }
        w1 = ((int64_t)pos + (rl ? rl : 1) - 1) >> 14;
        for(w = pos >> 14; w <= w1 && w < (int64_t)c->nlin; w++) if(!c->lin[w]) c->lin[w] = vo;
        if(!c->first) c->first = vo;
    }
    free(c->rm); free(c->em); free(c->rw); free(c->ew); free(c->recs);
}
static void par_compress(member_t *m, int level, bump_t *mine) {
    static __thread uint8_t out[70000 + 64]; z_stream zs; uint32_t crc; int clen; uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};
    memset(&zs, 0, sizeof(zs)); deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = m->raw; zs.avail_in = m->n; zs.next_out = out + 18; zs.avail_out = 70000;
    deflate(&zs, Z_FINISH); clen = (int)zs.total_out; deflateEnd(&zs);
    crc = crc32(crc32(0, NULL, 0), m->raw, m->n);
    { int bsize = clen + 25; hdr[16] = bsize & 0xff; hdr[17] = bsize >> 8; }
    memcpy(out, hdr, 18);
    { uint8_t tr[8] = {crc, crc >> 8, crc >> 16, crc >> 24, (uint8_t)m->n, (uint8_t)(m->n >> 8), (uint8_t)(m->n >> 16), (uint8_t)(m->n >> 24)}; memcpy(out + 18 + clen, tr, 8); }
    m->clen = (uint32_t)(18 + clen + 8); m->comp = bump(mine, m->clen); memcpy(m->comp, out, m->clen); m->raw = NULL;
}
static void *par_worker(void *arg) {
    par_ctx *P = arg; size_t i; bump_t mine = {NULL, 0};
    if(P->phase == 0) { while((i = par_take(P, (size_t)P->nct)) != (size_t)-1) { rng_t rg; rg.s = mix64(P->seed * 0x9E3779B97F4A7C15ULL + 0x1000 + i) | 1; make_contig(&P->ct[i], &rg); } }
    else if(P->phase == 1) { while((i = par_take(P, P->n_unit)) != (size_t)-1) { par_unit *u = &P->unit[i]; u->G.ct = P->ct; u->G.o = P->o; u->G.seed = P->seed; u->G.rr.s = mix64(P->seed ^ ((uint64_t)(u->t + 1) << 40) ^ (u->pi0 * 0x9E3779B97F4A7C15ULL)) | 1; gen_pairs(&u->G, u->t, u->pi0, u->pi1); } }
    else if(P->phase == 2) { while((i = par_take(P, (size_t)P->nct)) != (size_t)-1) par_cut_contig(P, (int)i); }
    else if(P->phase == 3) { while((i = par_take(P, (P->n_mem + 63) / 64)) != (size_t)-1) { size_t k; for(k = 64 * i; k < 64 * i + 64 && k < P->n_mem; k++) par_compress(P->mem[k], P->level, &mine); } }
    else if(P->phase == 4) { while((i = par_take(P, (P->n_mem + 255) / 256)) != (size_t)-1) { size_t k; for(k = 256 * i; k < 256 * i + 256 && k < P->n_mem; k++) { member_t *m = P->mem[k]; size_t done = 0; while(done < m->clen) { ssize_t w = pwrite(P->fd, m->comp + done, m->clen - done, (off_t)(m->fpos + done)); if(w <= 0) { perror("pwrite"); exit(1); } done += (size_t)w; } } } }
    else if(P->phase == 5) { while((i = par_take(P, (size_t)P->nct)) != (size_t)-1) par_index_contig(P, (int)i); }
    return NULL;
}
static void par_run(par_ctx *P, int phase, int nt) {
    pthread_t th[256]; int k, made = 0;
    P->phase = phase; P->next = 0;
    for(k = 0; k < nt - 1 && k < 255; k++) { if(pthread_create(&th[made], NULL, par_worker, P)) break; made++; }
    par_worker(P);
    for(k = 0; k < made; k++) pthread_join(th[k], NULL);
}
int main(int argc, char **argv) {
    static struct option lo[] = {{"bismark", 0, 0, 1}, {"extras", 0, 0, 2}, {"clean", 0, 0, 3}, {"bbm", 0, 0, 4}, {"single", 0, 0, 5}, {"bw", 0, 0, 6}, {"no-bai", 0, 0, 7}, {"split-records", 0, 0, 8}, {"illumina", 0, 0, 9}, {0, 0, 0, 0}};
    const char *prefix = NULL, *lens = "1000000"; double cov = 30; uint64_t seed = 0x5EED0001ULL; int level = 1, c, want_bbm = 0;
    opts_t o = {0, 0, 0, 0, 150}; int want_bw = 0, no_bai = 0, split_records = 0, par = 0; contig_t *ct = NULL; int nct = 0, t; rng_t rr, rg; rec_t *recs = NULL; size_t nrec = 0, mrec = 0, i;
    buf_t pool = {0, 0, 0}; size_t *offs = NULL; char fn[4096]; uint64_t ord = 0, npairs_total = 0, nbases = 0;
    while((c = getopt_long(argc, argv, "o:L:c:l:s:z:j:", lo, NULL)) >= 0) {
        switch(c) {
        case 'o': prefix = optarg; break; case 'L': lens = optarg; break; case 'c': cov = atof(optarg); break;
        case 'l': o.readlen = atoi(optarg); break; case 's': seed = strtoull(optarg, NULL, 0); break; case 'z': level = atoi(optarg); break; case 'j': par = atoi(optarg); break;
        case 1: o.bismark = 1; break; case 2: o.extras = 1; break; case 3: o.clean = 1; break; case 4: want_bbm = 1; break; case 5: o.single = 1; break; case 6: want_bw = 1; break; case 7: no_bai = 1; break; case 8: split_records = 1; break; case 9: o.illumina = 1; break;
        default: fprintf(stderr, "usage: mdk_synth -o PREFIX [-L len,len..] [-c cov] [-l readlen] [-s seed] [-z level] [-j threads] [--bismark] [--extras] [--clean] [--bbm] [--bw] [--single] [--no-bai] [--split-records] [--illumina]\n"); return 1;
        }
    }
    if(!prefix) { fprintf(stderr, "mdk_synth: -o PREFIX is required\n"); return 1; }
    rg.s = seed ? seed : 1; rr.s = mix64(seed + 1) | 1;
    { char *s = strdup(lens), *p = strtok(s, ","); while(p) { ct = realloc(ct, sizeof(*ct) * (nct + 1)); memset(&ct[nct], 0, sizeof(*ct)); ct[nct].len = atoll(p); asprintf(&ct[nct].name, nct ? "chrS%d" : "chrS1", nct + 1); nct++; p = strtok(NULL, ","); } free(s); }
    if(par > 0) {
        par_ctx P; size_t u = 0, k; uint64_t fpos = 0; bgzf_t zh; buf_t h = {0, 0, 0}; char *txt = malloc(65536 + 64 * (size_t)nct); int n = 0; FILE *f;
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if(par > 256) par = 256;
        memset(&P, 0, sizeof(P)); P.ct = ct; P.nct = nct; P.o = &o; P.seed = seed; P.cov = cov; P.split_records = split_records; P.level = level; pthread_mutex_init(&P.mu, NULL);
        par_run(&P, 0, par);                                                          /* the contigs' bases */
        snprintf(fn, sizeof(fn), "%s.fa", prefix); f = fopen(fn, "w"); if(!f) { perror(fn); return 1; }
        for(t = 0; t < nct; t++) { int64_t j; char *lines = malloc((size_t)ct[t].len + (size_t)ct[t].len / 60 + 2), *q = lines; fprintf(f, ">%s synthetic seed=%" PRIu64 "\n", ct[t].name, seed);
            for(j = 0; j < ct[t].len; j += 60) { const size_t w = (size_t)((ct[t].len - j) < 60 ? (ct[t].len - j) : 60); memcpy(q, ct[t].seq + j, w); q += w; *q++ = '\n'; }
            fwrite(lines, 1, (size_t)(q - lines), f); free(lines); }
        fclose(f);
        for(t = 0; t < nct; t++) { const uint64_t np = ct[t].len < 2 * o.readlen ? 0 : (uint64_t)(ct[t].len * cov / (2.0 * o.readlen)); P.n_unit += (size_t)((np + PAR_BLOCK - 1) / PAR_BLOCK); }
        P.unit = calloc(P.n_unit + 1, sizeof(par_unit));
        for(t = 0; t < nct; t++) { const uint64_t np = ct[t].len < 2 * o.readlen ? 0 : (uint64_t)(ct[t].len * cov / (2.0 * o.readlen)); uint64_t a; for(a = 0; a < np; a += PAR_BLOCK) { P.unit[u].t = t; P.unit[u].pi0 = a; P.unit[u].pi1 = a + PAR_BLOCK < np ? a + PAR_BLOCK : np; u++; } }
        par_run(&P, 1, par);                                                          /* the reads */
        P.pc = calloc((size_t)nct + 1, sizeof(par_contig));
        par_run(&P, 2, par);                                                          /* per contig: sorted, cut into members */
        memset(&zh, 0, sizeof(zh));                                                   /* the header's members */
        n += snprintf(txt + n, 65536, "@HD\tVN:1.6\tSO:coordinate\n");
        for(t = 0; t < nct; t++) n += snprintf(txt + n, 64, "@SQ\tSN:%s\tLN:%" PRId64 "\n", ct[t].name, ct[t].len);
        n += snprintf(txt + n, 256, "@PG\tID:mdk_synth\tPN:mdk_synth\tCL:seed=%" PRIu64 " -j\n", seed);
        bput(&h, "BAM\1", 4); b32(&h, (uint32_t)n); bput(&h, txt, (size_t)n); b32(&h, (uint32_t)nct);
        for(t = 0; t < nct; t++) { b32(&h, (uint32_t)strlen(ct[t].name) + 1); bput(&h, ct[t].name, strlen(ct[t].name) + 1); b32(&h, (uint32_t)ct[t].len); }
        bgzf_write(&zh, h.p, h.l); bgzf_flush(&zh);
        P.n_mem = zh.nm; for(t = 0; t < nct; t++) P.n_mem += P.pc[t].z.nm;
        P.mem = malloc((P.n_mem + 1) * sizeof(member_t *)); k = 0;
        { size_t i; for(i = 0; i < zh.nm; i++) P.mem[k++] = &zh.m[i]; for(t = 0; t < nct; t++) for(i = 0; i < P.pc[t].z.nm; i++) P.mem[k++] = &P.pc[t].z.m[i]; }
        par_run(&P, 3, par);                                                          /* compressed */
        { size_t i; for(i = 0; i < zh.nm; i++) { zh.m[i].fpos = fpos; fpos += zh.m[i].clen; } for(t = 0; t < nct; t++) { P.pc[t].fpos0 = fpos; for(i = 0; i < P.pc[t].z.nm; i++) { P.pc[t].z.m[i].fpos = fpos; fpos += P.pc[t].z.m[i].clen; } P.pc[t].fpos1 = fpos; } }
        snprintf(fn, sizeof(fn), "%s.bam", prefix); f = fopen(fn, "wb"); if(!f) { perror(fn); return 1; }
        P.fd = fileno(f); if(ftruncate(P.fd, (off_t)(fpos + 28)) != 0) { perror("ftruncate"); return 1; }
        par_run(&P, 4, par);                                                          /* written */
        if(pwrite(P.fd, eof, 28, (off_t)fpos) != 28) { perror("pwrite"); return 1; }
        fclose(f);
        par_run(&P, 5, par);                                                          /* indexed */
        for(t = 0; t < nct; t++) { nrec += P.pc[t].nrec; npairs_total += P.pc[t].npairs; nbases += P.pc[t].nbases; if(P.pc[t].first) P.pc[t].last = fpos << 16; }
        if(!no_bai) {
            buf_t x = {0, 0, 0}; FILE *bf;
            bput(&x, "BAI\1", 4); b32(&x, (uint32_t)nct);
            for(t = 0; t < nct; t++) {
                par_contig *c = &P.pc[t]; size_t w; uint64_t prev = 0;
                if(c->first) { b32(&x, 1); b32(&x, 0); b32(&x, 1); bput(&x, &c->first, 8); bput(&x, &c->last, 8); } else b32(&x, 0);
                for(w = 0; w < c->nlin; w++) { if(c->lin[w]) prev = c->lin[w]; else c->lin[w] = prev; }
                { size_t nn = c->nlin; while(nn && !c->lin[nn - 1]) nn--; b32(&x, (uint32_t)nn); bput(&x, c->lin, 8 * nn); }
            }
            snprintf(fn, sizeof(fn), "%s.bam.bai", prefix); bf = fopen(fn, "wb"); if(!bf) { perror(fn); return 1; }
            fwrite(x.p, 1, x.l, bf); fclose(bf); free(x.p);
        }
        goto tracks;
    }
    snprintf(fn, sizeof(fn), "%s.fa", prefix);
    { FILE *f = fopen(fn, "w"); if(!f) { perror(fn); return 1; }
      for(t = 0; t < nct; t++) { int64_t j; make_contig(&ct[t], &rg); fprintf(f, ">%s synthetic seed=%" PRIu64 "\n", ct[t].name, seed); for(j = 0; j < ct[t].len; j += 60) { fwrite(ct[t].seq + j, 1, (ct[t].len - j) < 60 ? (ct[t].len - j) : 60, f); fputc('\n', f); } }
      fclose(f); }

    { genctx G; memset(&G, 0, sizeof(G)); G.ct = ct; G.o = &o; G.seed = seed; G.rr = rr;
      for(t = 0; t < nct; t++) { const int64_t L = ct[t].len; const uint64_t npairs = (uint64_t)(L * cov / (2.0 * o.readlen)); if(L < 2 * o.readlen) continue; gen_pairs(&G, t, 0, npairs); }
      pool = G.pool; offs = G.offs; recs = G.recs; nrec = G.nrec; mrec = G.mrec; ord = G.ord; npairs_total = G.npairs_total; nbases = G.nbases; rr = G.rr; }
    if(getenv("MDK_SYNTH_PROFILE")) fprintf(stderr, "[synth] reads generated at %.2f s\n", clock() / (double)CLOCKS_PER_SEC);
    for(i = 0; i < nrec; i++) recs[i].d = pool.p + offs[i];
    qsort(recs, nrec, sizeof(rec_t), rec_cmp);
    if(getenv("MDK_SYNTH_PROFILE")) fprintf(stderr, "[synth] sorted at %.2f s (cpu)\n", clock() / (double)CLOCKS_PER_SEC);

    snprintf(fn, sizeof(fn), "%s.bam", prefix);
    { bgzf_t z; buf_t h = {0, 0, 0}; char txt[65536]; int n = 0;
      memset(&z, 0, sizeof(z)); z.f = fopen(fn, "wb"); z.level = level; if(!z.f) { perror(fn); return 1; }
      n += snprintf(txt + n, sizeof(txt) - n, "@HD\tVN:1.6\tSO:coordinate\n");
      for(t = 0; t < nct; t++) n += snprintf(txt + n, sizeof(txt) - n, "@SQ\tSN:%s\tLN:%" PRId64 "\n", ct[t].name, ct[t].len);
      n += snprintf(txt + n, sizeof(txt) - n, "@PG\tID:mdk_synth\tPN:mdk_synth\tCL:seed=%" PRIu64 "\n", seed);
      bput(&h, "BAM\1", 4); b32(&h, (uint32_t)n); bput(&h, txt, n); b32(&h, (uint32_t)nct);
      for(t = 0; t < nct; t++) { b32(&h, (uint32_t)strlen(ct[t].name) + 1); bput(&h, ct[t].name, strlen(ct[t].name) + 1); b32(&h, (uint32_t)ct[t].len); }
      bgzf_write(&z, h.p, h.l); bgzf_flush(&z);
      {   /* records + a BAI (linear index per contig, one catch-all bin 0 per contig holding the contig's byte range) */
          uint64_t **lin = calloc(nct, sizeof(uint64_t *)), *first = calloc(nct, 8), *last = calloc(nct, 8); size_t *nlin = calloc(nct, sizeof(size_t));
          /* where every record starts / ends: (member, offset in member); the virtual offsets follow once the members are compressed */
          size_t *rm = malloc((nrec + 1) * sizeof(size_t)), *em = malloc((nrec + 1) * sizeof(size_t)); uint32_t *rw = malloc((nrec + 1) * 4), *ew = malloc((nrec + 1) * 4); uint64_t end_fpos;
          for(t = 0; t < nct; t++) { nlin[t] = (size_t)((ct[t].len >> 14) + 1); lin[t] = calloc(nlin[t], 8); }
          for(i = 0; i < nrec; i++) {
              if(z.n + 4 > (int)sizeof(z.blk)) bgzf_flush(&z);          /* keep the block_size word inside one member */
              /* as htslib does (bam_write1 -> bgzf_flush_try): a record that fits a member never straddles two */
              if(!split_records && z.n && recs[i].n <= sizeof(z.blk) && z.n + recs[i].n > sizeof(z.blk)) bgzf_flush(&z);
              rm[i] = z.nm; rw[i] = (uint32_t)z.n;
              bgzf_write(&z, recs[i].d, recs[i].n);
              em[i] = z.nm; ew[i] = (uint32_t)z.n;
          }
          end_fpos = bgzf_finish(&z);
          for(i = 0; i < nrec; i++) {
              uint64_t vo = bgzf_voffset(&z, end_fpos, rm[i], rw[i]); int32_t tid = recs[i].tid, pos = recs[i].pos; int64_t w, w1; uint32_t ncig, k, rl = 0; const uint8_t *r = recs[i].d + 4;
              ncig = r[12] | (r[13] << 8);
              for(k = 0; k < ncig; k++) { const uint8_t *c = r + 32 + r[8] + 4 * k; uint32_t cv = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24), op = cv & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cv >> 4; }
              w1 = ((int64_t)pos + (rl ? rl : 1) - 1) >> 14;
              for(w = pos >> 14; w <= w1 && w < (int64_t)nlin[tid]; w++) if(!lin[tid][w]) lin[tid][w] = vo;
              if(!first[tid]) first[tid] = vo;
              last[tid] = bgzf_voffset(&z, end_fpos, em[i], ew[i]);
          }
          for(t = 0; t < nct; t++) if(first[t]) last[t] = end_fpos << 16;      /* generous end for the catch-all chunk */
          free(rm); free(em); free(rw); free(ew);
          if(!no_bai) {
              buf_t x = {0, 0, 0}; FILE *bf;
              bput(&x, "BAI\1", 4); b32(&x, (uint32_t)nct);
              for(t = 0; t < nct; t++) {
                  size_t w; uint64_t prev = 0;
                  if(first[t]) { b32(&x, 1); b32(&x, 0); b32(&x, 1); bput(&x, &first[t], 8); bput(&x, &last[t], 8); } else b32(&x, 0);
                  for(w = 0; w < nlin[t]; w++) { if(lin[t][w]) prev = lin[t][w]; else lin[t][w] = prev; }     /* samtools fills gaps with the previous offset */
                  { size_t n = nlin[t]; while(n && !lin[t][n - 1]) n--; b32(&x, (uint32_t)n); bput(&x, lin[t], 8 * n); }
              }
              snprintf(fn, sizeof(fn), "%s.bam.bai", prefix); bf = fopen(fn, "wb"); if(!bf) { perror(fn); return 1; }
              fwrite(x.p, 1, x.l, bf); fclose(bf); free(x.p);
          }
      }
      free(z.m); free(h.p); }

tracks:
    if(want_bbm || want_bw) {     /* synthetic mappability track: values {0, 0.5, 1.0}; written as BBM and/or bigWig (same values) */
        typedef struct { int64_t beg, end; uint8_t val; } mrun; mrun **runs = calloc(nct, sizeof(mrun *)); size_t *nr = calloc(nct, sizeof(size_t));
        for(t = 0; t < nct; t++) {
            int64_t p = 0; size_t cap = 0;
            while(p < ct[t].len) {
                int64_t run = (rndu(&rg) < 0.5) ? 1 + rndi(&rg, 3000) : 1 + rndi(&rg, 300); uint8_t val = (rndu(&rg) < 0.25) ? (rndu(&rg) < 0.5 ? 0 : 50) : 100;
                if(run > ct[t].len - p) run = ct[t].len - p;
                if(nr[t] && runs[t][nr[t] - 1].val == val) runs[t][nr[t] - 1].end += run;
                else { if(nr[t] == cap) { cap = cap ? cap * 2 : 1024; runs[t] = realloc(runs[t], cap * sizeof(mrun)); } runs[t][nr[t]].beg = p; runs[t][nr[t]].end = p + run; runs[t][nr[t]].val = val; nr[t]++; }
                p += run;
            }
        }
        if(want_bbm) {
            FILE *f; uint8_t ver = 1; uint32_t nc = (uint32_t)nct;
            snprintf(fn, sizeof(fn), "%s.bbm", prefix); f = fopen(fn, "wb"); if(!f) { perror(fn); return 1; }
            fwrite(&ver, 1, 1, f); fwrite(&nc, 4, 1, f);
            for(t = 0; t < nct; t++) {
                uint16_t nl = (uint16_t)strlen(ct[t].name); uint8_t z0 = 0; uint32_t cl = (uint32_t)ct[t].len; size_t k;
                fwrite(&nl, 2, 1, f); fwrite(ct[t].name, 1, nl, f); fwrite(&z0, 1, 1, f); fwrite(&cl, 4, 1, f);
                for(k = 0; k < nr[t]; k++) {
                    int64_t run = runs[t][k].end - runs[t][k].beg; uint8_t val = runs[t][k].val;
                    while(run > 0) {
                        if(run == 1) { fwrite(&val, 1, 1, f); run = 0; }
                        else if(run <= 155) { uint8_t rl = (uint8_t)(run + 99); fwrite(&rl, 1, 1, f); fwrite(&val, 1, 1, f); run = 0; }
                        else { uint8_t fl = 255; uint16_t rl = (uint16_t)(run > 65535 ? 65535 : run); fwrite(&fl, 1, 1, f); fwrite(&rl, 2, 1, f); fwrite(&val, 1, 1, f); run -= rl; }
                    }
                }
            }
            fclose(f);
        }
        if(want_bw) {      /* bigWig: bedGraph sections of <= 512 items, zlib-compressed, one-level R-tree; value-0 runs are left uncovered (NaN) */
            FILE *f; buf_t body = {0, 0, 0}, idx = {0, 0, 0}; uint32_t nblocks = 0, keySize = 0, maxraw = 0; uint64_t chromTree, dataOff, indexOff; buf_t hdr = {0, 0, 0};
            for(t = 0; t < nct; t++) if(strlen(ct[t].name) > keySize) keySize = (uint32_t)strlen(ct[t].name);
            chromTree = 64 + 40; dataOff = chromTree + 32 + 4 + (uint64_t)nct * (keySize + 8);
            b32(&body, 0); b32(&body, 0);                      /* section count (u64), patched below */
            for(t = 0; t < nct; t++) {
                size_t k = 0;
                while(k < nr[t]) {
                    buf_t raw = {0, 0, 0}; uint16_t cnt = 0; uint32_t s0 = 0, e0 = 0; size_t k0 = k; uLongf cl; uint8_t *comp;
                    b32(&raw, (uint32_t)t); b32(&raw, 0); b32(&raw, 0); b32(&raw, 0); b32(&raw, 0); b8(&raw, 1); b8(&raw, 0); b16(&raw, 0);
                    for(; k < nr[t] && cnt < 512; k++) {
                        float v = runs[t][k].val / 100.0f;
                        if(runs[t][k].val == 0) continue;
                        if(!cnt) s0 = (uint32_t)runs[t][k].beg;
                        e0 = (uint32_t)runs[t][k].end;
                        b32(&raw, (uint32_t)runs[t][k].beg); b32(&raw, (uint32_t)runs[t][k].end); bput(&raw, &v, 4); cnt++;
                    }
                    if(!cnt) { free(raw.p); if(k == k0) k++; continue; }
                    raw.p[4] = s0; raw.p[5] = s0 >> 8; raw.p[6] = s0 >> 16; raw.p[7] = s0 >> 24; raw.p[8] = e0; raw.p[9] = e0 >> 8; raw.p[10] = e0 >> 16; raw.p[11] = e0 >> 24;
                    raw.p[22] = cnt & 0xff; raw.p[23] = cnt >> 8;
                    if(raw.l > maxraw) maxraw = (uint32_t)raw.l;
                    cl = compressBound(raw.l); comp = malloc(cl); compress2(comp, &cl, raw.p, raw.l, 6);
                    b32(&idx, (uint32_t)t); b32(&idx, s0); b32(&idx, (uint32_t)t); b32(&idx, e0);
                    { uint64_t off = dataOff + body.l, sz = cl; bput(&idx, &off, 8); bput(&idx, &sz, 8); }
                    bput(&body, comp, cl); free(comp); free(raw.p); nblocks++;
                }
            }
            if(nblocks > 65535) { fprintf(stderr, "mdk_synth: too many bigWig blocks for a one-level index\n"); return 1; }
            { uint64_t nb = nblocks; memcpy(body.p, &nb, 8); }
            indexOff = dataOff + body.l;
            b32(&hdr, 0x888FFC26u); b16(&hdr, 4); b16(&hdr, 0); bput(&hdr, &chromTree, 8); bput(&hdr, &dataOff, 8); bput(&hdr, &indexOff, 8);
            b16(&hdr, 0); b16(&hdr, 0); { uint64_t z = 0, ts = 64; bput(&hdr, &z, 8); bput(&hdr, &ts, 8); } b32(&hdr, maxraw); { uint64_t z = 0; bput(&hdr, &z, 8); }
            { uint8_t summ[40] = {0}; bput(&hdr, summ, 40); }
            b32(&hdr, 0x78CA8C91u); b32(&hdr, (uint32_t)nct); b32(&hdr, keySize); b32(&hdr, 8); { uint64_t ic = nct, z = 0; bput(&hdr, &ic, 8); bput(&hdr, &z, 8); }
            b8(&hdr, 1); b8(&hdr, 0); b16(&hdr, (uint16_t)nct);
            for(t = 0; t < nct; t++) { char key[256] = {0}; strncpy(key, ct[t].name, keySize); bput(&hdr, key, keySize); b32(&hdr, (uint32_t)t); b32(&hdr, (uint32_t)ct[t].len); }
            snprintf(fn, sizeof(fn), "%s.bw", prefix); f = fopen(fn, "wb"); if(!f) { perror(fn); return 1; }
            fwrite(hdr.p, 1, hdr.l, f); fwrite(body.p, 1, body.l, f);
            { buf_t r = {0, 0, 0}; uint64_t ic = nblocks, endoff = indexOff; b32(&r, 0x2468ACE0u); b32(&r, 256); bput(&r, &ic, 8); b32(&r, 0); b32(&r, 0); b32(&r, (uint32_t)nct - 1); b32(&r, (uint32_t)ct[nct - 1].len);
              bput(&r, &endoff, 8); b32(&r, 512); b32(&r, 0); b8(&r, 1); b8(&r, 0); b16(&r, (uint16_t)nblocks); fwrite(r.p, 1, r.l, f); free(r.p); }
            fwrite(idx.p, 1, idx.l, f);
            fclose(f); free(hdr.p); free(body.p); free(idx.p);
        }
    }
    printf("{\"prefix\": \"%s\", \"contigs\": %d, \"records\": %zu, \"pairs\": %" PRIu64 ", \"query_bases\": %" PRIu64 ", \"seed\": %" PRIu64 "}\n", prefix, nct, nrec, npairs_total, nbases, seed);
    return 0;
}

/*
 * mdk_extract.c -- host side of the MI355X `MethylDackel extract` path.
 *
 * Division of labour:
 *   host  : BGZF inflate + BAM record framing (mdk_io.c), read admission (the flag/tag/MAPQ tests of
 *           filter_func, common.c:416-444), strand determination (getStrand, common.c:84-116), the
 *           qname pairing that htslib's constructor/destructor callbacks perform (overlaps.c:121-147),
 *           packing into the SoA batch of include/mdk_hip.h, the reference's chunk schedule
 *           (extract.c:325-350, common.c:466-493) and the text post-pass (extract.c:443-510, 39-99).
 *   device: everything per base -- trimming, overlap resolution, context classification, counting.
 * There is no CPU implementation of the per-base work in this library.
 */
#define _GNU_SOURCE
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
#include <unistd.h>
#include <zlib.h>
#include "mdk_extract.h"
#include "mdk_io.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

#define MDK_VERSION "0.6.1"

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
static void sb_put(sbuf *b, const char *s, size_t n) {
    if(b->l + n + 1 > b->m) { b->m = (b->l + n + 1) * 2; b->s = realloc(b->s, b->m); }
    memcpy(b->s + b->l, s, n); b->l += n; b->s[b->l] = 0;
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

struct mdk_plan {
    opts_t o;
    mdk_bam *bam; mdk_bai *bai; int need_seek; mdk_fasta fa; int *fa_of_tid;
    /* schedule cursor (main.c:10-13 globals) */
    uint32_t g_tid, g_pos, g_end, bin;
    int shard_rank, shard_world;       /* interval sharding: this process packs chunk k iff k % world == rank */
    uint64_t n_variant_positions;
    /* stream state */
    int32_t last_tid, last_pos; int at_eof;
    uint8_t *carry; size_t carry_len, carry_cap; int32_t carry_tid;
    uint8_t *carry2; size_t carry2_len, carry2_cap;
    /* chunk pipeline: reader thread -> worker threads -> ordered delivery (see the pipeline section) */
    struct pslot *slot; int n_slot, n_workers; pthread_t reader_th, *worker_th; int started, quit, pipe_rc, reader_done;
    pthread_mutex_t mu; pthread_cond_t cv_free, cv_raw, cv_done;
    uint32_t next_out; int held[2];
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

/* ------------------------------------------------------------------------------------------------ */
/* usage text                                                                                        */
/* ------------------------------------------------------------------------------------------------ */
static void usage(void) {
    fputs("\nUsage: MethylDackel extract [OPTIONS] <ref.fa> <sorted_alignments.bam>\n", stderr);
    fputs("\nOptions (MI355X build; same option surface as MethylDackel 0.6.1):\n"
" -q INT, -p INT, -d INT, -D INT(ignored), -r STR, -o/--opref STR, -@ INT,\n"
" -F/--ignoreFlags INT, -R/--requireFlags INT, --chunkSize INT, --mergeContext,\n"
" --keepDupes, --keepSingleton, --keepDiscordant, --noCpG, --CHG, --CHH,\n"
" --fraction, --counts, --logit, --methylKit, --cytosine_report, --ignoreNH,\n"
" --minOppositeDepth INT, --maxVariantFrac FLOAT, --minConversionEfficiency FLOAT,\n"
" --OT/--OB/--CTOT/--CTOB INT,INT,INT,INT, --nOT/--nOB/--nCTOT/--nCTOB INT,INT,INT,INT,\n"
" -B/--mappabilityBBM FILE, -t/--mappabilityThreshold FLOAT, -b/--minMappableBases INT,\n"
" -M/--mappability FILE, -O, -N FILE, -l FILE, --keepStrand, --version\n"
"\nNote that --fraction, --counts, and --logit are mutually exclusive!\n", stderr);
}

/* 4 comma-separated non-negative ints (the reference's parseBounds, common.c:11-43) */
static void parse_bounds(const char *arg, int *dst) {
    char *dup = strdup(arg), *tok, *end, *save = NULL; int k; int tmp[4];
    for(k = 0, tok = strtok_r(dup, ",", &save); k < 4; k++, tok = strtok_r(NULL, ",", &save)) {
        long v;
        if(!tok) break;
        errno = 0;                   /* the reference tests errno without clearing it (common.c:20-24): a stale errno would reject a literal 0 */
        v = strtol(tok, &end, 10);
        if((errno == ERANGE && (v == LONG_MAX || v == LONG_MIN)) || (errno != 0 && v == 0) || end == tok || v > INT_MAX || v < 0) break;
        tmp[k] = (int)v;
        dst[k] = tmp[k];             /* the reference stores values as it goes, so a bad later field keeps the earlier ones */
    }
    if(k < 4) fprintf(stderr, "Invalid bounds string, %s\n", arg);
    free(dup);
}

/* "chr", "chr:beg", "chr:beg-", "chr:beg-end", "chr:-end" (htslib hts_parse_reg as used at extract.c:1446) */
static const char *parse_region(const char *s, int *beg, int *end) {
    const char *colon = strrchr(s, ':'), *p; long long b = 0, e = 0; int nd = 0;
    if(!colon) { *beg = 0; *end = INT_MAX; return s + strlen(s); }
    p = colon + 1;
    if(*p == '-') {
        for(p++; (*p >= '0' && *p <= '9') || *p == ','; p++) if(*p != ',') { e = e * 10 + (*p - '0'); nd++; }
        if(*p || !nd) return NULL;
        *beg = 0; *end = e > INT_MAX ? INT_MAX : (int)e; return colon;
    }
    for(; (*p >= '0' && *p <= '9') || *p == ','; p++) if(*p != ',') { b = b * 10 + (*p - '0'); nd++; }
    b -= 1;
    if(b < 0) { if((nd && *p == '-') || *p) return NULL; *beg = 0; *end = INT_MAX; return colon; }
    if(*p == 0) e = INT_MAX;
    else if(*p == '-') { for(p++; (*p >= '0' && *p <= '9') || *p == ','; p++) if(*p != ',') e = e * 10 + (*p - '0'); if(*p) return NULL; }
    else return NULL;
    if(e == 0 || e > INT_MAX) e = INT_MAX;
    if(b >= e) return NULL;
    *beg = (int)b; *end = (int)e; return colon;
}

/* ------------------------------------------------------------------------------------------------ */
/* BBM mappability (BBM_Specification.md; loader semantics of extract.c:1236-1339)                   */
/* ------------------------------------------------------------------------------------------------ */
static int load_bbm(mdk_plan *p, FILE *f) {
    uint8_t ver = 0; uint32_t nchrom = 0, c;
    fprintf(stderr, "loading mappability data from %s\n", p->o.bbm_name);
    if(fread(&ver, 1, 1, f) != 1 || ver != 1) { fprintf(stderr, "fatal: %s has wrong BBM version or is malformed\n", p->o.bbm_name); return -10; }
    if(fread(&nchrom, 4, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; }
    p->map_n = nchrom; p->map_names = calloc(nchrom + 1, sizeof(char *)); p->map_len = calloc(nchrom + 1, 4); p->map_bits = calloc(nchrom + 1, sizeof(uint8_t *));
    for(c = 0; c < nchrom; c++) {
        uint16_t nl = 0; uint8_t z = 1; uint32_t len = 0, at = 0; size_t nbytes; double cut = p->o.map_cutoff * 100.0;
        if(fread(&nl, 2, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; }
        p->map_names[c] = calloc((size_t)nl + 1, 1);
        if(nl && fread(p->map_names[c], 1, nl, f) != nl) { printf("fatal: malformed BBM file\n"); return -9; }
        if(fread(&z, 1, 1, f) != 1 || z) { printf("fatal: malformed BBM file\n"); return -9; }
        if(fread(&len, 4, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; }
        p->map_len[c] = len; nbytes = (size_t)len / 8 + ((len % 8) ? 1 : 0);
        p->map_bits[c] = calloc(nbytes + 8, 1);
        while(at < len) {
            uint8_t v; uint32_t run = 1; int above;
            if(fread(&v, 1, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; }
            if(v > 100) {
                if(v == 255) { uint16_t r16 = 0; if(fread(&r16, 2, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; } run = r16; }
                else run = (uint32_t)v - 99u;
                if(fread(&v, 1, 1, f) != 1 || run == 0) { printf("fatal: malformed BBM file\n"); return -9; }
            }
            above = ((double)v >= cut);
            if(above) { uint32_t k, stop = (at + run < len) ? at + run : len; for(k = at; k < stop; k++) p->map_bits[c][k >> 3] |= (uint8_t)(1u << (k & 7)); }
            at += run;
        }
    }
    p->map_on = 1;
    return 0;
}
/* the 0..100 value the reference stores for one bigWig value (extract.c:1137-1144): (char)(raw*100 + 0.5), NaN -> 0 */
static unsigned char map_value(float raw) { if(isnan(raw)) return 0; return (unsigned char)(char)((raw * 100) + 0.5); }

/* -M: mappability from a bigWig (extract.c:1071-1233), optionally re-encoded as BBM (-O / -N).  The run-length writer
 * follows the reference's state machine (runs of 2..155 as [len+99][v], longer as [255][u16 len][v], at most 65535 per run,
 * a trailing run of exactly 155 in the long form) so that the bytes on disk agree. */
static int load_bigwig(mdk_plan *p) {
    opts_t *o = &p->o; mdk_bigwig *bw = mdk_bigwig_open(o->bw_name); FILE *f = NULL; uint32_t c;
    if(!bw) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bw_name); return -4; }
    if(o->out_bbm_name) {
        f = fopen(o->out_bbm_name, "wb");
        if(!f) { fprintf(stderr, "Couldn't open %s for writing! Insufficient permissions?\n", o->out_bbm_name); mdk_bigwig_close(bw); return -7; }
        fputc(1, f);
    }
    fprintf(stderr, "loading mappability data from %s\n", o->bw_name);
    if(f) { uint32_t n = bw->n; fwrite(&n, 4, 1, f); fprintf(stderr, "writing .bbm file to %s\n", o->out_bbm_name); }
    p->map_n = bw->n; p->map_names = calloc(bw->n + 1, sizeof(char *)); p->map_len = calloc(bw->n + 1, 4); p->map_bits = calloc(bw->n + 1, sizeof(uint8_t *));
    for(c = 0; c < bw->n; c++) {
        uint32_t len = bw->len[c], j; float *v = mdk_bigwig_values(bw, c); double cut = o->map_cutoff * 100.0;
        unsigned char last = 255; uint16_t run = 0;
        if(!v) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bw_name); if(f) fclose(f); mdk_bigwig_close(bw); return -4; }
        p->map_names[c] = strdup(bw->name[c]); p->map_len[c] = len; p->map_bits[c] = calloc((size_t)len / 8 + 9, 1);
        if(f) { uint16_t nl = (uint16_t)strlen(bw->name[c]); fwrite(&nl, 2, 1, f); fwrite(bw->name[c], 1, nl, f); fputc(0, f); fwrite(&len, 4, 1, f); }
        for(j = 0; j < len; j++) {
            unsigned char val = map_value(v[j]);
            if(f) {
                if(val == last && run < 65535) run++;
                else {
                    if(run > 1) { if(run < 156) { fputc(run + 99, f); fputc(last, f); } else { fputc(255, f); fwrite(&run, 2, 1, f); fputc(last, f); } run = 0; }
                    if(j + 1 < len && map_value(v[j + 1]) == val) { last = val; run = 1; }
                    else { fputc(val, f); last = val; run = 0; }
                }
            }
            if((double)val >= cut) p->map_bits[c][j >> 3] |= (uint8_t)(1u << (j & 7));
        }
        if(f && run > 1) { if(run < 155) { fputc(run + 99, f); fputc(last, f); } else { fputc(255, f); fwrite(&run, 2, 1, f); fputc(last, f); } }
        free(v);
    }
    if(f) fclose(f);
    mdk_bigwig_close(bw);
    p->map_on = 1;
    return 0;
}

/* number of set bits in [start, start+l) of chromosome c; bits outside the stored array are 0 */
static int64_t map_popcount(const mdk_plan *p, int c, int64_t start, int64_t l) {
    int64_t nbits = ((int64_t)p->map_len[c] / 8 + ((p->map_len[c] % 8) ? 1 : 0)) * 8, end = start + l, cnt = 0, k;
    if(start < 0 || c < 0) return 0;          /* a negative start is a huge uint32 in the reference: past the array */
    if(end > nbits) end = nbits;
    for(k = start; k < end && (k & 7); k++) cnt += (p->map_bits[c][k >> 3] >> (k & 7)) & 1;
    for(; k + 8 <= end; k += 8) cnt += __builtin_popcount(p->map_bits[c][k >> 3]);
    for(; k < end; k++) cnt += (p->map_bits[c][k >> 3] >> (k & 7)) & 1;
    return cnt;
}
/* one window of check_mappability (common.c:305-316): the running counter is a signed char */
static int map_window_passes(const mdk_plan *p, int c, int64_t start, int l) {
    int need = p->o.min_mappable;
    if(l <= 0) return 0;
    if(need <= 0) return 1;
    if(need > 127) return 0;
    return map_popcount(p, c, start, l) >= need;
}

/* ------------------------------------------------------------------------------------------------ */
/* plan open / option surface                                                                        */
/* ------------------------------------------------------------------------------------------------ */
enum { O_NOCPG = 1, O_CHG, O_CHH, O_KEEPDUPES, O_KEEPSINGLETON, O_KEEPDISCORDANT, O_OT, O_OB, O_CTOT, O_CTOB, O_MERGE, O_METHYLKIT,
       O_NOT, O_NOB, O_NCTOT, O_NCTOB, O_MINOPP, O_MAXVARFRAC, O_CHUNKSIZE, O_KEEPSTRAND, O_CYTREPORT, O_MINCONVEFF, O_IGNORENH };


/* ------------------------------------------------------------------------------------------------ */
/* -l FILE / --keepStrand (bed.c)                                                                    */
/* ------------------------------------------------------------------------------------------------ */
/* The reference walks its sorted region list with cursors that only move forward (spanOverlapsBED for chunks and
 * reads, posOverlapsBED for columns; bed.c:22-53).  What those cursors compute is a function of the position alone:
 * the region that governs position x is the FIRST region, in sorted order, that does not end at or before x; x is
 * inside iff that region has started.  build_runs() turns the list into that function -- disjoint runs, each with the
 * strand of its governing region -- once; chunks, reads and (on the device) columns then test against the runs. */
typedef struct { int32_t tid, start, end; int strand; } bedreg;
static int bedreg_order(const void *a, const void *b) {      /* sortBED_func, bed.c:66-80 */
    const bedreg *x = a, *y = b;
    if(x->tid != y->tid) return x->tid < y->tid ? -1 : 1;
    if(x->start != y->start) return x->start < y->start ? -1 : 1;
    if(x->end != y->end) return x->end < y->end ? -1 : 1;
    return (x->strand > y->strand) - (x->strand < y->strand);
}
static size_t skip_field(const char *s, size_t i) { while(s[i] && !isspace((unsigned char)s[i])) i++; return i; }
static size_t skip_blank(const char *s, size_t i) { while(s[i] && isspace((unsigned char)s[i])) i++; return i; }

/* One line of the BED file, by the rules of parseBED (bed.c:118-219): the name ends at the first white-space character;
 * the start is read (scanf %d, so leading blanks are tolerated) right after that ONE separator; the start column is
 * taken to begin there too, so a doubled separator makes the start be read a second time as the end; the strand is the
 * first character of the third column after the end.  Returns 1 = region, 0 = skipped, -1 = error (message printed). */
static int bed_line(char *s, size_t l, int lnum, const char *fn, const mdk_bam *bam, int keep_strand, bedreg *r) {
    size_t a, b, c; int t; char save;
    if(s[0] == '#') return 0;
    a = skip_field(s, 0);
    save = s[a]; s[a] = 0;
    for(t = 0; t < bam->n_targets; t++) if(!strcmp(s, bam->target_name[t])) break;
    if(t == bam->n_targets) {
        if(!strcmp(s, "track") || !strcmp(s, "browser")) return 0;
        fprintf(stderr, "Couldn't properly parse line number %i in %s.\n", lnum, fn);
        return -1;
    }
    s[a] = save;
    r->tid = t; r->start = -1; r->end = -1; r->strand = 0;
    if(a >= l || sscanf(s + a + 1, "%" SCNd32, &r->start) != 1 || r->start == -1) { fprintf(stderr, "Line %" PRId32 " of %s is malformed.\n", (int32_t)lnum, fn); return -1; }
    b = skip_field(s, a + 1);
    if(b >= l || sscanf(s + b + 1, "%" SCNd32, &r->end) != 1 || r->end == -1) { fprintf(stderr, "Line %" PRId32 " of %s is malformed.\n", (int32_t)lnum, fn); return -1; }
    if(r->start >= r->end) { fprintf(stderr, "The position on line %" PRId32 " of %s is incorrect (%" PRId32 " >= %" PRId32 ".\n", (int32_t)lnum, fn, r->start, r->end); return -1; }
    if(r->start < 0) r->start = 0;
    if((int64_t)r->end > (int64_t)bam->target_len[t] + 1) r->end = (int32_t)(bam->target_len[t] + 1);
    if(!keep_strand) return 1;
    c = skip_field(s, b + 1);                                  /* the end column */
    c = skip_blank(s, c); if(!s[c]) return 1; c = skip_field(s, c); if(!s[c]) return 1;      /* column 4 */
    c = skip_blank(s, c); if(!s[c]) return 1; c = skip_field(s, c); if(!s[c]) return 1;      /* column 5 */
    c = skip_blank(s, c);
    if(s[c] == '+') r->strand = 1; else if(s[c] == '-') r->strand = 2;
    return 1;
}

static void build_runs(mdk_plan *p, const bedreg *reg, size_t n) {
    size_t i = 0; int32_t nt = p->bam->n_targets, t;
    p->bed_run = calloc((size_t)nt + 1, sizeof(md_region *)); p->bed_nrun = calloc((size_t)nt + 1, sizeof(int64_t));
    for(t = 0; t < nt; t++) {
        size_t j = i, k; int64_t x = 0, m = 0; md_region *run;
        while(j < n && reg[j].tid == t) j++;
        run = malloc(sizeof(md_region) * (j - i + 1));
        for(k = i; k < j; k++) {
            if((int64_t)reg[k].end <= x) continue;                       /* over before x: never governs anything from here on */
            run[m].start = reg[k].start > x ? reg[k].start : (int32_t)x; run[m].end = reg[k].end; run[m].strand = reg[k].strand; m++;
            x = reg[k].end;
        }
        p->bed_run[t] = run; p->bed_nrun[t] = m; i = j;
    }
}
/* does [beg, end) touch a run of the contig?  (spanOverlapsBED == 1, bed.c:11-41) */
static int bed_touches(const mdk_plan *p, int32_t tid, int64_t beg, int64_t end) {
    const md_region *run = p->bed_run[tid]; int64_t a = 0, b = p->bed_nrun[tid];
    while(a < b) { int64_t m = (a + b) >> 1; if((int64_t)run[m].end <= beg) a = m + 1; else b = m; }
    return a < p->bed_nrun[tid] && (int64_t)run[a].start < end;
}

static int load_bed(mdk_plan *p) {
    const opts_t *o = &p->o; gzFile f; char *data = NULL, *line = NULL; size_t n = 0, cap = 0, at = 0, nreg = 0, creg = 0; bedreg *reg = NULL; int lnum = 0, rc = 0;
    if((f = gzopen(o->bed_name, "r")) == NULL) { fprintf(stderr, "Couldn't open %s for reading.\n", o->bed_name); return -1; }
    for(;;) {
        int got;
        if(cap - n < (1u << 16)) { cap = cap ? cap * 2 : 1u << 20; data = realloc(data, cap); if(!data) { gzclose(f); return -1; } }
        got = gzread(f, data + n, 1u << 16);
        if(got <= 0) break;
        n += (size_t)got;
    }
    gzclose(f);
    while(at < n && rc >= 0) {
        size_t e = at, l; bedreg r;
        while(e < n && data[e] != '\n') e++;
        l = e - at; if(l > 1 && data[e - 1] == '\r') l--;
        if(l == 0) break;                                /* the reference's line loop ends at the first empty line */
        line = realloc(line, l + 2); memcpy(line, data + at, l); line[l] = line[l + 1] = 0;
        at = e + 1; lnum++;
        rc = bed_line(line, strlen(line) < l ? strlen(line) : l, lnum, o->bed_name, p->bam, o->keep_strand, &r);
        if(rc == 1) {
            if(nreg == creg) { creg = creg ? creg * 2 : 1024; reg = realloc(reg, sizeof(bedreg) * creg); }
            reg[nreg++] = r;
        }
    }
    free(line); free(data);
    if(rc < 0) { free(reg); return -1; }
    qsort(reg, nreg, sizeof(bedreg), bedreg_order);
    fprintf(stderr, "Parsed %" PRId32 " regions in %s\n", (int32_t)nreg, o->bed_name);
    build_runs(p, reg, nreg);
    free(reg);
    p->bed_on = 1;
    return 0;
}
int mdk_plan_regions(const mdk_plan *p, int32_t tid, const md_region **runs, int64_t *n) {
    if(!p || !runs || !n || tid < 0 || tid >= p->bam->n_targets) return -1;
    if(!p->bed_on) { *runs = NULL; *n = -1; return 0; }
    *runs = p->bed_run[tid]; *n = p->bed_nrun[tid];
    return 0;
}

static void plan_free(mdk_plan *p);
static void pipeline_stop(mdk_plan *p);
static int pipeline_start(mdk_plan *p);

/* everything after option parsing that `extract` and `mbias` share: inputs, (extract only) mappability and output
 * files, -r, -l.  Frees the plan and returns the reference's code on failure. */
static int plan_attach_inputs(mdk_plan *p, char *argv[], int first_positional) {
    opts_t *o = &p->o; int i; char *oname; FILE *bbm = NULL;
    o->fasta_name = argv[first_positional]; o->bam_name = argv[first_positional + 1];
    if(o->n_threads < 1) o->n_threads = 1;
    p->bam = mdk_bam_open(o->bam_name, o->n_threads);
    if(!p->bam) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bam_name); plan_free(p); return -4; }
    p->bai = getenv("MDK_NO_INDEX") ? NULL : mdk_bai_load(o->bam_name);        /* optional: lets -r and sharded runs skip most of the file */
    if(!o->mbias && !o->perread && o->bbm_name && (bbm = fopen(o->bbm_name, "rb")) == NULL) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bbm_name); plan_free(p); return -8; }
    if(!o->mbias && !o->perread && o->bw_name) { int rc = load_bigwig(p); if(rc) { if(bbm) fclose(bbm); plan_free(p); return rc; } }
    if(bbm) {                  /* as in the reference, a BBM given together with a bigWig replaces the bigWig's bitmaps */
        if(p->map_on) { uint32_t k; for(k = 0; k < p->map_n; k++) { free(p->map_names[k]); free(p->map_bits[k]); } free(p->map_names); free(p->map_len); free(p->map_bits); p->map_names = NULL; p->map_len = NULL; p->map_bits = NULL; p->map_n = 0; }
        { int rc = load_bbm(p, bbm); fclose(bbm); if(rc) { plan_free(p); return rc; } }
    }
    if(mdk_fasta_load(o->fasta_name, &p->fa) != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", o->fasta_name); plan_free(p); return -4; }
    p->fa_of_tid = malloc(sizeof(int) * (size_t)(p->bam->n_targets + 1));
    for(i = 0; i < p->bam->n_targets; i++) p->fa_of_tid[i] = mdk_fasta_find(&p->fa, p->bam->target_name[i]);
    if(p->map_on) {
        p->map_of_tid = malloc(sizeof(int) * (size_t)(p->bam->n_targets + 1));
        for(i = 0; i < p->bam->n_targets; i++) { uint32_t k; p->map_of_tid[i] = -1; for(k = 0; k < p->map_n; k++) if(!strcmp(p->map_names[k], p->bam->target_name[i])) { p->map_of_tid[i] = (int)k; break; } }
    }

    if(o->mbias || o->perread) goto region;
    /* output files and headers (extract.c:1343-1439) */
    if(!o->opref) {
        char *dot; o->opref = strdup(o->bam_name); dot = strrchr(o->opref, '.'); if(dot) *dot = 0;
        fprintf(stderr, "writing to prefix:'%s'\n", o->opref);
    }
    oname = malloc(strlen(o->opref) + 40);
    if(o->cytosine_report) {
        sprintf(oname, "%s.cytosine_report.txt", o->opref);
        p->out[0] = fopen(getenv("MDK_NO_OUTPUT") ? "/dev/null" : oname, "w"); p->out[1] = p->out[2] = p->out[0];
        if(!p->out[0]) { fprintf(stderr, "Couldn't open the output CpG metrics file for writing! Insufficient permissions?\n"); free(oname); plan_free(p); return -3; }
    } else {
        static const char *cn[3] = {"CpG", "CHG", "CHH"};
        for(i = 0; i < 3; i++) {
            const char *ext = o->fraction ? ".meth.bedGraph" : o->counts ? ".counts.bedGraph" : o->logit ? ".logit.bedGraph" : o->methylkit ? ".methylKit" : ".bedGraph";
            if(!o->ctx_on[i]) continue;
            sprintf(oname, "%s_%s%s", o->opref, cn[i], ext);
            p->out[i] = fopen(getenv("MDK_NO_OUTPUT") ? "/dev/null" : oname, "w");     /* MDK_NO_OUTPUT: non-writer rank of a sharded run */
            if(!p->out[i]) { fprintf(stderr, "Couldn't open the output %s metrics file for writing! Insufficient permissions?\n", cn[i]); free(oname); plan_free(p); return -3; }
            if(o->methylkit) fputs("chrBase\tchr\tbase\tstrand\tcoverage\tfreqC\tfreqT\n", p->out[i]);
            else fprintf(p->out[i], "track type=\"bedGraph\" description=\"%s %s%s%s\"\n", o->opref, cn[i], o->merge ? " merged" : "",
                         o->fraction ? " methylation fractions" : o->counts ? " methylation counts" : o->logit ? " logit transformed methylation fractions" : " methylation levels");
        }
    }
    free(oname);
region:
    /* -r (extract.c:1441-1468, MBias.c:497-523) */
    if(o->region) {
        int s = 0, e = 0, t; const char *colon = parse_region(o->region, &s, &e); char *name;
        if(!colon) { fprintf(stderr, "Could not parse the specified region!\n"); plan_free(p); return -4; }
        name = strndup(o->region, (size_t)(colon - o->region));
        for(t = 0; t < p->bam->n_targets; t++) if(!strcmp(p->bam->target_name[t], name)) break;
        free(name);
        if(t == p->bam->n_targets) { fprintf(stderr, "%s did not match a known chromosome/contig name!\n", o->region); plan_free(p); return -6; }
        p->g_tid = (uint32_t)t;
        if(s > 0) p->g_pos = (uint32_t)s;
        if(e > 0) p->g_end = (uint32_t)e;
        if(p->g_end > p->bam->target_len[t]) p->g_end = p->bam->target_len[t];
        p->need_seek = 1;
    }
    /* -l (extract.c:1469-1477, MBias.c:524-532) */
    if(o->bed_name && load_bed(p) != 0) { fprintf(stderr, "There was an error while reading in your BED file!\n"); plan_free(p); return 1; }
    return 0;
}

int mdk_plan_open(int argc, char *argv[], mdk_plan **out) {
    static const struct option longopts[] = {
        {"opref", required_argument, 0, 'o'}, {"fraction", no_argument, 0, 'f'}, {"counts", no_argument, 0, 'c'}, {"logit", no_argument, 0, 'm'},
        {"minDepth", required_argument, 0, 'd'}, {"noCpG", no_argument, 0, O_NOCPG}, {"CHG", no_argument, 0, O_CHG}, {"CHH", no_argument, 0, O_CHH},
        {"keepDupes", no_argument, 0, O_KEEPDUPES}, {"keepSingleton", no_argument, 0, O_KEEPSINGLETON}, {"keepDiscordant", no_argument, 0, O_KEEPDISCORDANT},
        {"OT", required_argument, 0, O_OT}, {"OB", required_argument, 0, O_OB}, {"CTOT", required_argument, 0, O_CTOT}, {"CTOB", required_argument, 0, O_CTOB},
        {"mergeContext", no_argument, 0, O_MERGE}, {"methylKit", no_argument, 0, O_METHYLKIT},
        {"nOT", required_argument, 0, O_NOT}, {"nOB", required_argument, 0, O_NOB}, {"nCTOT", required_argument, 0, O_NCTOT}, {"nCTOB", required_argument, 0, O_NCTOB},
        {"minOppositeDepth", required_argument, 0, O_MINOPP}, {"maxVariantFrac", required_argument, 0, O_MAXVARFRAC}, {"chunkSize", required_argument, 0, O_CHUNKSIZE},
        {"keepStrand", no_argument, 0, O_KEEPSTRAND}, {"cytosine_report", no_argument, 0, O_CYTREPORT}, {"minConversionEfficiency", required_argument, 0, O_MINCONVEFF},
        {"ignoreNH", no_argument, 0, O_IGNORENH}, {"ignoreFlags", required_argument, 0, 'F'}, {"requireFlags", required_argument, 0, 'R'},
        {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"mappability", required_argument, 0, 'M'},
        {"mappabilityThreshold", required_argument, 0, 't'}, {"minMappableBases", required_argument, 0, 'b'},
        {"outputBBMFile", required_argument, 0, 'O'}, {"outputBBMFileName", required_argument, 0, 'N'}, {"mappabilityBBM", required_argument, 0, 'B'},
        {0, 0, 0, 0}};
    mdk_plan *p; opts_t *o; int c;
    *out = NULL;
    p = calloc(1, sizeof(*p)); if(!p) return -5;
    o = &p->o;
    o->ctx_on[0] = 1; o->min_mapq = 10; o->min_phred = 5; o->min_depth = 1; o->ignore_flags = 0xF00;
    o->n_threads = 1; o->chunk_size = 1000000; o->map_cutoff = 0.01f; o->min_mappable = 15;
    p->shard_rank = 0; p->shard_world = 1;
    p->last_tid = -1; p->last_pos = -1; p->carry_tid = -1;

    optind = 1;     /* the reference relies on a fresh process; being a library we reset getopt */
    /* NB -f, -c and -m take an argument in the short-option string although --fraction/--counts/--logit do not
     * (extract.c:796 vs 757-759); kept as is, it is part of the option surface. */
    while((c = getopt_long(argc, argv, "hvq:p:r:l:o:D:f:c:m:d:F:R:@:M:t:b:ON:B:", longopts, NULL)) >= 0) {
        switch(c) {
        case 'h': usage(); plan_free(p); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", MDK_VERSION, "none; methyldackel_amd MI355X build"); plan_free(p); return 0;
        case 'o': free(o->opref); o->opref = strdup(optarg); break;
        case 'D': break;
        case 'd': o->min_depth = atoi(optarg); if(o->min_depth < 1) { fprintf(stderr, "Error, the minimum depth must be at least 1!\n"); plan_free(p); return 1; } break;
        case 'r': o->region = optarg; break;
        case 'l': o->bed_name = optarg; break;
        case O_NOCPG: o->ctx_on[0] = 0; break;
        case O_CHG: o->ctx_on[1] = 1; break;
        case O_CHH: o->ctx_on[2] = 1; break;
        case O_KEEPDUPES: o->keep_dupes = 1; break;
        case O_KEEPSINGLETON: o->keep_singleton = 1; break;
        case O_KEEPDISCORDANT: o->keep_discordant = 1; break;
        case O_OT: case O_OB: case O_CTOT: case O_CTOB: parse_bounds(optarg, o->rel_bounds + 4 * (c - O_OT)); break;
        case O_NOT: case O_NOB: case O_NCTOT: case O_NCTOB: parse_bounds(optarg, o->abs_bounds + 4 * (c - O_NOT)); break;
        case O_MERGE: o->merge = 1; break;
        case O_METHYLKIT: o->methylkit = 1; break;
        case O_MINOPP: o->min_opp_depth = atoi(optarg); break;
        case O_MAXVARFRAC: o->max_variant_frac = atof(optarg); break;
        case O_CHUNKSIZE: o->chunk_size = strtoul(optarg, NULL, 10); if(o->chunk_size < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); plan_free(p); return 1; } break;
        case O_KEEPSTRAND: o->keep_strand = 1; break;
        case O_CYTREPORT: o->cytosine_report = 1; break;
        case O_MINCONVEFF: o->min_conv_eff = (float)atof(optarg); break;
        case O_IGNORENH: o->ignore_nh = 1; break;
        case 'M': o->bw_name = optarg; break;
        case 't': o->map_cutoff = (float)atof(optarg); break;
        case 'b': o->min_mappable = atoi(optarg); break;
        case 'O': o->output_bb = 1; free(o->out_bbm_name); o->out_bbm_name = NULL; break;
        case 'N': o->output_bb = 1; free(o->out_bbm_name); o->out_bbm_name = malloc(strlen(optarg) + 5); sprintf(o->out_bbm_name, "%s.bbm", optarg); break;
        case 'B': o->bbm_name = optarg; break;
        case 'F': o->ignore_flags = atoi(optarg); break;     /* atoi: "0xD00" parses as 0, as in the reference */
        case 'R': o->require_flags = atoi(optarg); break;
        case 'q': o->min_mapq = atoi(optarg); break;
        case 'p': o->min_phred = atoi(optarg); break;
        case 'm': o->logit = 1; break;
        case 'f': o->fraction = 1; break;
        case 'c': o->counts = 1; break;
        case '@': o->n_threads = atoi(optarg); break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); usage(); plan_free(p); return 1;
        }
    }
    if(o->output_bb && !o->out_bbm_name && o->bw_name) {       /* -O: the bigWig's name with its extension replaced by .bbm */
        char *dot; o->out_bbm_name = malloc(strlen(o->bw_name) + 5); strcpy(o->out_bbm_name, o->bw_name);
        dot = strrchr(o->out_bbm_name, '.'); if(dot) *dot = 0;
        strcat(o->out_bbm_name, ".bbm");
    }
    if(o->output_bb && !o->bw_name) { fprintf(stderr, "You must specify a bigWig file when attempting to create a BBM file!\n"); usage(); plan_free(p); return -1; }
    if(argc == 1) { usage(); plan_free(p); return 0; }
    if(argc - optind < 2) {
        if(o->output_bb) o->no_bam = 1;
        else { fprintf(stderr, "You must supply a reference genome in fasta format and an input BAM file!!!\n"); usage(); plan_free(p); return -1; }
    }
    if(o->min_phred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", o->min_phred); o->min_phred = 1; }
    if(o->min_mapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", o->min_mapq); o->min_mapq = 0; }
    if(o->keep_dupes > 0 && (o->ignore_flags & 0x400)) o->ignore_flags -= 0x400;
    if(o->fraction + o->counts + o->logit + o->methylkit + o->cytosine_report > 1) {
        fprintf(stderr, "More than one of --fraction, --counts, --methylKit, --cytosine_report and --logit were specified. These are mutually exclusive.\n");
        usage(); plan_free(p); return 1;
    }
    if(o->methylkit + o->merge == 2) { fprintf(stderr, "--mergeContext and --methylKit are mutually exclusive.\n"); usage(); plan_free(p); return 1; }
    if(o->cytosine_report + o->merge == 2) { fprintf(stderr, "--mergeContext and --cytosine_report are mutually exclusive.\n"); usage(); plan_free(p); return 1; }
    if(o->fraction + o->counts + o->logit > 1) { fprintf(stderr, "You may specify AT MOST one of -c/--counts, -f/--fraction, or -m/--logit.\n"); plan_free(p); return -6; }
    if(!(o->ctx_on[0] + o->ctx_on[1] + o->ctx_on[2])) {
        fprintf(stderr, "You haven't specified any metrics to output!\nEither don't use the --noCpG option or specify --CHG and/or --CHH.\n");
        plan_free(p); return -1;
    }
    if(o->no_bam) {            /* only the bigWig -> BBM conversion was asked for (extract.c:983-994,1217-1230) */
        int rc = load_bigwig(p);
        plan_free(p);
        return rc;
    }

    { int rc = plan_attach_inputs(p, argv, optind); if(rc) return rc; }
    *out = p;
    return 0;
}

static void bb_free(batchbuf *b) { md_host_free(b->seg); md_host_free(b->blob); free(b->ri); free(b->cig); free(b->qn); free(b->pr); memset(b, 0, sizeof(*b)); }
static void plan_free(mdk_plan *p) {
    uint32_t k; int i;
    if(!p) return;
    pipeline_stop(p);               /* the reader and the workers use the BAM reader, the FASTA and the bitmaps: stop them first */
    if(p->bed_run) { for(i = 0; i < p->bam->n_targets; i++) free(p->bed_run[i]); free(p->bed_run); free(p->bed_nrun); }
    if(p->bam) mdk_bam_close(p->bam);
    mdk_bai_free(p->bai);
    mdk_fasta_free(&p->fa); free(p->fa_of_tid); free(p->map_of_tid);
    for(k = 0; k < p->map_n; k++) { free(p->map_names[k]); free(p->map_bits[k]); }
    free(p->map_names); free(p->map_len); free(p->map_bits);
    free(p->carry); free(p->carry2);
    if(p->o.cytosine_report) { if(p->out[0]) fclose(p->out[0]); }
    else for(i = 0; i < 3; i++) if(p->out[i]) fclose(p->out[i]);
    if(p->pr_out && p->pr_out_owned) fclose(p->pr_out);
    for(i = 0; i < 3; i++) { free(p->ob[i].s); free(p->ec.ob[i].s); }
    free(p->o.opref); free(p->o.out_bbm_name); free(p->ref_dev); free(p->ref_tid);
    free(p);
}
void mdk_plan_close(mdk_plan *p) { plan_free(p); }

int mdk_plan_set_shard(mdk_plan *p, int rank, int world) {
    if(!p || world < 1 || rank < 0 || rank >= world) return -1;
    p->shard_rank = rank; p->shard_world = world;
    return 0;
}
int mdk_plan_n_targets(const mdk_plan *p) { return p->bam->n_targets; }
const char *mdk_plan_target_name(const mdk_plan *p, int32_t tid) { return (tid >= 0 && tid < p->bam->n_targets) ? p->bam->target_name[tid] : NULL; }
int64_t mdk_plan_target_len(const mdk_plan *p, int32_t tid) { return (tid >= 0 && tid < p->bam->n_targets) ? (int64_t)p->bam->target_len[tid] : -1; }

void mdk_plan_dev_cfg(const mdk_plan *p, md_dev_cfg *cfg) {
    int i;
    memset(cfg, 0, sizeof(*cfg));
    cfg->keepCpG = p->o.ctx_on[0]; cfg->keepCHG = p->o.ctx_on[1]; cfg->keepCHH = p->o.ctx_on[2];
    cfg->minPhred = p->o.min_phred; cfg->minOppositeDepth = p->o.min_opp_depth > 0 ? p->o.min_opp_depth : 0;
    for(i = 0; i < 16; i++) { cfg->bounds[i] = p->o.rel_bounds[i]; cfg->absoluteBounds[i] = p->o.abs_bounds[i]; }
    cfg->n_slots = 2;
    if(getenv("MDK_TILE")) cfg->tile = atoi(getenv("MDK_TILE"));
}

int mdk_plan_ensure_reference(mdk_plan *p, md_dev *dev, int32_t tid) {
    int i, fi;
    for(i = 0; i < p->n_ref; i++) if(p->ref_dev[i] == dev && p->ref_tid[i] == tid) return 0;
    if(tid < 0 || tid >= p->bam->n_targets || (fi = p->fa_of_tid[tid]) < 0) return MDK_ERR_NOREF;
    i = md_dev_set_reference(dev, tid, p->fa.seq[fi], p->fa.len[fi]);
    if(i) return i;
    if(p->bed_on && !p->o.perread && (i = md_dev_set_regions(dev, tid, p->bed_run[tid], p->bed_nrun[tid])) != 0) return i;     /* perRead uses -l only to pass over chunks (perRead.c:150-166) */
    if(p->n_ref == p->cap_ref) { p->cap_ref = p->cap_ref ? p->cap_ref * 2 : 32; p->ref_dev = realloc(p->ref_dev, sizeof(md_dev *) * p->cap_ref); p->ref_tid = realloc(p->ref_tid, sizeof(int32_t) * p->cap_ref); }
    p->ref_dev[p->n_ref] = dev; p->ref_tid[p->n_ref] = tid; p->n_ref++;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* per-record helpers                                                                                */
/* ------------------------------------------------------------------------------------------------ */
static inline uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline int cigar_is_match(int op) { return op == 0 || op == 7 || op == 8; }
static int32_t cigar_ref_len(const mdk_rec *r) {
    int32_t l = 0; int k;
    for(k = 0; k < r->n_cigar; k++) { uint32_t c = rd_u32(r->cigar + 4 * k); int op = c & 15; if(cigar_is_match(op) || op == 2 || op == 3) l += (int32_t)(c >> 4); }
    return l;
}

/* one pass over the aux area for the two tags the path looks at.  Pointers are to the TYPE byte of the first
 * occurrence, like bam_aux_get; a malformed aux area ends the scan (tags after it are "absent"). */
static void scan_aux(const mdk_rec *r, const uint8_t **nh, const uint8_t **xg) {
    const uint8_t *s = r->aux, *e = r->aux + r->aux_len;
    *nh = *xg = NULL;
    while(e - s >= 3) {
        const uint8_t *ty = s + 2, *v = s + 3; size_t sz;
        switch(*ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': { const uint8_t *z = memchr(v, 0, (size_t)(e - v)); if(!z) return; sz = (size_t)(z - v) + 1; break; }
        case 'B': { size_t es; if(e - v < 5) return; switch(v[0]) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: return; } sz = 5 + es * (size_t)rd_u32(v + 1); break; }
        default: return;
        }
        if((size_t)(e - v) < sz) return;
        if(s[0] == 'N' && s[1] == 'H' && !*nh) *nh = ty;
        else if(s[0] == 'X' && s[1] == 'G' && !*xg) *xg = ty;
        s = v + sz;
    }
}
static int64_t aux_int(const uint8_t *ty) {
    switch(*ty) {
    case 'c': return (int8_t)ty[1]; case 'C': return ty[1];
    case 's': { int16_t v; memcpy(&v, ty + 1, 2); return v; } case 'S': { uint16_t v; memcpy(&v, ty + 1, 2); return v; }
    case 'i': { int32_t v; memcpy(&v, ty + 1, 4); return v; } case 'I': { uint32_t v; memcpy(&v, ty + 1, 4); return v; }
    }
    return 0;
}
/* strand of origin from FLAG and an optional Bismark-style XG tag (common.c:84-116) */
static int strand_of(uint16_t flag, const uint8_t *xg) {
    int conv = 0;    /* 0: no usable XG, 'C' / 'G': converted genome */
    if(xg && (xg[1] == 'C' || xg[1] == 'G')) conv = xg[1];
    if(!conv) {
        if(!(flag & 0x1)) return (flag & 0x10) ? 2 : 1;
        if((flag & 0x50) == 0x50) return 2;
        if(flag & 0x40) return 1;
        if((flag & 0x90) == 0x90) return 1;
        if(flag & 0x80) return 2;
        return 0;
    }
    {   /* orientation classes in the reference's test order (a FLAG with both 0x40 and 0x80 resolves as read #1) */
        int fwdlike;
        if((flag & 0x51) == 0x41) fwdlike = 1;            /* read #1 forward */
        else if((flag & 0x51) == 0x51) fwdlike = 0;       /* read #1 reverse */
        else if((flag & 0x91) == 0x81) fwdlike = 0;       /* read #2 forward */
        else if((flag & 0x91) == 0x91) fwdlike = 1;       /* read #2 reverse */
        else fwdlike = !(flag & 0x10);                    /* single end */
        if(conv == 'C') return fwdlike ? 1 : 3;
        return fwdlike ? 4 : 2;
    }
}

/* --minConversionEfficiency (common.c:338-404).  `win` is the chunk window [woff, woff+wlen) of contig letters. */
static int ctx_code(const char *seq, int64_t len, int64_t i) {   /* 0 none, 1 CpG, 2 CHG, 3 CHH (sign = direction not needed here) */
    char c = seq[i] & 0x5f;
    if(c == 'C') { if(i + 1 < len && (seq[i + 1] & 0x5f) == 'G') return 1; if(i + 2 < len && (seq[i + 2] & 0x5f) == 'G') return 2; return 3; }
    if(c == 'G') { if(i > 0 && (seq[i - 1] & 0x5f) == 'C') return 1; if(i > 1 && (seq[i - 2] & 0x5f) == 'C') return 2; return 3; }
    return 0;
}
static float conv_efficiency(const mdk_rec *r, int strand, int min_phred, const char *win, int64_t woff, int64_t wlen) {
    unsigned nm = 0, nu = 0; int64_t pos = r->pos; int q = 0, k;
    for(k = 0; k < r->n_cigar; k++) {
        uint32_t c = rd_u32(r->cigar + 4 * k); int op = c & 15, len = (int)(c >> 4), j;
        if(cigar_is_match(op)) {
            for(j = 0; j < len; j++, q++) {
                int64_t wi = pos + j - woff; int ctx, b;
                if(pos + j >= woff + wlen) goto done;
                if(wi < 0) continue;                 /* reference reads before its buffer here (UB) */
                ctx = ctx_code(win, wlen, wi);
                if(ctx < 2) continue;                /* CpG and non-C/G positions do not count */
                if(strand == 0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
                if(q >= r->l_qseq || r->qual[q] < min_phred) continue;
                b = (r->seq[q >> 1] >> ((~q & 1) << 2)) & 15;
                if(strand & 1) { if(b == 2) nm++; else if(b == 8) nu++; }
                else { if(b == 4) nm++; else if(b == 1) nu++; }
            }
            /* NB the reference never advances `pos` after an M run (common.c:373-391); keep that */
        } else if(op == 1 || op == 4) q += len;
        else if(op == 2 || op == 3) pos += len;
    }
done:
    if(nm + nu == 0) return 1.0f;
    return nu / ((float)(nm + nu));
}

/* ------------------------------------------------------------------------------------------------ */
/* batch building                                                                                    */
/* ------------------------------------------------------------------------------------------------ */
static int bb_reserve(batchbuf *b, size_t more_reads, size_t more_blob, size_t more_qn, size_t more_cig) {
    if(b->n + more_reads > b->cap_ri) { b->cap_ri = (b->n + more_reads) * 2 + 1024; b->ri = realloc(b->ri, b->cap_ri * sizeof(rinfo)); if(!b->ri) return -1; }
    if(b->blob_len + more_blob > b->cap_blob) {
        size_t nc = (b->blob_len + more_blob) * 2 + (1 << 20); uint8_t *d = md_host_alloc(nc);
        if(!d) return -1;
        if(b->blob_len) memcpy(d, b->blob, b->blob_len);
        md_host_free(b->blob); b->blob = d; b->cap_blob = nc;
    }
    if(b->qn_len + more_qn > b->qn_cap) { b->qn_cap = (b->qn_len + more_qn) * 2 + 65536; b->qn = realloc(b->qn, b->qn_cap); if(!b->qn) return -1; }
    if(b->cig_len + more_cig > b->cig_cap) { b->cig_cap = (b->cig_len + more_cig) * 2 + 4096; b->cig = realloc(b->cig, b->cig_cap * 4); if(!b->cig) return -1; }
    return 0;
}
/* one-shot reservation at the start of a chunk (buffers are empty): no doubling, pinned memory is precious */
static int bb_reserve_exact(batchbuf *b, size_t reads, size_t blob, size_t qn, size_t cig, size_t segs) {
    if(reads > b->cap_ri) { b->cap_ri = reads + reads / 8; b->ri = realloc(b->ri, b->cap_ri * sizeof(rinfo)); if(!b->ri) return -1; }
    if(blob > b->cap_blob) { md_host_free(b->blob); b->cap_blob = blob + blob / 8; b->blob = md_host_alloc(b->cap_blob); if(!b->blob) return -1; }
    if(qn > b->qn_cap) { b->qn_cap = qn + qn / 8; b->qn = realloc(b->qn, b->qn_cap); if(!b->qn) return -1; }
    if(cig > b->cig_cap) { b->cig_cap = cig + cig / 8; b->cig = realloc(b->cig, b->cig_cap * 4); if(!b->cig) return -1; }
    if(segs > b->cap_seg) { md_host_free(b->seg); b->cap_seg = segs + segs / 8; b->seg = md_host_alloc(b->cap_seg * sizeof(md_seg)); if(!b->seg) return -1; }
    return 0;
}
static int seg_reserve(batchbuf *b, size_t more) {
    if(b->n_seg + more > b->cap_seg) {
        size_t nc = (b->n_seg + more) * 2 + 4096; md_seg *d = md_host_alloc(nc * sizeof(md_seg));
        if(!d) return -1;
        if(b->n_seg) memcpy(d, b->seg, b->n_seg * sizeof(md_seg));
        md_host_free(b->seg); b->seg = d; b->cap_seg = nc;
    }
    return 0;
}

static uint64_t hash_str(const char *s) { uint64_t h = 0xcbf29ce484222325ULL; for(; *s; s++) h = (h ^ (uint8_t)*s) * 0x100000001b3ULL; return h ? h : 1; }

/* The qname bookkeeping htslib's pileup does through the constructor/destructor callbacks
 * (overlaps.c:121-147), evaluated lazily per qname.  A buffered read whose end precedes the position of the
 * most recently pulled read has been swept out of the pileup buffer, and its destructor erased the qname key. */
static __thread qent *t_qt = NULL; static __thread size_t t_qt_cap = 0; static __thread int t_gen = 0;     /* one qname table per worker thread; `used` holds the chunk generation */
static __thread struct { int32_t end, next; } *t_side; static __thread size_t t_side_n, t_side_cap;      /* live ends beyond the two kept inline */
static qent *qt_get(const batchbuf *b, uint32_t qoff, uint32_t h) {
    const char *name = b->qn + qoff; size_t mask = t_qt_cap - 1, i = (size_t)h & mask;
    for(;; i = (i + 1) & mask) {
        qent *e = &t_qt[i];
        if(e->used != t_gen) { e->used = t_gen; e->h = h; e->qoff = qoff; e->pending = -1; e->nlive = 0; e->more = 0; return e; }
        if(e->h == h && !strcmp(b->qn + e->qoff, name)) return e;
    }
}
static void qt_prepare(size_t expect) {
    size_t want = 1024;
    while(want < expect + expect / 2 + 16) want <<= 1;
    if(want > t_qt_cap) { free(t_qt); t_qt = calloc(want, sizeof(qent)); t_qt_cap = want; t_gen = 0; }
    if(++t_gen == 0x7fffffff) { size_t i; for(i = 0; i < t_qt_cap; i++) t_qt[i].used = 0; t_gen = 1; }     /* a new generation empties the table */
    t_side_n = 0;
}
static void pair_reads(batchbuf *b, int32_t tid) {
    size_t i, n = b->n; int32_t prev_pos = 0; int first = 1;
    qt_prepare(n);
    for(i = 0; i < n; i++) {
        rinfo *r = &b->ri[i]; int32_t pos = r->pos, end = r->rend; int inserted; qent *e; int k, w, evicted = 0;
        r->mate = -1; r->second = 0;
        /* bam_plp_push: a read enters the buffer iff its end lies beyond the column about to be emitted */
        if(first) inserted = (tid > 0) || (end > 0); else inserted = end > prev_pos;
        if(inserted) {
            e = qt_get(b, r->qn_off, r->qn_hash);
            for(k = 0, w = 0; k < e->nlive; k++) { if(!first && e->live[k] < prev_pos) evicted = 1; else e->live[w++] = e->live[k]; }
            e->nlive = w;
            if(e->more) {       /* drop the swept-out ends of the side list too, refilling the inline slots from it */
                int32_t *link = &e->more;
                while(*link) {
                    int32_t idx = *link - 1;
                    if(!first && t_side[idx].end < prev_pos) { evicted = 1; *link = t_side[idx].next; }
                    else if(e->nlive < 2) { e->live[e->nlive++] = t_side[idx].end; *link = t_side[idx].next; }
                    else link = &t_side[idx].next;
                }
            }
            if(evicted) e->pending = -1;
            if((r->bamflag & 0x1) && !(r->bamflag & 12)) {
                if(e->pending < 0) e->pending = (int32_t)i;
                else { int32_t a = e->pending; b->ri[a].mate = (int32_t)i; r->mate = a; r->second = 1; e->pending = -1; }
            }
            if(e->nlive < 2) e->live[e->nlive++] = end;
            else {
                if(t_side_n == t_side_cap) { t_side_cap = t_side_cap ? t_side_cap * 2 : 1024; t_side = realloc(t_side, sizeof(*t_side) * t_side_cap); }
                t_side[t_side_n].end = end; t_side[t_side_n].next = e->more; e->more = (int32_t)++t_side_n;
            }
        }
        prev_pos = pos; first = 0;
    }
}

/* CIGAR -> gapless runs (reference start, query start, length); what calculate_positions (overlaps.c:27-52) and
 * htslib's resolve_cigar2 compute base by base */
typedef struct { int32_t x, y, l; } run_t;
static int cigar_runs(const uint32_t *cig, int ncig, int32_t pos, int32_t lq, run_t **out, int *cap) {
    int n = 0, k; int32_t x = pos, y = 0;
    for(k = 0; k < ncig; k++) {
        int op = cig[k] & 15; int32_t len = (int32_t)(cig[k] >> 4);
        if(cigar_is_match(op)) {
            int32_t l = len; if(y + l > lq) l = lq - y;            /* malformed CIGAR guard */
            if(l > 0) { if(n == *cap) { *cap = *cap ? *cap * 2 : 16; *out = realloc(*out, sizeof(run_t) * *cap); } (*out)[n].x = x; (*out)[n].y = y; (*out)[n].l = l; n++; }
            x += len; y += len;
        } else if(op == 1 || op == 4) y += len;
        else if(op == 2 || op == 3) x += len;
    }
    return n;
}

/* segments of every read of the chunk: its gapless runs, cut where the overlap partner's runs begin/end.
 * Segments are emitted in ascending order of their reference start: the later runs of a read (after a deletion or a
 * long ref-skip) wait in a small heap until the stream of reads has reached their position, so that the segments
 * overlapping any window of the reference are one tight contiguous run of the array. */
typedef struct { md_seg *v; size_t n, cap; } segheap;
static int heap_push(segheap *h, const md_seg *g) {
    size_t i;
    if(h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 256; h->v = realloc(h->v, h->cap * sizeof(md_seg)); if(!h->v) return -1; }
    for(i = h->n++; i > 0 && h->v[(i - 1) / 2].rpos > g->rpos; i = (i - 1) / 2) h->v[i] = h->v[(i - 1) / 2];
    h->v[i] = *g;
    return 0;
}
static void heap_pop(segheap *h, md_seg *out) {
    size_t i = 0, c; md_seg last;
    *out = h->v[0]; last = h->v[--h->n];
    for(;;) {
        c = 2 * i + 1; if(c >= h->n) break;
        if(c + 1 < h->n && h->v[c + 1].rpos < h->v[c].rpos) c++;
        if(h->v[c].rpos >= last.rpos) break;
        h->v[i] = h->v[c]; i = c;
    }
    if(h->n) h->v[i] = last;
}
static int build_segments(mdk_plan *p, batchbuf *b, int64_t beg, int64_t end) {
    static __thread run_t *ro = NULL, *rm = NULL; static __thread int co = 0, cm = 0; static __thread segheap hp = {NULL, 0, 0};
    size_t i;
    b->n_seg = 0; hp.n = 0;
    for(i = 0; i <= b->n; i++) {
        const rinfo *r, *m = NULL; int no, nm = 0, a, j = 0; uint8_t sf, msf = 0;
        /* everything that starts at or before this read's position can go out now */
        while(hp.n && (i == b->n || hp.v[0].rpos <= b->ri[i].pos)) { if(seg_reserve(b, 1)) return -1; heap_pop(&hp, &b->seg[b->n_seg]); b->n_seg++; }
        if(i == b->n) break;
        r = &b->ri[i];
        sf = (uint8_t)((r->strand & 7) | ((r->bamflag & 0x80) ? MDK_SF_READ2 : 0) | (r->second ? MDK_SF_SECOND : 0));
        no = cigar_runs(b->cig + r->cig_off, r->ncig, r->pos, (int32_t)r->lq, &ro, &co);
        /* only pairs whose strands agree in parity are resolved against each other (overlaps.c:63-65) */
        if(r->mate >= 0 && (((int)r->strand - (int)b->ri[r->mate].strand) & 1) == 0) {
            m = &b->ri[r->mate];
            nm = cigar_runs(b->cig + m->cig_off, m->ncig, m->pos, (int32_t)m->lq, &rm, &cm);
            msf = (uint8_t)((m->strand & 7) | ((m->bamflag & 0x80) ? MDK_SF_READ2 : 0));
        }
        for(a = 0; a < no; a++) {
            int32_t cur = ro[a].x, stop = ro[a].x + ro[a].l;
            while(cur < stop) {
                int32_t pe = stop; int covered = 0; md_seg g;
                while(j < nm && rm[j].x + rm[j].l <= cur) j++;        /* partner runs are ascending, so is cur */
                if(j < nm) { if(rm[j].x <= cur) { covered = 1; if(rm[j].x + rm[j].l < pe) pe = rm[j].x + rm[j].l; } else if(rm[j].x < pe) pe = rm[j].x; }
                if(pe - cur > 65535) pe = cur + 65535;
                if(pe > beg && cur < end) {                         /* pieces wholly outside the counted columns are not needed */
                    g.rpos = cur; g.off4 = r->off4; g.l_qseq = r->lq; g.q0 = (uint32_t)(ro[a].y + (cur - ro[a].x)); g.len = (uint16_t)(pe - cur);
                    g.sf = sf; g.msf = 0; g.m_off4 = 0; g.m_l_qseq = 0; g.m_q0 = 0;
                    if(covered) { g.sf |= MDK_SF_PARTNER; g.msf = msf; g.m_off4 = m->off4; g.m_l_qseq = m->lq; g.m_q0 = (uint32_t)(rm[j].y + (cur - rm[j].x)); }
                    if(cur <= r->pos) { if(seg_reserve(b, 1)) return -1; b->seg[b->n_seg++] = g; }      /* in order already */
                    else if(heap_push(&hp, &g)) return -1;
                }
                cur = pe;
            }
        }
    }
    (void)p;
    return 0;
}

/* admission (filter_func, common.c:416-444) + packing of one candidate record; returns 1 if admitted */
static int admit_and_pack(mdk_plan *p, batchbuf *b, const mdk_rec *r, int32_t rlen, const char *win, int64_t woff, int64_t wlen, int64_t beg, int64_t end) {
    const opts_t *o = &p->o; const uint8_t *nh, *xg; int strand; size_t seqb, seqpad, qualpad, need; uint8_t *d; rinfo *ri; int k;
    if(o->perread) {         /* perRead.c:178-183: alignments that start inside the chunk; flag masks and MAPQ only */
        if(r->pos < beg || r->pos >= end) return 0;
        if(o->require_flags && (o->require_flags & r->flag) != o->require_flags) return 0;
        if(o->ignore_flags && (o->ignore_flags & r->flag) != 0) return 0;
        if(r->mapq < o->min_mapq) return 0;
        scan_aux(r, &nh, &xg);
        strand = strand_of(r->flag, xg);
        goto pack;
    }
    if(r->tid == -1 || (r->flag & 0x4)) return 0;
    if(r->mapq < o->min_mapq) return 0;
    if(r->flag & o->ignore_flags) return 0;
    if(o->require_flags && (r->flag & o->require_flags) != o->require_flags) return 0;
    if(!o->keep_dupes && (r->flag & 0x400)) return 0;
    scan_aux(r, &nh, &xg);
    if(!o->ignore_nh && nh) { int v = (int)aux_int(nh); if(v > 1) return 0; }
    if(p->map_on) {
        int c = p->map_of_tid[r->tid], l = r->l_qseq; int64_t s1, s2;
        if((r->flag & 0x40) || ((r->flag & 0x10) && (r->flag & 0x80))) { s1 = r->pos; s2 = r->mpos; } else { s2 = r->pos; s1 = r->mpos; }
        if(!map_window_passes(p, c, s1, l) && !map_window_passes(p, c, s2, l)) return 0;
    }
    if(!o->keep_singleton && (r->flag & 0x9) == 0x9) return 0;
    if(!o->keep_discordant && (r->flag & 0x3) == 0x1) return 0;
    if(p->bed_on && !bed_touches(p, r->tid, r->pos, (int64_t)r->pos + (rlen > 0 ? rlen : 1))) return 0;      /* common.c:432-439 */
    strand = strand_of(r->flag, xg);
    if(o->min_conv_eff > 0.0) { if(conv_efficiency(r, strand, o->min_phred, win, woff, wlen) < o->min_conv_eff) return 0; }
pack:
    seqb = ((size_t)r->l_qseq + 1) / 2; seqpad = (seqb + 3) & ~(size_t)3; qualpad = ((size_t)r->l_qseq + 3) & ~(size_t)3;
    need = seqpad + qualpad;
    if(bb_reserve(b, 1, need, (size_t)r->l_qname + 1, r->n_cigar)) return -1;
    ri = &b->ri[b->n];
    ri->pos = r->pos; ri->rend = r->pos + rlen; ri->mate = -1; ri->second = 0;
    ri->off4 = (uint32_t)(b->blob_len >> 2); ri->lq = (uint32_t)r->l_qseq; ri->ncig = r->n_cigar; ri->bamflag = r->flag; ri->strand = (uint8_t)strand;
    ri->cig_off = (uint32_t)b->cig_len;
    for(k = 0; k < r->n_cigar; k++) b->cig[b->cig_len++] = rd_u32(r->cigar + 4 * k);
    d = b->blob + b->blob_len;
    memcpy(d, r->seq, seqb); memset(d + seqb, 0, seqpad - seqb); d += seqpad;
    memcpy(d, r->qual, (size_t)r->l_qseq); memset(d + r->l_qseq, 0, qualpad - (size_t)r->l_qseq);
    b->blob_len += need;
    ri->qn_off = (uint32_t)b->qn_len; memcpy(b->qn + b->qn_len, r->qname, r->l_qname); b->qn[b->qn_len + r->l_qname] = 0;
    { uint64_t hh = hash_str(b->qn + b->qn_len); ri->qn_hash = (uint32_t)(hh ^ (hh >> 32)); }        /* while the name is in cache: the pairing pass then only compares names that collide */
    b->qn_len += (size_t)r->l_qname + 1;
    b->algo_bytes += 16 + 4ull * r->n_cigar + seqb + (uint64_t)r->l_qseq;
    b->n++;
    return 1;
}

static int carry_push(uint8_t **buf, size_t *len, size_t *cap, const mdk_rec *r) {
    size_t need = *len + 4 + r->raw_len;
    if(need > *cap) { *cap = need * 2 + 65536; *buf = realloc(*buf, *cap); if(!*buf) return -1; }
    memcpy(*buf + *len, &r->raw_len, 4); memcpy(*buf + *len + 4, r->raw, r->raw_len); *len = need;
    return 0;
}

/* the end of a chunk may not split a CpG / CHG (adjustBounds, common.c:466-493) */
static uint32_t adjust_end(const mdk_plan *p, uint32_t tid, uint32_t end) {
    int fi = p->fa_of_tid[tid]; int64_t L, s, e, n; const char *q;
    if(fi < 0) return end;
    L = p->fa.len[fi]; s = end > 0 ? (int64_t)end - 1 : 0; e = (int64_t)end + 1;      /* inclusive window [s,e], clamped */
    if(s >= L) return end;
    if(e >= L) e = L - 1;
    n = e - s + 1; q = p->fa.seq[fi] + s;
    if(n > 1) {
        if(n > 2 && (q[0] & 0x5f) == 'C' && (q[2] & 0x5f) == 'G') return end + 2;
        if((q[1] & 0x5f) == 'G') return end + 1;
    }
    return end;
}

/* ------------------------------------------------------------------------------------------------ */
/* chunk pipeline                                                                                    */
/*   reader  : walks the reference's chunk schedule over the (block-parallel inflated) BAM stream and copies the      */
/*             raw records of each chunk -- straddlers carried over from the previous chunk first -- into a slot     */
/*   workers : admission, packing, pairing, CIGAR expansion of one chunk each (chunks are independent)               */
/*   consumer: mdk_plan_next_chunk hands the chunks out in schedule order                                            */
/* ------------------------------------------------------------------------------------------------ */
enum { S_FREE = 0, S_FILL, S_RAW, S_WORK, S_DONE, S_HELD };
typedef struct { mdk_slab *slab; size_t beg, end; } rrange;      /* records parsed in place from an inflate slab */
typedef struct pslot {
    int state; mdk_chunk c;
    uint8_t *raw; size_t raw_len, raw_cap;                /* copied records (straddlers from earlier chunks): [u32 len][record bytes]... */
    rrange *rg; int n_rg, cap_rg;                         /* then these ranges of the stream, in order */
    uint64_t n_stream;                                    /* records in the ranges (for up-front reservation) */
    const char *win; int64_t woff, wlen;
    batchbuf bb; int rc;
} pslot;

static int raw_push(pslot *sl, const mdk_rec *r) {
    size_t need = sl->raw_len + 4 + r->raw_len;
    if(need > sl->raw_cap) { sl->raw_cap = need * 2 + (1 << 20); sl->raw = realloc(sl->raw, sl->raw_cap); if(!sl->raw) return -1; }
    memcpy(sl->raw + sl->raw_len, &r->raw_len, 4); memcpy(sl->raw + sl->raw_len + 4, r->raw, r->raw_len); sl->raw_len = need;
    return 0;
}

/* schedule step + raw collection for one chunk; 1 = produced, 0 = schedule finished, <0 error */
static int reader_fill(mdk_plan *p, pslot *sl) {
    const opts_t *o = &p->o; mdk_bam *bam = p->bam; uint32_t tid, beg, end, tmp; int rc, fi, collect; mdk_rec r; size_t off; mdk_chunk *c = &sl->c;
    memset(c, 0, sizeof(*c)); sl->raw_len = 0; sl->n_rg = 0; sl->n_stream = 0; sl->win = NULL; sl->woff = sl->wlen = 0;
    /* extract.c:325-350 */
    c->index = p->bin++;
    tid = p->g_tid; beg = p->g_pos; end = (uint32_t)(beg + o->chunk_size);
    if(tid >= (uint32_t)bam->n_targets) return 0;
    if(p->g_end && end > p->g_end) end = p->g_end;
    if(!o->perread) end = adjust_end(p, tid, end);       /* perRead does not move chunk ends (perRead.c:131-147) */
    if(beg > end) { tmp = beg; beg = end; end = tmp; }
    p->g_pos = end;
    if(p->g_end > 0 && p->g_pos >= p->g_end) p->g_tid = (uint32_t)-1;
    if(p->g_tid != (uint32_t)-1 && p->g_pos >= bam->target_len[tid]) { end = bam->target_len[tid]; p->g_tid++; p->g_pos = 0; }
    if(p->g_end && beg >= p->g_end) return 0;
    c->tid = (int32_t)tid; c->beg = beg; c->end = end;
    if(p->shard_world > 1 && (int)(c->index % (uint32_t)p->shard_world) != p->shard_rank) c->skipped |= MDK_CHUNK_FOREIGN;
    if(p->bed_on && !bed_touches(p, (int32_t)tid, beg, end)) {      /* extract.c:352-369: the chunk is passed over before anything else happens */
        c->skipped |= MDK_CHUNK_BED;
        if(p->bai) { p->need_seek = 1; p->carry_len = 0; p->carry_tid = -1; return 1; }      /* do not even read its records */
    }
    fi = p->fa_of_tid[tid];
    if(c->skipped & MDK_CHUNK_BED) ;
    else if(fi < 0 && o->perread) c->skipped |= MDK_CHUNK_NOREF;       /* perRead.c:176 ignores the failed fetch: every read of the chunk comes out with zero calls */
    else if(fi < 0) {
        if(!(c->skipped & MDK_CHUNK_FOREIGN)) fprintf(stderr, "faidx_fetch_seq returned %i while trying to fetch the sequence for tid %s:%" PRIu32 "-%" PRIu32 "!\n", -2, bam->target_name[tid], beg > 1 ? beg - 2 : 0, end);
        if(!(c->skipped & MDK_CHUNK_FOREIGN)) fprintf(stderr, "Note that the output will be truncated!\n");
        c->skipped |= MDK_CHUNK_NOREF;
    } else {
        if(o->mbias) { sl->woff = beg; sl->wlen = (int64_t)end + 1; }                       /* faidx_fetch_seq(localPos, localEnd), MBias.c:147 */
        else { sl->woff = beg > 1 ? (int64_t)beg - 2 : 0; sl->wlen = (int64_t)end + 10 + 1; }   /* (localPos2, localEnd+10), extract.c:381 */
        if(sl->wlen > p->fa.len[fi]) sl->wlen = p->fa.len[fi];
        sl->wlen -= sl->woff;
        if(sl->wlen < 0) sl->wlen = 0;
        sl->win = p->fa.seq[fi] + sl->woff;
    }
    /* With a .bai the stream is repositioned instead of read through: once at the start of a -r region, and before every
     * own chunk of a sharded run (foreign chunks are then not read at all, like the reference's per-chunk region query). */
    if(p->bai && (p->need_seek || p->shard_world > 1)) {
        p->carry_len = 0; p->carry_tid = -1;
        if(c->skipped & MDK_CHUNK_FOREIGN) return 1;
        {
            uint64_t vo = mdk_bai_start(p->bai, (int32_t)tid, beg);
            if(!vo) { p->need_seek = p->shard_world > 1; p->at_eof = 1; }
            else { rc = mdk_bam_seek(bam, vo); if(rc < 0) { fprintf(stderr, "[mdk] error while reading %s: %s\n", o->bam_name, bam->err); return -2; } p->at_eof = rc == 0; }
            p->last_tid = -1; p->last_pos = -1; p->need_seek = 0;
        }
        if(p->at_eof) { if(p->shard_world <= 1) p->need_seek = 1; return 1; }      /* no records for this chunk */
    }
    /* reads of this chunk, file order: straddlers carried over from the previous chunk, then the stream */
    collect = !c->skipped || (o->perread && c->skipped == MDK_CHUNK_NOREF);      /* perRead still lists the reads of a contig the FASTA lacks (all zero) */
    p->carry2_len = 0;
    if(p->carry_tid == (int32_t)tid) {
        for(off = 0; off < p->carry_len;) {
            uint32_t len; int32_t rlen, endp;
            memcpy(&len, p->carry + off, 4);
            if(mdk_rec_parse(p->carry + off + 4, len, &r) != 0) return -2;
            off += 4 + (size_t)len;
            rlen = cigar_ref_len(&r); endp = r.pos + (rlen > 0 ? rlen : 1);
            c->n_records_seen++;
            if(endp > (int32_t)beg && r.pos < (int32_t)end && collect) { if(raw_push(sl, &r)) return -5; }
            if((uint32_t)endp > end && carry_push(&p->carry2, &p->carry2_len, &p->carry2_cap, &r)) return -5;
        }
    }
    for(;;) {
        mdk_rsum q; const uint8_t *raw;
        rc = mdk_bam_peek_sum(bam, &q, &raw);
        if(rc != 1) break;
        if(q.tid >= 0) {
            if(q.tid < p->last_tid || (q.tid == p->last_tid && q.pos < p->last_pos)) { fprintf(stderr, "[mdk] %s is not coordinate sorted; `extract` needs sorted alignments\n", o->bam_name); return -2; }
            if(q.tid > (int32_t)tid) break;
            if(q.tid == (int32_t)tid && q.pos >= (int32_t)end) break;
            p->last_tid = q.tid; p->last_pos = q.pos;
        }
        if(q.tid == (int32_t)tid) {
            c->n_records_seen++;
            if(q.endp > (int32_t)beg && collect) {          /* in place: extend the open range or start a new one */
                size_t roff; mdk_slab *cs = mdk_bam_cur_slab(bam, &roff); rrange *g = sl->n_rg ? &sl->rg[sl->n_rg - 1] : NULL;
                if(g && g->slab == cs && g->end == roff) g->end = roff + 4 + q.len;
                else {
                    if(sl->n_rg == sl->cap_rg) { sl->cap_rg = sl->cap_rg ? sl->cap_rg * 2 : 16; sl->rg = realloc(sl->rg, sizeof(rrange) * sl->cap_rg); if(!sl->rg) return -5; }
                    g = &sl->rg[sl->n_rg++]; g->slab = cs; g->beg = roff; g->end = roff + 4 + q.len; mdk_slab_ref(bam, cs);
                }
                sl->n_stream++;
            }
            if((uint32_t)q.endp > end) { r.raw = raw; r.raw_len = q.len; if(carry_push(&p->carry2, &p->carry2_len, &p->carry2_cap, &r)) return -5; }
        }
        mdk_bam_advance_sum(bam, &q);
    }
    if(rc < 0) { fprintf(stderr, "[mdk] error while reading %s: %s\n", o->bam_name, bam->err); return -2; }
    { uint8_t *t = p->carry; size_t tc = p->carry_cap; p->carry = p->carry2; p->carry_len = p->carry2_len; p->carry_cap = p->carry2_cap; p->carry2 = t; p->carry2_cap = tc; p->carry2_len = 0; p->carry_tid = (int32_t)tid; }
    return 1;
}

/* admission + packing + pairing + segments of one chunk */
static int worker_process(mdk_plan *p, pslot *sl) {
    batchbuf *b = &sl->bb; mdk_chunk *c = &sl->c; size_t off; mdk_rec r; double t0 = now_s(), t1, t2;
    int g; size_t bytes = sl->raw_len;
    b->n = 0; b->blob_len = 0; b->qn_len = 0; b->cig_len = 0; b->n_seg = 0; b->algo_bytes = 0;
    /* one reservation per chunk instead of growing (the blob is pinned memory, which is expensive to allocate): the
     * payload, names and CIGARs of the admitted reads are all smaller than the raw records they come from */
    for(g = 0; g < sl->n_rg; g++) bytes += sl->rg[g].end - sl->rg[g].beg;
    if(bb_reserve_exact(b, (size_t)sl->n_stream + c->n_records_seen + 16, bytes - bytes / 8 + 65536, bytes / 4 + 4096, bytes / 16 + 4096, 2 * (size_t)sl->n_stream + 4096)) return -5;
    for(off = 0; off < sl->raw_len;) {
        uint32_t len; memcpy(&len, sl->raw + off, 4);
        if(mdk_rec_parse(sl->raw + off + 4, len, &r) != 0) return -2;
        off += 4 + (size_t)len;
        if(admit_and_pack(p, b, &r, cigar_ref_len(&r), sl->win, sl->woff, sl->wlen, c->beg, c->end) < 0) return -5;
    }
    for(g = 0; g < sl->n_rg; g++) {
        const uint8_t *base = sl->rg[g].slab->buf;
        for(off = sl->rg[g].beg; off < sl->rg[g].end;) {
            uint32_t len; memcpy(&len, base + off, 4);
            if(mdk_rec_parse(base + off + 4, len, &r) != 0) return -2;
            off += 4 + (size_t)len;
            if(admit_and_pack(p, b, &r, cigar_ref_len(&r), sl->win, sl->woff, sl->wlen, c->beg, c->end) < 0) return -5;
        }
        mdk_slab_unref(p->bam, sl->rg[g].slab);
    }
    sl->n_rg = 0;
    if(bb_reserve(b, 1, 16, 16, 1) || seg_reserve(b, 1)) return -5;          /* never hand out NULL arrays */
    t1 = now_s();
    if(p->o.perread) {        /* no pairing, no segments: the device walks each read's CIGAR itself */
        size_t i;
        if(b->cap_pr < b->n + 1) { b->cap_pr = (b->n + 1) * 2; free(b->pr); b->pr = malloc(sizeof(md_pr_read) * b->cap_pr); if(!b->pr) return -5; }
        for(i = 0; i < b->n; i++) {
            const rinfo *ri = &b->ri[i]; md_pr_read *q = &b->pr[i];
            q->pos = ri->pos; q->off4 = ri->off4; q->l_qseq = ri->lq; q->cig_off = ri->cig_off; q->n_cigar = ri->ncig; q->strand = ri->strand; q->reserved = 0;
        }
        c->pr.tid = c->tid; c->pr.beg = c->beg; c->pr.end = c->end; c->pr.n_reads = (int32_t)b->n; c->pr.read = b->pr; c->pr.cigar = b->cig; c->pr.n_cigar = b->cig_len;
        c->pr.blob = b->blob; c->pr.blob_bytes = b->blob_len; c->host = b;
        pthread_mutex_lock(&p->mu); p->t_collect += t1 - t0; pthread_mutex_unlock(&p->mu);
        return 0;
    }
    if(!p->o.mbias) pair_reads(b, c->tid);          /* mbias installs no overlap handler (MBias.c:158-161): every read counts on its own */
    t2 = now_s();
    if(build_segments(p, b, c->beg, c->end)) return -5;
    c->batch.tid = c->tid; c->batch.beg = c->beg; c->batch.end = c->end; c->batch.n_segs = (int32_t)b->n_seg; c->batch.seg = b->seg;
    c->batch.blob = b->blob; c->batch.blob_bytes = b->blob_len; c->batch.n_reads = (int32_t)b->n; c->batch.algo_bytes = b->algo_bytes;
    pthread_mutex_lock(&p->mu); p->t_collect += t1 - t0; p->t_pair += t2 - t1; p->t_segs += now_s() - t2; pthread_mutex_unlock(&p->mu);
    return 0;
}

static void *reader_main(void *arg) {
    mdk_plan *p = arg;
    for(;;) {
        pslot *sl = NULL; int i, rc; double t0 = now_s(), t1;
        pthread_mutex_lock(&p->mu);
        while(!p->quit) { for(i = 0; i < p->n_slot; i++) if(p->slot[i].state == S_FREE) { sl = &p->slot[i]; break; } if(sl) break; pthread_cond_wait(&p->cv_free, &p->mu); }
        if(p->quit) { pthread_mutex_unlock(&p->mu); break; }
        sl->state = S_FILL;
        pthread_mutex_unlock(&p->mu);
        t1 = now_s();
        rc = reader_fill(p, sl);
        pthread_mutex_lock(&p->mu);
        p->t_rwait += t1 - t0; p->t_rfill += now_s() - t1;
        if(rc == 1) { sl->state = S_RAW; pthread_cond_signal(&p->cv_raw); }
        else { sl->state = S_FREE; if(rc < 0) p->pipe_rc = rc; p->reader_done = 1; pthread_cond_broadcast(&p->cv_raw); pthread_cond_broadcast(&p->cv_done); }
        pthread_mutex_unlock(&p->mu);
        if(rc != 1) break;
    }
    return NULL;
}
static void *worker_main(void *arg) {
    mdk_plan *p = arg;
    for(;;) {
        pslot *sl = NULL; int i, rc; uint32_t best = 0; double tw0 = now_s();
        pthread_mutex_lock(&p->mu);
        for(;;) {
            sl = NULL;
            for(i = 0; i < p->n_slot; i++) if(p->slot[i].state == S_RAW && (!sl || p->slot[i].c.index < best)) { sl = &p->slot[i]; best = sl->c.index; }
            if(sl || p->quit || p->reader_done) break;
            pthread_cond_wait(&p->cv_raw, &p->mu);
        }
        if(!sl) { pthread_mutex_unlock(&p->mu); break; }       /* nothing left and the reader has finished (or we are quitting) */
        sl->state = S_WORK; p->t_widle += now_s() - tw0;
        pthread_mutex_unlock(&p->mu);
        tw0 = now_s();
        rc = worker_process(p, sl);
        pthread_mutex_lock(&p->mu);
        p->t_wbusy += now_s() - tw0;
        sl->rc = rc; sl->state = S_DONE; if(rc < 0 && !p->pipe_rc) p->pipe_rc = rc;
        pthread_cond_broadcast(&p->cv_done);
        pthread_mutex_unlock(&p->mu);
    }
    free(t_qt); t_qt = NULL; t_qt_cap = 0; free(t_side); t_side = NULL; t_side_cap = t_side_n = 0;
    return NULL;
}
static int pipeline_start(mdk_plan *p) {
    int i;
    /* beyond a dozen workers the serial reader is the limit, and every slot pins ~1.2 bytes of host memory per raw BAM
     * byte of its chunk (expensive to allocate), so the pipeline depth is bounded; -@ still sizes the inflate pool */
    p->n_workers = p->o.n_threads < 1 ? 1 : p->o.n_threads;
    { int cap = getenv("MDK_WORKERS") ? atoi(getenv("MDK_WORKERS")) : 12; if(cap < 1) cap = 1; if(p->n_workers > cap) p->n_workers = cap; }
    p->n_slot = p->n_workers + 3;
    p->slot = calloc((size_t)p->n_slot, sizeof(pslot));
    p->worker_th = calloc((size_t)p->n_workers, sizeof(pthread_t));
    if(!p->slot || !p->worker_th) return -5;
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_free, NULL); pthread_cond_init(&p->cv_raw, NULL); pthread_cond_init(&p->cv_done, NULL);
    p->held[0] = p->held[1] = -1; p->next_out = 0; p->started = 1;
    pthread_create(&p->reader_th, NULL, reader_main, p);
    for(i = 0; i < p->n_workers; i++) pthread_create(&p->worker_th[i], NULL, worker_main, p);
    return 0;
}
static void pipeline_stop(mdk_plan *p) {
    int i;
    if(!p->started) return;
    pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->cv_free); pthread_cond_broadcast(&p->cv_raw); pthread_cond_broadcast(&p->cv_done); pthread_mutex_unlock(&p->mu);
    mdk_bam_abort(p->bam);          /* wake the reader if it is waiting for inflated data */
    pthread_join(p->reader_th, NULL);
    for(i = 0; i < p->n_workers; i++) pthread_join(p->worker_th[i], NULL);
    for(i = 0; i < p->n_slot; i++) { int g; for(g = 0; g < p->slot[i].n_rg; g++) mdk_slab_unref(p->bam, p->slot[i].rg[g].slab); bb_free(&p->slot[i].bb); free(p->slot[i].raw); free(p->slot[i].rg); }
    free(p->slot); free(p->worker_th); p->slot = NULL; p->started = 0;
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_free); pthread_cond_destroy(&p->cv_raw); pthread_cond_destroy(&p->cv_done);
}

int mdk_plan_next_chunk(mdk_plan *p, mdk_chunk *c) {
    int i, found = -1, rc = 0;
    if(!p->started && pipeline_start(p)) return -5;
    pthread_mutex_lock(&p->mu);
    /* the chunk handed out two calls ago is no longer referenced by the caller: recycle its buffers */
    if(p->held[1] >= 0) { p->slot[p->held[1]].state = S_FREE; pthread_cond_signal(&p->cv_free); }
    p->held[1] = p->held[0]; p->held[0] = -1;
    for(;;) {
        int active = 0;
        for(i = 0; i < p->n_slot; i++) {
            int st = p->slot[i].state;
            if(st == S_DONE && p->slot[i].c.index == p->next_out) { found = i; break; }
            if(st == S_FILL || st == S_RAW || st == S_WORK || st == S_DONE) active = 1;
        }
        if(found >= 0) break;
        if(p->pipe_rc < 0) { rc = p->pipe_rc; break; }
        if(p->reader_done && !active) { rc = 0; break; }
        pthread_cond_wait(&p->cv_done, &p->mu);
    }
    if(found >= 0) {
        pslot *sl = &p->slot[found];
        if(sl->rc < 0) rc = sl->rc; else { *c = sl->c; rc = 1; }
        sl->state = S_HELD; p->held[0] = found; p->next_out++;
    }
    pthread_mutex_unlock(&p->mu);
    if(rc <= 0) memset(c, 0, sizeof(*c));
    return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* text post-pass (extract.c:443-510 driving writeCall/processLast, extract.c:39-99,207-222)          */
/* ------------------------------------------------------------------------------------------------ */
static void put_site(mdk_plan *p, sbuf *dst, const char *chrom, int32_t pos, int width, uint32_t m, uint32_t u, int ref_is_c, const char *cctx, const char *tri) {
    const opts_t *o = &p->o; char line[10000]; int n; uint32_t cov = m + u;      /* the size of writeCall's buffer (extract.c:40): lines longer than that are cut the same way */
    if(cov < (uint32_t)o->min_depth && !o->cytosine_report) return;
    if(o->fraction) n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%f\n", chrom, pos, pos + width, ((double)m) / cov);
    else if(o->counts) n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%i\n", chrom, pos, pos + width, cov);
    else if(o->logit) { double f = ((double)m) / cov; n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%f\n", chrom, pos, pos + width, log(f) - log(1 - f)); }
    else if(o->methylkit) n = snprintf(line, sizeof(line), "%s.%i\t%s\t%i\t%c\t%i\t%6.2f\t%6.2f\n", chrom, pos + 1, chrom, pos + 1, ref_is_c ? 'F' : 'R', cov, 100.0 * ((double)m) / cov, 100.0 * ((double)u) / cov);
    else if(o->cytosine_report) n = snprintf(line, sizeof(line), "%s\t%i\t%c\t%" PRIu32 "\t%" PRIu32 "\tC%s\t%s\n", chrom, pos + 1, ref_is_c ? '+' : '-', m, u, cctx, tri);
    else n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%i\t%" PRIu32 "\t%" PRIu32 "\n", chrom, pos, pos + width, (int)(100.0 * ((double)m) / cov), m, u);
    if(n > 0) sb_put(dst, line, (size_t)n < sizeof(line) ? (size_t)n : sizeof(line) - 1);
}

/* trinucleotide context string of a C (direction +1) or G (direction -1) at contig index i (extract.c:120-180) */
static const char *trinuc(const char *seq, int64_t len, int64_t i, int dir, char out[4]) {
    static const char comp[256] = {['A'] = 'T', ['a'] = 'T', ['C'] = 'G', ['c'] = 'G', ['G'] = 'C', ['g'] = 'C', ['T'] = 'A', ['t'] = 'A'};
    int k;
    out[0] = 'C'; out[3] = 0;
    for(k = 1; k <= 2; k++) {
        int64_t j = i + (int64_t)k * dir; char ch = 'N';
        if(j >= 0 && j < len) { ch = seq[j]; if(dir < 0) ch = comp[(uint8_t)ch] ? comp[(uint8_t)ch] : 'N'; else { ch &= 0x5f; if(ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ch = 'N'; } }
        out[k] = ch;
    }
    return out;
}
static const char *cctx_name(int type) { return type == 0 ? "G" : type == 1 ? "HG" : "HH"; }

/* zero-coverage rows of --cytosine_report between *from and upto (extract.c:182-205) */
static void put_blanks(mdk_plan *p, sbuf *dst, const char *chrom, const char *seq, int64_t len, int64_t *from, int64_t upto) {
    char tri[4];
    for(; *from < upto; (*from)++) {
        int code; int dir, type;
        if(*from >= len) continue;
        code = ctx_code(seq, len, *from);
        if(!code) continue;
        type = code - 1;
        if(!p->o.ctx_on[type]) continue;
        dir = ((seq[*from] & 0x5f) == 'C') ? 1 : -1;
        put_site(p, dst, chrom, (int32_t)*from, 1, 0, 0, dir > 0, cctx_name(type), trinuc(seq, len, *from, dir, tri));
    }
}

/* text of one chunk (variant filter, --mergeContext, formats; extract.c:443-510) into e->ob[]; touches nothing shared */
static void emit_format(mdk_plan *p, const mdk_chunk *c, const md_sites *s, emit_ctx *e) {
    const opts_t *o = &p->o; const char *chrom; int64_t i; int fi; const char *seq = NULL; int64_t slen = 0, blank_from;
    char tri[4];
    e->ob[0].l = e->ob[1].l = e->ob[2].l = 0; e->n_variant = 0; e->lastcpg_tid = e->lastchg_tid = -1;
    if(c->skipped & (MDK_CHUNK_NOREF | MDK_CHUNK_BED)) return;
    chrom = p->bam->target_name[c->tid];
    fi = p->fa_of_tid[c->tid]; if(fi >= 0) { seq = p->fa.seq[fi]; slen = p->fa.len[fi]; }
    blank_from = c->beg;
    for(i = 0; i < s->n_sites; i++) {
        int32_t pos = (int32_t)s->site[i].pos; uint32_t m = s->site[i].nmeth, u = s->site[i].nunmeth; int type = (s->site[i].meta >> 1) & 3, is_g = s->site[i].meta & 1;
        if(o->min_opp_depth > 0 && s->var) {
            uint32_t noff = s->var[i].noff, nvar = s->var[i].nvar;
            if(noff >= (uint32_t)o->min_opp_depth && ((double)nvar) / ((double)noff) >= o->max_variant_frac) {
                e->n_variant++;
                if(o->merge && is_g) {
                    if(type == 0 && e->lastcpg_tid == c->tid && e->lastcpg_pos == pos - 1) { e->lastcpg_m = 0; e->lastcpg_u = 0; }
                    else if(type == 1 && e->lastchg_tid == c->tid && e->lastchg_pos == pos - 2) { e->lastchg_m = 0; e->lastchg_u = 0; }
                }
                continue;
            }
        }
        if(m + u == 0 && !o->cytosine_report) continue;
        if(!o->merge || type == 2) {
            if(o->cytosine_report) {
                put_blanks(p, &e->ob[0], chrom, seq, slen, &blank_from, pos);
                put_site(p, &e->ob[0], chrom, pos, 1, m, u, !is_g, cctx_name(type), trinuc(seq, slen, pos, is_g ? -1 : 1, tri));
                blank_from = (int64_t)pos + 1;
            } else put_site(p, &e->ob[type], chrom, pos, 1, m, u, !is_g, NULL, NULL);
        } else if(type == 0) {
            int32_t key = is_g ? pos - 1 : pos;
            if(e->lastcpg_tid == c->tid && e->lastcpg_pos == key) { put_site(p, &e->ob[0], chrom, key, 2, m + e->lastcpg_m, u + e->lastcpg_u, !is_g, NULL, NULL); e->lastcpg_tid = -1; }
            else {
                if(e->lastcpg_tid != -1) put_site(p, &e->ob[0], p->bam->target_name[e->lastcpg_tid], e->lastcpg_pos, 2, e->lastcpg_m, e->lastcpg_u, !is_g, NULL, NULL);
                e->lastcpg_tid = c->tid; e->lastcpg_pos = key; e->lastcpg_m = m; e->lastcpg_u = u;
            }
        } else {
            int32_t key = is_g ? pos - 2 : pos;
            if(e->lastchg_tid == c->tid && e->lastchg_pos == key) { put_site(p, &e->ob[1], chrom, key, 3, m + e->lastchg_m, u + e->lastchg_u, !is_g, NULL, NULL); e->lastchg_tid = -1; }
            else {
                if(e->lastchg_tid != -1) put_site(p, &e->ob[1], p->bam->target_name[e->lastchg_tid], e->lastchg_pos, 3, e->lastchg_m, e->lastchg_u, !is_g, NULL, NULL);
                e->lastchg_tid = c->tid; e->lastchg_pos = key; e->lastchg_m = m; e->lastchg_u = u;
            }
        }
    }
    if(o->merge) {      /* pending sites never cross a chunk boundary (extract.c:496-507) */
        if(o->ctx_on[0] && e->lastcpg_tid != -1) { put_site(p, &e->ob[0], p->bam->target_name[e->lastcpg_tid], e->lastcpg_pos, 2, e->lastcpg_m, e->lastcpg_u, 1, NULL, NULL); e->lastcpg_tid = -1; }
        if(o->ctx_on[1] && e->lastchg_tid != -1) { put_site(p, &e->ob[1], p->bam->target_name[e->lastchg_tid], e->lastchg_pos, 3, e->lastchg_m, e->lastchg_u, 1, NULL, NULL); e->lastchg_tid = -1; }
    } else if(o->cytosine_report) put_blanks(p, &e->ob[0], chrom, seq, slen, &blank_from, c->end);
}
/* append a formatted chunk to the output files (ordered flush, extract.c:514-535) */
static void emit_write(mdk_plan *p, emit_ctx *e) {
    int k;
    if(p->o.cytosine_report) { if(e->ob[0].l) fputs(e->ob[0].s, p->out[0]); }
    else for(k = 0; k < 3; k++) if(p->o.ctx_on[k] && e->ob[k].l) fputs(e->ob[k].s, p->out[k]);
    p->n_variant_positions += e->n_variant;
}

int mdk_plan_emit(mdk_plan *p, const mdk_chunk *c, const md_sites *s) {
    double te0 = now_s();
    if(c->index != p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    p->next_emit++;
    emit_format(p, c, s, &p->ec);
    emit_write(p, &p->ec);
    p->t_emit += now_s() - te0;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* extract_main's emitter: chunks are formatted by a few threads and written in chunk order          */
/* ------------------------------------------------------------------------------------------------ */
enum { EJ_FREE = 0, EJ_READY, EJ_BUSY };
typedef struct { int state; mdk_chunk c; md_sites s; md_site *site; md_site_var *var; int64_t cap; emit_ctx e; } ejob;
typedef struct {
    mdk_plan *p; ejob *job; int n_job, n_th; pthread_t *th; pthread_mutex_t mu; pthread_cond_t cv_job, cv_free, cv_turn;
    uint32_t next_write; int quit; double t_format;
} emitter;
static void *emitter_main(void *arg) {
    emitter *E = arg;
    for(;;) {
        ejob *j = NULL; int i; double t0;
        pthread_mutex_lock(&E->mu);
        for(;;) {
            for(i = 0; i < E->n_job; i++) if(E->job[i].state == EJ_READY && (!j || E->job[i].c.index < j->c.index)) j = &E->job[i];
            if(j || E->quit) break;
            pthread_cond_wait(&E->cv_job, &E->mu);
        }
        if(!j) { pthread_mutex_unlock(&E->mu); break; }
        j->state = EJ_BUSY;
        pthread_mutex_unlock(&E->mu);
        t0 = now_s();
        emit_format(E->p, &j->c, &j->s, &j->e);
        pthread_mutex_lock(&E->mu);
        E->t_format += now_s() - t0;
        while(E->next_write != j->c.index) pthread_cond_wait(&E->cv_turn, &E->mu);
        emit_write(E->p, &j->e);                 /* in turn, so under the lock: nobody else may write now anyway */
        E->next_write++; j->state = EJ_FREE;
        pthread_cond_broadcast(&E->cv_turn); pthread_cond_signal(&E->cv_free);
        pthread_mutex_unlock(&E->mu);
    }
    return NULL;
}
static int emitter_start(emitter *E, mdk_plan *p, int n_th) {
    int i;
    memset(E, 0, sizeof(*E));
    E->p = p; E->n_th = n_th < 1 ? 1 : n_th; E->n_job = E->n_th + 2; E->next_write = p->next_emit;
    E->job = calloc((size_t)E->n_job, sizeof(ejob)); E->th = calloc((size_t)E->n_th, sizeof(pthread_t));
    if(!E->job || !E->th) return -5;
    pthread_mutex_init(&E->mu, NULL); pthread_cond_init(&E->cv_job, NULL); pthread_cond_init(&E->cv_free, NULL); pthread_cond_init(&E->cv_turn, NULL);
    for(i = 0; i < E->n_th; i++) pthread_create(&E->th[i], NULL, emitter_main, E);
    return 0;
}
/* hand a chunk and its sites over (both are copied: the caller's buffers are recycled) */
static int emitter_push(emitter *E, const mdk_chunk *c, const md_sites *s) {
    ejob *j = NULL; int i;
    if(c->index != E->p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    E->p->next_emit++;
    pthread_mutex_lock(&E->mu);
    for(;;) { for(i = 0; i < E->n_job; i++) if(E->job[i].state == EJ_FREE) { j = &E->job[i]; break; } if(j) break; pthread_cond_wait(&E->cv_free, &E->mu); }
    j->state = EJ_BUSY;                          /* being filled */
    pthread_mutex_unlock(&E->mu);
    j->c = *c; j->s = *s;
    if(s->n_sites > j->cap) {
        j->cap = s->n_sites + s->n_sites / 4 + 1024; free(j->site); free(j->var);
        j->site = malloc(sizeof(md_site) * (size_t)j->cap); j->var = malloc(sizeof(md_site_var) * (size_t)j->cap);
        if(!j->site || !j->var) { fprintf(stderr, "[mdk] out of memory while queueing a chunk for output\n"); abort(); }      /* nothing sensible can be written in order any more */
    }
    if(s->n_sites) { memcpy(j->site, s->site, sizeof(md_site) * (size_t)s->n_sites); if(s->var) memcpy(j->var, s->var, sizeof(md_site_var) * (size_t)s->n_sites); }
    j->s.site = j->site; j->s.var = s->var ? j->var : NULL;
    pthread_mutex_lock(&E->mu); j->state = EJ_READY; pthread_cond_signal(&E->cv_job); pthread_mutex_unlock(&E->mu);
    return 0;
}
static void emitter_stop(emitter *E) {
    int i;
    if(!E->th) return;
    pthread_mutex_lock(&E->mu);
    while(E->next_write != E->p->next_emit) pthread_cond_wait(&E->cv_turn, &E->mu);       /* everything handed over has been written */
    E->quit = 1; pthread_cond_broadcast(&E->cv_job);
    pthread_mutex_unlock(&E->mu);
    for(i = 0; i < E->n_th; i++) pthread_join(E->th[i], NULL);
    for(i = 0; i < E->n_job; i++) { int k; free(E->job[i].site); free(E->job[i].var); for(k = 0; k < 3; k++) free(E->job[i].e.ob[k].s); }
    E->p->t_emit += E->t_format;
    free(E->job); free(E->th); E->th = NULL;
}


int mdk_plan_finish(mdk_plan *p) {
    int i;
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk host] inflate+frame+admit+pack %.3fs (inflate alone %.3fs)  pairing %.3fs  segments %.3fs  emit %.3fs; records found in the inflate threads' tables %" PRIu64 ", by walking %" PRIu64 "; reader: scanning %.3fs, waiting for a free slot %.3fs; workers busy %.3fs idle %.3fs (sum over %d)\n", p->t_collect, p->bam->t_inflate, p->t_pair, p->t_segs, p->t_emit, p->bam->n_fast, p->bam->n_slow, p->t_rfill, p->t_rwait, p->t_wbusy, p->t_widle, p->n_workers);
    if(p->n_variant_positions) printf("%" PRIu64 " positions were excluded due to likely being variants.\n", p->n_variant_positions);
    if(p->o.cytosine_report) { if(p->out[0]) fclose(p->out[0]); p->out[0] = p->out[1] = p->out[2] = NULL; }
    else for(i = 0; i < 3; i++) if(p->out[i]) { fclose(p->out[i]); p->out[i] = NULL; }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* the drop-in entry point                                                                           */
/* ------------------------------------------------------------------------------------------------ */
/* The `MethylDackel` command asks (MDK_FAST_EXIT) to leave with _exit once the outputs are closed, skipping the unpinning
 * of buffers and the HIP shutdown.  Not under a profiler or another injected tool: those finalise at normal exit. */
static int fast_exit_wanted(void) {
    const char *pre = getenv("LD_PRELOAD");
    if(!getenv("MDK_FAST_EXIT")) return 0;
    if(getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCPROF_OUTPUT_PATH")) return 0;
    if(pre && (strstr(pre, "rocprof") || strstr(pre, "roctx") || strstr(pre, "rocm"))) return 0;
    return 1;
}
/* leave now: outputs are flushed and closed.  When the `MethylDackel` command runs the work in a child process (main.c),
 * MDK_DONE_FD names the pipe on which the parent waits for the result: it is told first, and the standard streams are
 * closed, so that nobody waits for the kernel to unpin ~1 GB of staging buffers and tear the GPU context down. */
static void leave_fast(int ret) {
    const char *fd = getenv("MDK_DONE_FD");
    fflush(stdout); fflush(stderr);
    if(fd) { int f = atoi(fd), rc = ret; if(f > 2 && write(f, &rc, sizeof(rc)) == (ssize_t)sizeof(rc)) { close(f); close(0); close(1); close(2); } }
    _exit(ret & 0xff);
}
typedef struct { int device; md_dev_cfg cfg; md_dev *dev; int rc; char err[512]; } devopen_t;
/* the HIP runtime takes 0.1-0.4 s to come up: start that before anything else (options, BAM header, FASTA), on its own thread */
static void *hipwarm_main(void *arg) { (void)arg; (void)md_dev_count(); return NULL; }
/* only in the command's child process, which always ends with _exit: a library caller whose bad command line makes us return
 * at once must not find a half-initialised runtime racing its exit handlers */
static void hip_warm_up(void) { pthread_t th; if(getenv("MDK_DONE_FD") && !pthread_create(&th, NULL, hipwarm_main, NULL)) pthread_detach(th); }
/* md_dev_last_error is per thread: keep the text of a failed open for the thread that reports it */
static void *devopen_main(void *arg) { devopen_t *d = arg; d->rc = md_dev_open(d->device, &d->cfg, &d->dev); if(d->rc) snprintf(d->err, sizeof(d->err), "%s", md_dev_last_error()); return NULL; }

int extract_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; mdk_chunk ch[2]; int have[2] = {0, 0}; int rc, k = 0, ret = 0, more = 1; devopen_t dop; pthread_t dth; emitter em;
    double T0 = now_s(), t_open, t_dev, w_next = 0, w_sub = 0, w_down = 0, w_emit = 0, ta;
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open(argc, argv, &p);
    t_open = now_s() - T0;
    if(rc != 0 || !p) return rc;
    /* HIP initialisation takes a few hundred ms: do it while the host pipeline already inflates and packs */
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    pthread_create(&dth, NULL, devopen_main, &dop);
    if(!p->started && pipeline_start(p)) { pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    pthread_join(dth, NULL);
    t_dev = now_s() - T0;
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `extract`.\n", dop.device, dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    if(emitter_start(&em, p, p->o.n_threads >= 8 ? 8 : p->o.n_threads)) { md_dev_close(dev); mdk_plan_close(p); return -5; }
    /* two chunks in flight: build+submit chunk k while chunk k-1 finishes on the device, then hand k-1 to the emitter */
    while(more || have[0] || have[1]) {
        int cur = k & 1, prev = cur ^ 1;
        if(more) {
            ta = now_s();
            rc = mdk_plan_next_chunk(p, &ch[cur]);
            w_next += now_s() - ta;
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) more = 0;
            else {
                if(!ch[cur].skipped) {
                    ta = now_s();
                    rc = mdk_plan_ensure_reference(p, dev, ch[cur].tid);
                    if(!rc) rc = md_dev_submit(dev, cur, &ch[cur].batch);
                    w_sub += now_s() - ta;
                    if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                }
                have[cur] = 1;
            }
        }
        if(have[prev]) {
            md_sites sites; memset(&sites, 0, sizeof(sites));
            if(!ch[prev].skipped) {
                ta = now_s();
                rc = md_dev_download(dev, prev, &sites);
                w_down += now_s() - ta;
                if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            }
            ta = now_s();
            if(emitter_push(&em, &ch[prev], &sites)) { ret = MDK_RC_DEVICE; break; }
            w_emit += now_s() - ta;
            have[prev] = 0;
        }
        k++;
        if(!more && !have[0] && !have[1]) break;
    }
    { double tw = now_s(); emitter_stop(&em); w_emit += now_s() - tw; }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] plan open %.3fs, device ready at %.3fs, loop: wait-for-chunk %.3fs submit %.3fs download %.3fs emit %.3fs, total %.3fs\n", t_open, t_dev, w_next, w_sub, w_down, w_emit, now_s() - T0);
    if(ret == 0) mdk_plan_finish(p);
    if(fast_exit_wanted()) leave_fast(ret);
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}

/* ------------------------------------------------------------------------------------------------ */
/* mbias (MBias.c): the same schedule, admission and segments; the device accumulates a histogram    */
/* over (strand, read number, position in read) across all chunks, which is read back once.          */
/* ------------------------------------------------------------------------------------------------ */
static void mbias_usage(void) {
    fputs("\nUsage: MethylDackel mbias [OPTIONS] <ref.fa> <sorted_alignments.bam> <output.prefix>\n", stderr);
    fputs("\nOptions (MI355X build; same option surface as MethylDackel 0.6.1):\n"
" -q INT, -p INT, -D INT(ignored), -r STR, -l FILE, --keepStrand, -@ INT, --chunkSize INT,\n"
" --keepDupes, --keepSingleton, --keepDiscordant, -F/--ignoreFlags INT, -R/--requireFlags INT,\n"
" --ignoreNH, --minConversionEfficiency FLOAT, --txt, --noSVG (implies --txt; no prefix needed),\n"
" --noCpG, --CHG, --CHH, --nOT/--nOB/--nCTOT/--nCTOB INT,INT,INT,INT, --version\n", stderr);
}

int mdk_plan_open_mbias(int argc, char *argv[], mdk_plan **out) {
    enum { M_NOCPG = 1, M_CHG, M_CHH, M_KEEPDUPES, M_KEEPSINGLETON, M_KEEPDISCORDANT, M_TXT, M_NOSVG, M_NOT, M_NOB, M_NCTOT, M_NCTOB,
           M_CHUNKSIZE, M_KEEPSTRAND, M_MINCONVEFF, M_IGNORENH };
    static const struct option longopts[] = {            /* MBias.c:330-352 */
        {"noCpG", no_argument, 0, M_NOCPG}, {"CHG", no_argument, 0, M_CHG}, {"CHH", no_argument, 0, M_CHH}, {"keepDupes", no_argument, 0, M_KEEPDUPES},
        {"keepSingleton", no_argument, 0, M_KEEPSINGLETON}, {"keepDiscordant", no_argument, 0, M_KEEPDISCORDANT}, {"txt", no_argument, 0, M_TXT},
        {"noSVG", no_argument, 0, M_NOSVG}, {"nOT", required_argument, 0, M_NOT}, {"nOB", required_argument, 0, M_NOB}, {"nCTOT", required_argument, 0, M_NCTOT},
        {"nCTOB", required_argument, 0, M_NCTOB}, {"chunkSize", required_argument, 0, M_CHUNKSIZE}, {"keepStrand", no_argument, 0, M_KEEPSTRAND},
        {"minConversionEfficiency", required_argument, 0, M_MINCONVEFF}, {"ignoreNH", no_argument, 0, M_IGNORENH},
        {"ignoreFlags", required_argument, 0, 'F'}, {"requireFlags", required_argument, 0, 'R'}, {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'},
        {0, 0, 0, 0}};
    mdk_plan *p; opts_t *o; int c;
    *out = NULL;
    p = calloc(1, sizeof(*p)); if(!p) return -5;
    o = &p->o;
    o->mbias = 1; o->svg = 1;
    o->ctx_on[0] = 1; o->min_mapq = 10; o->min_phred = 5; o->min_depth = 1; o->ignore_flags = 0xF00; o->n_threads = 1; o->chunk_size = 1000000;
    p->shard_rank = 0; p->shard_world = 1;
    p->last_tid = -1; p->last_pos = -1; p->carry_tid = -1;
    optind = 1;
    while((c = getopt_long(argc, argv, "hvq:p:r:l:D:F:@:", longopts, NULL)) >= 0) {      /* NB no R: in the short options (MBias.c:353) */
        switch(c) {
        case 'h': mbias_usage(); plan_free(p); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", MDK_VERSION, "none; methyldackel_amd MI355X build"); plan_free(p); return 0;
        case 'D': break;
        case 'r': o->region = optarg; break;
        case 'l': o->bed_name = optarg; break;
        case M_NOCPG: o->ctx_on[0] = 0; break;
        case M_CHG: o->ctx_on[1] = 1; break;
        case M_CHH: o->ctx_on[2] = 1; break;
        case M_KEEPDUPES: o->keep_dupes = 1; break;       /* unlike extract, 0x400 stays in ignoreFlags, so this alone changes nothing */
        case M_KEEPSINGLETON: o->keep_singleton = 1; break;
        case M_KEEPDISCORDANT: o->keep_discordant = 1; break;
        case M_TXT: o->txt = 1; break;
        case M_NOSVG: o->svg = 0; o->txt = 1; break;
        case M_NOT: case M_NOB: case M_NCTOT: case M_NCTOB: parse_bounds(optarg, o->abs_bounds + 4 * (c - M_NOT)); break;
        case M_CHUNKSIZE: o->chunk_size = strtoul(optarg, NULL, 10); if(o->chunk_size < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); plan_free(p); return 1; } break;
        case M_KEEPSTRAND: o->keep_strand = 1; break;
        case M_MINCONVEFF: o->min_conv_eff = (float)atof(optarg); break;
        case M_IGNORENH: o->ignore_nh = 1; break;
        case 'F': o->ignore_flags = atoi(optarg); break;
        case 'R': o->require_flags = atoi(optarg); break;
        case 'q': o->min_mapq = atoi(optarg); break;
        case 'p': o->min_phred = atoi(optarg); break;
        case '@': o->n_threads = atoi(optarg); break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); mbias_usage(); plan_free(p); return 1;
        }
    }
    if(argc == 1) { mbias_usage(); plan_free(p); return 0; }
    if((o->svg && argc - optind != 3) || (!o->svg && argc - optind < 2)) {
        fprintf(stderr, "You must supply a reference genome in fasta format, an input BAM file, and an output prefix!!!\n");
        mbias_usage(); plan_free(p); return -1;
    }
    if(o->min_phred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", o->min_phred); o->min_phred = 1; }
    if(o->min_mapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", o->min_mapq); o->min_mapq = 0; }
    if(!(o->ctx_on[0] + o->ctx_on[1] + o->ctx_on[2])) {
        fprintf(stderr, "You haven't specified any metrics to output!\nEither don't use the --noCpG option or specify --CHG and/or --CHH.\n");
        plan_free(p); return -1;
    }
    if(o->svg) o->mb_opref = argv[optind + 2];
    { int rc = plan_attach_inputs(p, argv, optind); if(rc) return rc; }
    *out = p;
    return 0;
}
int mdk_plan_mbias_outputs(const mdk_plan *p, const char **opref, int *svg, int *txt, int *which) {
    if(!p || !p->o.mbias) return -1;
    if(opref) *opref = p->o.mb_opref;
    if(svg) *svg = p->o.svg;
    if(txt) *txt = p->o.txt;
    if(which) *which = p->o.ctx_on[0] + 2 * p->o.ctx_on[1] + 4 * p->o.ctx_on[2];
    return 0;
}

int mbias_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; mdk_chunk ch; int rc, k = 0, ret = 0; devopen_t dop; pthread_t dth; md_mbias hist;
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open_mbias(argc, argv, &p);
    if(rc != 0 || !p) return rc;
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    pthread_create(&dth, NULL, devopen_main, &dop);
    if(!p->started && pipeline_start(p)) { pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    pthread_join(dth, NULL);
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `mbias`.\n", dop.device, dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    for(;; k++) {
        /* chunk k goes to slot k&1; the batch handed out two calls ago is recycled by the next call, so its upload must be over */
        if((rc = md_dev_slot_sync(dev, k & 1)) != 0) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
        rc = mdk_plan_next_chunk(p, &ch);
        if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
        if(rc == 0) break;
        if(ch.skipped & MDK_CHUNK_NOREF) { ret = -4; break; }        /* the reference's worker gives up here and its caller then crashes (MBias.c:150-155,543) */
        if(ch.skipped) continue;
        rc = mdk_plan_ensure_reference(p, dev, ch.tid);
        if(!rc) rc = md_dev_mbias_submit(dev, k & 1, &ch.batch);
        if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
    }
    if(ret == 0) {
        rc = md_dev_mbias_read(dev, &hist);
        if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
        if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; }
        else if(mdk_mbias_report(&hist, p->o.mb_opref, p->o.svg, p->o.txt, p->o.ctx_on[0] + 2 * p->o.ctx_on[1] + 4 * p->o.ctx_on[2])) ret = -3;
    }
    if(fast_exit_wanted()) leave_fast(ret);
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}

/* ------------------------------------------------------------------------------------------------ */
/* perRead (perRead.c): chunks without adjustBounds; the reads that start in a chunk and pass the    */
/* flag/MAPQ tests go to the device, which walks each CIGAR (k_perread); one text line per read.     */
/* ------------------------------------------------------------------------------------------------ */
static void perread_usage(void) {
    fputs("\nUsage: MethylDackel perRead [OPTIONS] <ref.fa> <input>\n", stderr);
    fputs("\nOutput columns: read name, chromosome, position, CpG methylation (%), number of informative bases.\n"
"Options (MI355X build; same option surface as MethylDackel 0.6.1):\n"
" -q INT, -p INT, -r STR, -l FILE, --keepStrand, -o STR, -F/--ignoreFlags INT (default 0),\n"
" -R/--requireFlags INT, -@ INT, --chunkSize INT, --version\n", stderr);
}

int mdk_plan_open_perread(int argc, char *argv[], mdk_plan **out) {
    static const struct option longopts[] = {            /* perRead.c:300-308; --ignoreNH is in the help text only */
        {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"chunkSize", required_argument, 0, 19}, {"keepStrand", no_argument, 0, 20},
        {"ignoreFlags", required_argument, 0, 'F'}, {"requireFlags", required_argument, 0, 'R'}, {0, 0, 0, 0}};
    mdk_plan *p; opts_t *o; int c;
    *out = NULL;
    p = calloc(1, sizeof(*p)); if(!p) return -5;
    o = &p->o;
    o->perread = 1;
    o->ctx_on[0] = 1; o->min_mapq = 10; o->min_phred = 5; o->min_depth = 1; o->ignore_flags = 0; o->n_threads = 1; o->chunk_size = 1000000;
    p->shard_rank = 0; p->shard_world = 1;
    p->last_tid = -1; p->last_pos = -1; p->carry_tid = -1;
    p->pr_out = stdout;
    optind = 1;
    while((c = getopt_long(argc, argv, "hvq:p:o:@:r:l:F:R:", longopts, NULL)) >= 0) {
        switch(c) {
        case 'h': perread_usage(); plan_free(p); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", MDK_VERSION, "none; methyldackel_amd MI355X build"); plan_free(p); return 0;
        case 'o':
            if(p->pr_out_owned) fclose(p->pr_out);
            if((p->pr_out = fopen(optarg, "w")) == NULL) { fprintf(stderr, "Couldn't open %s for writing\n", optarg); p->pr_out_owned = 0; plan_free(p); return 2; }
            p->pr_out_owned = 1;
            break;
        case 'q': o->min_mapq = atoi(optarg); break;
        case 'p': o->min_phred = atoi(optarg); break;
        case '@': o->n_threads = atoi(optarg); break;
        case 'r': o->region = optarg; break;
        case 'l': o->bed_name = optarg; break;
        case 'F': o->ignore_flags = atoi(optarg); break;
        case 'R': o->require_flags = atoi(optarg); break;
        case 19: o->chunk_size = strtoul(optarg, NULL, 10); if(o->chunk_size < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); plan_free(p); return 1; } break;
        case 20: o->keep_strand = 1; break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); perread_usage(); plan_free(p); return 1;
        }
    }
    if(argc == 1) { perread_usage(); plan_free(p); return 0; }
    if(argc - optind != 2) { fprintf(stderr, "You must supply a reference genome in fasta format and a BAM or CRAM file\n"); perread_usage(); plan_free(p); return -1; }
    if(o->min_phred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", o->min_phred); o->min_phred = 1; }
    if(o->min_mapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", o->min_mapq); o->min_mapq = 0; }
    /* the reference opens the FASTA first (-2 with the usage text), then the BAM (-4) (perRead.c:386-396) */
    { FILE *f = fopen(argv[optind], "r"); if(!f) { fprintf(stderr, "Couldn't open the index for %s!\n", argv[optind]); perread_usage(); plan_free(p); return -2; } fclose(f); }
    { int rc = plan_attach_inputs(p, argv, optind); if(rc) return rc; }
    *out = p;
    return 0;
}

int mdk_plan_emit_perread(mdk_plan *p, const mdk_chunk *c, const md_pr_count *counts, int64_t n) {
    const batchbuf *b; const char *chrom; int64_t i; char line[10000]; sbuf *ob;
    if(!p || !c || !p->o.perread) return -1;
    if(c->index != p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    p->next_emit++;
    if(c->skipped & ~MDK_CHUNK_NOREF) return 0;
    b = c->host;
    if(!b || (int64_t)b->n != c->pr.n_reads) return -2;
    if(counts && n != c->pr.n_reads) return -2;
    if(!counts && !(c->skipped & MDK_CHUNK_NOREF) && c->pr.n_reads) return -2;
    chrom = p->bam->target_name[c->tid];
    ob = &p->ob[0]; ob->l = 0;
    for(i = 0; i < c->pr.n_reads; i++) {             /* addRead, perRead.c:16-36 */
        uint32_t m = counts ? counts[i].nmeth : 0, u = counts ? counts[i].nunmeth : 0; int l;
        const char *qn = b->qn + b->ri[i].qn_off;
        if(m + u > 0) l = snprintf(line, sizeof(line), "%s\t%s\t%" PRId64 "\t%f\t%" PRIu32 "\n", qn, chrom, (int64_t)b->ri[i].pos, 100. * ((double)m) / (m + u), m + u);
        else l = snprintf(line, sizeof(line), "%s\t%s\t%" PRId64 "\t0.0\t%" PRIu32 "\n", qn, chrom, (int64_t)b->ri[i].pos, m + u);
        if(l >= (int)sizeof(line)) l = (int)sizeof(line) - 1;
        sb_put(ob, line, (size_t)l);
    }
    if(ob->l) fputs(ob->s, p->pr_out);
    ob->l = 0;
    return 0;
}

int perRead_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; mdk_chunk ch[2]; int have[2] = {0, 0}; int rc, k = 0, ret = 0, more = 1; devopen_t dop; pthread_t dth;
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open_perread(argc, argv, &p);
    if(rc != 0 || !p) return rc;
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    pthread_create(&dth, NULL, devopen_main, &dop);
    if(!p->started && pipeline_start(p)) { pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    pthread_join(dth, NULL);
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `perRead`.\n", dop.device, dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    while(more || have[0] || have[1]) {       /* two chunks in flight, as in extract_main */
        int cur = k & 1, prev = cur ^ 1;
        if(more) {
            rc = mdk_plan_next_chunk(p, &ch[cur]);
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) more = 0;
            else {
                if(!ch[cur].skipped && ch[cur].pr.n_reads) {
                    rc = mdk_plan_ensure_reference(p, dev, ch[cur].tid);
                    if(!rc) rc = md_dev_perread_submit(dev, cur, &ch[cur].pr);
                    if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                }
                have[cur] = 1;
            }
        }
        if(have[prev]) {
            const md_pr_count *cnt = NULL; int64_t n = 0;
            if(!ch[prev].skipped && ch[prev].pr.n_reads) {
                rc = md_dev_perread_download(dev, prev, &cnt, &n);
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            }
            if(mdk_plan_emit_perread(p, &ch[prev], cnt, n)) { ret = MDK_RC_DEVICE; break; }
            have[prev] = 0;
        }
        k++;
        if(!more && !have[0] && !have[1]) break;
    }
    fflush(p->pr_out);
    if(fast_exit_wanted()) { if(p->pr_out_owned) fclose(p->pr_out); leave_fast(ret); }
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}

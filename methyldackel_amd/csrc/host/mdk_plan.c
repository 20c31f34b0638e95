/* mdk_plan.c -- options, inputs and lifetime of the plan object (see mdk_plan.h). */
#include "mdk_plan.h"

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
MDK_LOCAL void parse_bounds(const char *arg, int *dst) {
    char *dup = xstrdup(arg), *tok, *end, *save = NULL; int k; int tmp[4];
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
    p->map_n = nchrom; p->map_names = xcalloc(nchrom + 1, sizeof(char *)); p->map_len = xcalloc(nchrom + 1, 4); p->map_bits = xcalloc(nchrom + 1, sizeof(uint8_t *));
    for(c = 0; c < nchrom; c++) {
        uint16_t nl = 0; uint8_t z = 1; uint32_t len = 0, at = 0; size_t nbytes; double cut = p->o.map_cutoff * 100.0;
        if(fread(&nl, 2, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; }
        p->map_names[c] = xcalloc((size_t)nl + 1, 1);
        if(nl && fread(p->map_names[c], 1, nl, f) != nl) { printf("fatal: malformed BBM file\n"); return -9; }
        if(fread(&z, 1, 1, f) != 1 || z) { printf("fatal: malformed BBM file\n"); return -9; }
        if(fread(&len, 4, 1, f) != 1) { printf("fatal: malformed BBM file\n"); return -9; }
        p->map_len[c] = len; nbytes = (size_t)len / 8 + ((len % 8) ? 1 : 0);
        p->map_bits[c] = xcalloc(nbytes + 8, 1);
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
/* the 0..100 value the reference stores for one bigWig value (extract.c:1137-1144): the float goes into a DOUBLE first (`double val_raw =
 * vals->value[j]`), then (char)(val_raw*100 + 0.5), NaN -> 0.  The arithmetic is in double: 0.005f is 0.00499999988..., so it becomes 0, while
 * a product formed in float would round to 0.5 and give 1 (tests/test_bigwig_edges.py holds the hand-computed values). */
static unsigned char map_value(float raw) { const double val_raw = raw; if(isnan(raw)) return 0; return (unsigned char)(char)((val_raw * 100) + 0.5); }

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
    p->map_n = bw->n; p->map_names = xcalloc(bw->n + 1, sizeof(char *)); p->map_len = xcalloc(bw->n + 1, 4); p->map_bits = xcalloc(bw->n + 1, sizeof(uint8_t *));
    for(c = 0; c < bw->n; c++) {
        uint32_t len = bw->len[c], j; float *v = mdk_bigwig_values(bw, c); double cut = o->map_cutoff * 100.0;
        unsigned char last = 255; uint16_t run = 0;
        if(!v) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bw_name); if(f) fclose(f); mdk_bigwig_close(bw); return -4; }
        p->map_names[c] = xstrdup(bw->name[c]); p->map_len[c] = len; p->map_bits[c] = xcalloc((size_t)len / 8 + 9, 1);
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
MDK_LOCAL int map_window_passes(const mdk_plan *p, int c, int64_t start, int l) {
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
    p->bed_run = xcalloc((size_t)nt + 1, sizeof(md_region *)); p->bed_nrun = xcalloc((size_t)nt + 1, sizeof(int64_t));
    for(t = 0; t < nt; t++) {
        size_t j = i, k; int64_t x = 0, m = 0; md_region *run;
        while(j < n && reg[j].tid == t) j++;
        run = xmalloc(sizeof(md_region) * (j - i + 1));
        for(k = i; k < j; k++) {
            if((int64_t)reg[k].end <= x) continue;                       /* over before x: never governs anything from here on */
            run[m].start = reg[k].start > x ? reg[k].start : (int32_t)x; run[m].end = reg[k].end; run[m].strand = reg[k].strand; m++;
            x = reg[k].end;
        }
        p->bed_run[t] = run; p->bed_nrun[t] = m; i = j;
    }
}
/* does [beg, end) touch a run of the contig?  (spanOverlapsBED == 1, bed.c:11-41) */
MDK_LOCAL int bed_touches(const mdk_plan *p, int32_t tid, int64_t beg, int64_t end) {
    const md_region *run = p->bed_run[tid]; int64_t a = 0, b = p->bed_nrun[tid];
    while(a < b) { int64_t m = (a + b) >> 1; if((int64_t)run[m].end <= beg) a = m + 1; else b = m; }
    return a < p->bed_nrun[tid] && (int64_t)run[a].start < end;
}

static int load_bed(mdk_plan *p) {
    const opts_t *o = &p->o; gzFile f; char *data = NULL, *line = NULL; size_t n = 0, cap = 0, at = 0, nreg = 0, creg = 0; bedreg *reg = NULL; int lnum = 0, rc = 0;
    if((f = gzopen(o->bed_name, "r")) == NULL) { fprintf(stderr, "Couldn't open %s for reading.\n", o->bed_name); return -1; }
    for(;;) {
        int got;
        if(cap - n < (1u << 16)) { cap = cap ? cap * 2 : 1u << 20; data = xrealloc(data, cap); if(!data) { gzclose(f); return -1; } }
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
        line = xrealloc(line, l + 2); memcpy(line, data + at, l); line[l] = line[l + 1] = 0;
        at = e + 1; lnum++;
        rc = bed_line(line, strlen(line) < l ? strlen(line) : l, lnum, o->bed_name, p->bam, o->keep_strand, &r);
        if(rc == 1) {
            if(nreg == creg) { creg = creg ? creg * 2 : 1024; reg = xrealloc(reg, sizeof(bedreg) * creg); }
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


/* everything after option parsing that `extract` and `mbias` share: inputs, (extract only) mappability and output
 * files, -r, -l.  Frees the plan and returns the reference's code on failure. */
/* the FASTA is read (and compacted) on a thread of its own while the BAM is opened: 128 MB take as long as everything else here together */
typedef struct { const char *fn; mdk_fasta *fa; int rc; } faload_t;
static void *faload_main(void *arg) { faload_t *f = arg; f->rc = mdk_fasta_load(f->fn, f->fa); return NULL; }
MDK_LOCAL int plan_attach_inputs(mdk_plan *p, char *argv[], int first_positional) {
    opts_t *o = &p->o; int i; char *oname; FILE *bbm = NULL; faload_t fl; pthread_t fth; int fth_ok;
    o->fasta_name = argv[first_positional]; o->bam_name = argv[first_positional + 1];
    if(o->n_threads < 1) o->n_threads = 1;
    fl.fn = o->fasta_name; fl.fa = &p->fa; fl.rc = 0;
    fth_ok = pthread_create(&fth, NULL, faload_main, &fl) == 0;
#define FA_JOIN() do { if(fth_ok) { pthread_join(fth, NULL); fth_ok = 0; } } while(0)
    /* staging memory (the reader's slabs, the batch buffers) is huge-page memory that libmdk_hip registers with the runtime the
     * first time an upload reads from it (md_host_alloc): allocating it never waits for the device to come up, and every upload
     * is a DMA off the submitting thread -- the pinned double-buffered feed of the north star, for inputs of any size */
    md_host_set_pinned(0);
    /* the test hook that leaves every piece after the header's to the device only makes sense where a device will be attached */
    if(o->mbias || o->perread || getenv("MDK_HOST_INFLATE") || getenv("MDK_HOST_PREP")) unsetenv("MDK_DEVICE_INFLATE_ONLY");
    p->bam = mdk_bam_open(o->bam_name, o->n_threads);
    if(!p->bam) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bam_name); FA_JOIN(); plan_free(p); return -4; }
    p->bai = getenv("MDK_NO_INDEX") ? NULL : mdk_bai_load(o->bam_name);        /* optional: lets -r and sharded runs skip most of the file */
    if(!o->mbias && !o->perread && o->bbm_name && (bbm = fopen(o->bbm_name, "rb")) == NULL) { fprintf(stderr, "Couldn't open %s for reading!\n", o->bbm_name); FA_JOIN(); plan_free(p); return -8; }
    if(!o->mbias && !o->perread && o->bw_name) { int rc = load_bigwig(p); if(rc) { if(bbm) fclose(bbm); FA_JOIN(); plan_free(p); return rc; } }
    if(bbm) {                  /* as in the reference, a BBM given together with a bigWig replaces the bigWig's bitmaps */
        if(p->map_on) { uint32_t k; for(k = 0; k < p->map_n; k++) { free(p->map_names[k]); free(p->map_bits[k]); } free(p->map_names); free(p->map_len); free(p->map_bits); p->map_names = NULL; p->map_len = NULL; p->map_bits = NULL; p->map_n = 0; }
        { int rc = load_bbm(p, bbm); fclose(bbm); if(rc) { FA_JOIN(); plan_free(p); return rc; } }
    }
    if(fth_ok) FA_JOIN(); else fl.rc = mdk_fasta_load(o->fasta_name, &p->fa);
    if(fl.rc != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", o->fasta_name); plan_free(p); return -4; }
    p->fa_of_tid = xmalloc(sizeof(int) * (size_t)(p->bam->n_targets + 1));
    for(i = 0; i < p->bam->n_targets; i++) p->fa_of_tid[i] = mdk_fasta_find(&p->fa, p->bam->target_name[i]);
    if(p->map_on) {
        p->map_of_tid = xmalloc(sizeof(int) * (size_t)(p->bam->n_targets + 1));
        for(i = 0; i < p->bam->n_targets; i++) { uint32_t k; p->map_of_tid[i] = -1; for(k = 0; k < p->map_n; k++) if(!strcmp(p->map_names[k], p->bam->target_name[i])) { p->map_of_tid[i] = (int)k; break; } }
    }

    if(o->mbias || o->perread) goto region;
    /* output files and headers (extract.c:1343-1439) */
    if(!o->opref) {
        char *dot; o->opref = xstrdup(o->bam_name); dot = strrchr(o->opref, '.'); if(dot) *dot = 0;
        fprintf(stderr, "writing to prefix:'%s'\n", o->opref);
    }
    oname = xmalloc(strlen(o->opref) + 40);
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

/* BGZF inflate on the device as well (mdk_io.c): `extract` and `mbias` in device-preparation mode -- `perRead` prints from the records'
 * bytes on the host.  MDK_HOST_INFLATE=1 keeps every piece on the host's inflate threads. */
int mdk_plan_attach_device(mdk_plan *p, md_dev *dev) {
    if(!p || !dev || !p->bam) return -1;
    if(!p->dev_prep || p->o.perread || getenv("MDK_HOST_INFLATE")) return 0;       /* (perRead prints from the records' bytes on the host: mdk_plan_emit_perread_raw) */
    if(p->bai && p->shard_world > 1) return 0;      /* a rank of a sharded run seeks before every chunk of its own: a 64 MB device piece per seek would be thrown away with the next */
    /* eight teams: a 64 MB piece is ~3,400 members = wavefronts, about half of what the device holds at once, and a team spends a third of its
     * cycle copying the piece into its staging block (512 Mb: 1.23 s inside with 3 teams, 1.04 with 5, 0.96 with 8; 128 Mb: no difference) */
    return mdk_bam_attach_device(p->bam, dev, getenv("MDK_GPU_INFLATE_TEAMS") ? atoi(getenv("MDK_GPU_INFLATE_TEAMS")) : 8);
}
void mdk_plan_detach_device(mdk_plan *p) {
    if(!p || !p->bam || !p->bam->dev) return;
    if(p->started) pipeline_stop(p);            /* the reader holds slabs of device pieces */
    mdk_bam_detach_device(p->bam);
}

int mdk_plan_open(int argc, char *argv[], mdk_plan **out) { return plan_open_ex(argc, argv, out, NULL, NULL); }
/* after_options: called once the command line has been parsed and checked, before the inputs are opened (extract_main starts opening the
 * device there: what the device needs to know are options, and the inputs take another 0.1 s) */
MDK_LOCAL int plan_open_ex(int argc, char *argv[], mdk_plan **out, void (*after_options)(mdk_plan *, void *), void *ctx) {
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
        case 'o': free(o->opref); o->opref = xstrdup(optarg); break;
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
        case 'N': o->output_bb = 1; free(o->out_bbm_name); o->out_bbm_name = xmalloc(strlen(optarg) + 5); sprintf(o->out_bbm_name, "%s.bbm", optarg); break;
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
        char *dot; o->out_bbm_name = xmalloc(strlen(o->bw_name) + 5); strcpy(o->out_bbm_name, o->bw_name);
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

    if(after_options) after_options(p, ctx);
    { int rc = plan_attach_inputs(p, argv, optind); if(rc) return rc; }
    *out = p;
    return 0;
}

MDK_LOCAL void bb_free(batchbuf *b) { md_host_free(b->seg); md_host_free(b->blob); free(b->ri); free(b->cig); free(b->qn); free(b->pr); memset(b, 0, sizeof(*b)); }
MDK_LOCAL void plan_free(mdk_plan *p) {
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
int mdk_plan_set_prep(mdk_plan *p, int mode) {
    if(!p || p->started || mode < 0 || mode > 1) return -1;
    p->dev_prep = mode;
    return 0;
}
int mdk_plan_set_hold(mdk_plan *p, int n) {
    if(!p || p->started || n < 2 || n > MDK_HOLD_MAX) return -1;
    p->n_hold = n;
    return 0;
}
void mdk_plan_prep_cfg(const mdk_plan *p, md_prep_cfg *cfg) {
    const opts_t *o = &p->o;
    memset(cfg, 0, sizeof(*cfg));
    cfg->min_mapq = o->min_mapq; cfg->ignore_flags = o->ignore_flags; cfg->require_flags = o->require_flags; cfg->keep_dupes = o->keep_dupes;
    cfg->ignore_nh = o->ignore_nh; cfg->keep_singleton = o->keep_singleton; cfg->keep_discordant = o->keep_discordant;
    cfg->min_phred = o->min_phred; cfg->min_conv_eff = (float)o->min_conv_eff; cfg->map_on = p->map_on; cfg->min_mappable = o->min_mappable;
    cfg->no_pairing = o->mbias; cfg->perread = o->perread;
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
    if(p->dev_prep && p->map_on) {       /* the admission windows are tested on the device: the contig's track goes with its bases */
        int c = p->map_of_tid[tid];
        if(c >= 0) i = md_dev_set_mappability(dev, tid, (const uint32_t *)p->map_bits[c], (int64_t)p->map_len[c]);
        else i = md_dev_set_mappability(dev, tid, NULL, 0);
        if(i) return i;
    }
    if(p->n_ref == p->cap_ref) { p->cap_ref = p->cap_ref ? p->cap_ref * 2 : 32; p->ref_dev = xrealloc(p->ref_dev, sizeof(md_dev *) * p->cap_ref); p->ref_tid = xrealloc(p->ref_tid, sizeof(int32_t) * p->cap_ref); }
    p->ref_dev[p->n_ref] = dev; p->ref_tid[p->n_ref] = tid; p->n_ref++;
    return 0;
}


/* mdk_mergecontext.c -- `MethylDackel mergeContext`: fold the per-cytosine lines of an `extract` bedGraph into per-CpG /
 * per-CHG lines (mergeContext.c of the reference; main.c:19,53-54 dispatches to mergeContext_main).  A text-to-text
 * host tool: there is nothing data-parallel in it, so no device is involved.  Same messages, return codes and output. */
#include <ctype.h>
#include <getopt.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mdk_extract.h"
#include "mdk_io.h"

#define MC_VERSION "0.6.1"

/* a C waiting for the G of its CpG / CHG (or the other way round) */
typedef struct { char *chrom; int32_t start, end; uint32_t meth, unmeth; } pending;

static void put_line(FILE *out, const char *chrom, int32_t start, int32_t end, uint32_t meth, uint32_t unmeth) {      /* mergeContext.c:24-28 */
    fprintf(out, "%s\t%" PRId32 "\t%" PRId32 "\t%i\t%" PRIu32 "\t%" PRIu32 "\n", chrom, start, end, (int)(100.0 * ((double)meth) / (meth + unmeth)), meth, unmeth);
}
static void flush_pending(FILE *out, pending *q) {
    if(!q->chrom) return;
    put_line(out, q->chrom, q->start, q->end, q->meth, q->unmeth);
    free(q->chrom); q->chrom = NULL;
}
/* the site [start, end) this cytosine belongs to meets its partner, or waits for it (mergeContext.c:30-56) */
static void meet_or_wait(FILE *out, pending *q, char *chrom, int32_t start, int32_t end, uint32_t meth, uint32_t unmeth) {
    if(q->chrom && !strcmp(q->chrom, chrom) && q->start == start && q->end == end) {
        put_line(out, chrom, start, end, meth + q->meth, unmeth + q->unmeth);
        free(q->chrom); q->chrom = NULL; free(chrom);
        return;
    }
    flush_pending(out, q);
    q->chrom = chrom; q->start = start; q->end = end; q->meth = meth; q->unmeth = unmeth;
}

/* 0 CpG, 1 CHG, 2 neither; lo and hi receive the interval of the site.  The partner base must lie within two bases AND inside the
 * contig (mergeContext.c:58-97 looks at the five-base window around pos, clipped to the contig). */
static int site_of(const char *seq, int64_t len, int64_t pos, int32_t *lo, int32_t *hi) {
    int c = toupper((unsigned char)seq[pos]);
    if(c == 'C') {
        if(pos + 1 < len && toupper((unsigned char)seq[pos + 1]) == 'G') { *lo = (int32_t)pos; *hi = (int32_t)pos + 2; return 0; }
        if(pos + 2 < len && toupper((unsigned char)seq[pos + 2]) == 'G') { *lo = (int32_t)pos; *hi = (int32_t)pos + 3; return 1; }
        return 2;
    }
    if(c != 'G') { fprintf(stderr, "[mergeContext] position %" PRId64 " is neither C nor G in the reference\n", pos); abort(); }     /* the reference asserts */
    if(pos >= 1 && toupper((unsigned char)seq[pos - 1]) == 'C') { *lo = (int32_t)pos - 1; *hi = (int32_t)pos + 1; return 0; }
    if(pos >= 2 && toupper((unsigned char)seq[pos - 2]) == 'C') { *lo = (int32_t)pos - 2; *hi = (int32_t)pos + 1; return 1; }
    return 2;
}

static void usage(void) {
    fputs("\nUsage: MethylDackel mergeContext [OPTIONS] <ref.fa> <input>\n\n"
"Merges single-cytosine methylation metrics (a coordinate-sorted bedGraph written by `MethylDackel extract`) into\n"
"per-CpG / per-CHG metrics.\n\nOptions:\n  -o STR    Output file name [stdout]\n  --version Print version and quit\n", stderr);
}
static void malformed(const char *what) { fprintf(stderr, "[mergeContext] malformed input line (%s)\n", what); abort(); }       /* the reference asserts */

int mergeContext_main(int argc, char *argv[]) {
    static const struct option longopts[] = {{"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {0, 0, 0, 0}};
    mdk_fasta fa; FILE *in, *out = stdout; int c; char *line = NULL; size_t cap = 0; ssize_t n; pending cpg = {0}, chg = {0};
    optind = 1;
    while((c = getopt_long(argc, argv, "hvo:", longopts, NULL)) >= 0) {
        switch(c) {
        case 'h': usage(); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", MC_VERSION, "none; methyldackel_amd MI355X build"); return 0;
        case 'o': if((out = fopen(optarg, "w")) == NULL) { fprintf(stderr, "Couldn't open %s for writing\n", optarg); return 2; } break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); usage(); return 1;
        }
    }
    if(argc == 1) { usage(); return 0; }
    if(argc - optind != 2) { fprintf(stderr, "You must supply a reference genome in fasta format and an input bedGraph files\n"); usage(); return -1; }
    memset(&fa, 0, sizeof(fa));
    if(mdk_fasta_load(argv[optind], &fa) != 0) { fprintf(stderr, "Couldn't open the index for %s!\n", argv[optind]); usage(); return -2; }
    if((in = fopen(argv[optind + 1], "r")) == NULL) { fprintf(stderr, "Couldn't open %s for reading!\n", argv[optind + 1]); mdk_fasta_free(&fa); return -3; }
    fputs("track type=\"bedGraph\" description=\"merged Methylation metrics\"\n", out);
    while((n = getline(&line, &cap, in)) >= 0) {
        char *f[6], *end, *chrom; int k, fi, type; int32_t start, stop, lo = 0, hi = 0; uint32_t meth, unmeth;
        if(n && line[n - 1] == '\n') line[--n] = 0;
        if(n > 1 && line[n - 1] == '\r') line[--n] = 0;
        if(n == 0) malformed("empty line");
        if(!strncmp(line, "track", 5)) continue;
        for(k = 0, f[0] = strtok(line, "\t"); k < 5 && f[k]; k++) f[k + 1] = strtok(NULL, k == 4 ? "\n" : "\t");
        if(k < 5 || !f[5]) malformed("fewer than six columns");
        start = (int32_t)strtoll(f[1], &end, 10); if(end == f[1]) malformed("start");
        stop = (int32_t)strtoll(f[2], &end, 10); if(end == f[2]) malformed("end");
        meth = (uint32_t)strtoul(f[4], &end, 10); if(end == f[4]) malformed("methylated count");
        unmeth = (uint32_t)strtoul(f[5], &end, 10); if(end == f[5]) malformed("unmethylated count");
        fi = mdk_fasta_find(&fa, f[0]);
        if(fi < 0) { fprintf(stderr, "[mergeContext] Error, %s is an unknown chromosome name!\n", f[0]); break; }
        /* a start outside the contig (bedGraph made against another FASTA, corrupt line): the reference fetches an empty window
         * and reads past it; here the line is refused the way an unknown contig is */
        if(start < 0 || (int64_t)start >= fa.len[fi]) { fprintf(stderr, "[mergeContext] Error, position %d is outside of %s (%" PRId64 " bases)!\n", start, f[0], fa.len[fi]); break; }
        chrom = xstrdup(f[0]);
        type = site_of(fa.seq[fi], fa.len[fi], start, &lo, &hi);
        if(type == 0) meet_or_wait(out, &cpg, chrom, lo, hi, meth, unmeth);
        else if(type == 1) meet_or_wait(out, &chg, chrom, lo, hi, meth, unmeth);
        else { put_line(out, chrom, start, stop, meth, unmeth); free(chrom); }
    }
    flush_pending(out, &cpg);
    flush_pending(out, &chg);
    free(line);
    if(out != stdout) fclose(out);
    fclose(in);
    mdk_fasta_free(&fa);
    return 0;
}

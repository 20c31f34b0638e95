/* mdk_mbias.c -- what `mbias` does with the histogram once the device has filled it: the tab-separated table, the four
 * per-strand SVG plots and the suggested inclusion bounds.  Mirrors svg.c of the reference (makeTXT 439-454, makeSVGs
 * 300-437, getThresholds 239-294, CI 8-27, axis helpers 29-172); the floating-point expressions keep the reference's
 * operation order so that the "%f"-formatted coordinates come out identical.
 *
 * The histogram arrives as md_mbias rows count[q*16 + (strand-1)*4 + (read 2 ? 2 : 0) + (unmethylated ? 1 : 0)]. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mdk_extract.h"
#include "mdk_io.h"

/* calls of one (strand, read number): methylated / unmethylated per position in the read */
typedef struct { const uint32_t *row; int col; int len; } series;
static inline uint32_t s_meth(const series *s, int q) { return q < s->len ? s->row[(size_t)q * 16 + s->col] : 0; }
static inline uint32_t s_unmeth(const series *s, int q) { return q < s->len ? s->row[(size_t)q * 16 + s->col + 1] : 0; }
static inline int s_any(const series *s, int q) { return s_meth(s, q) || s_unmeth(s, q); }

/* Agresti-Coull bound of the methylated fraction, z = qnorm(0.9995) (svg.c:10-27) */
static double ac_bound(uint32_t unmeth, uint32_t meth, int upper) {
    const double zz = 10.8275661707, z = 3.2905267315;
    double x = (double)meth, n = (double)(meth + unmeth), nd = n + zz, pd = (1.0 / nd) * (x + 0.5 * zz), v;
    if(upper) { v = pd + z * sqrt((pd / nd) * (1 - pd)); if(v > 1.) v = 1.0; }
    else { v = pd - z * sqrt((pd / nd) * (1 - pd)); if(v < 0.) v = 0.0; }
    return v;
}
static double frac(const series *s, int q) { return ((double)s_meth(s, q)) / ((double)(s_meth(s, q) + s_unmeth(s, q))); }

/* plot geometry: an 80-pixel margin around a 500x500 area */
enum { MARGIN = 80, AREA = 500 };
static double px(int x, int xmax) { return MARGIN + ((double)AREA) * x / ((double)xmax); }
static double py(double y, double ymin, double ymax) { return MARGIN + AREA - ((double)AREA) * (y - ymin) / (ymax - ymin); }

/* y range: the extreme confidence bounds of both reads, padded by 0.03, snapped outwards to a multiple of 0.05, and
 * opened up to 0 / 1 when close (svg.c:29-79) */
static void y_range(const series s[2], int len, double *ymin, double *ymax) {
    double lo = 1.0, hi = 0.0, v; int q, r, c;
    for(q = 0; q < len; q++) for(r = 0; r < 2; r++) if(s_meth(&s[r], q) + s_unmeth(&s[r], q)) {
        v = ac_bound(s_unmeth(&s[r], q), s_meth(&s[r], q), 1); hi = v > hi ? v : hi;
        v = ac_bound(s_unmeth(&s[r], q), s_meth(&s[r], q), 0); lo = v < lo ? v : lo;
    }
    hi += 0.03;
    c = (int)ceil(100 * hi);
    hi = (5 * (c / 5) - c) ? (1 + c / 5) * 0.05 : (c / 5) * 0.05;
    if(hi > 0.8) hi = 1.0;
    lo -= 0.03;
    lo = 0.01 * (5 * (((int)(100 * lo)) / 5));
    if(lo < 0.2) lo = 0.0;
    *ymin = lo; *ymax = hi;
}

/* inclusion bounds suggested for one read of one strand (svg.c:230-294): the mean of the middle 60 % and the envelope
 * of its confidence bounds; walking outwards from the middle, the first position that is significantly and by more
 * than 0.05 off that mean ends the included stretch */
static int off_plateau(const series *s, int q, double mean, double min_upper, double max_lower) {
    double f = frac(s, q);
    if(ac_bound(s_unmeth(s, q), s_meth(s, q), 1) < mean && f < min_upper && fabs(f - mean) > 0.05) return 1;
    if(ac_bound(s_unmeth(s, q), s_meth(s, q), 0) > mean && f > max_lower && fabs(f - mean) > 0.05) return 1;
    return 0;
}
static void suggest(const series *s, int len, int *left, int *right) {
    int q, n = 0, mid = len / 2; double mean = 0.0, min_upper = 1.0, max_lower = 0.0, v;
    *left = *right = 0;
    for(q = (int)(0.2 * len); q <= (int)(0.8 * len); q++) if(s_any(s, q)) {
        n++; mean += frac(s, q);
        v = ac_bound(s_unmeth(s, q), s_meth(s, q), 1); if(min_upper > v) min_upper = v;
        v = ac_bound(s_unmeth(s, q), s_meth(s, q), 0); if(max_lower < v) max_lower = v;
    }
    if(!n) return;
    mean /= n;
    for(q = mid; q >= 0; q--) if(s_any(s, q) && off_plateau(s, q, mean, min_upper, max_lower)) break;
    if(q >= 0) *left = q + 2;
    for(q = mid + 1; q < len; q++) if(s_any(s, q) && off_plateau(s, q, mean, min_upper, max_lower)) break;
    if(q < len) *right = q;
}

/* shaded confidence band (lower bounds left to right, upper bounds back) and the line of the fractions (svg.c:174-228) */
static void draw_series(FILE *f, const series *s, int len, int first, int xmax, double ymin, double ymax, const char *colour) {
    int q;
    fprintf(f, "<path d=\"M %f %f\n", px(first + 1, xmax), py(ac_bound(s_unmeth(s, first), s_meth(s, first), 0), ymin, ymax));
    for(q = first + 1; q <= len; q++) if(s_any(s, q)) fprintf(f, "  L %f %f\n", px(q + 1, xmax), py(ac_bound(s_unmeth(s, q), s_meth(s, q), 0), ymin, ymax));
    for(q = len - 1; q >= 0; q--) if(s_any(s, q)) fprintf(f, "  L %f %f\n", px(q + 1, xmax), py(ac_bound(s_unmeth(s, q), s_meth(s, q), 1), ymin, ymax));
    fprintf(f, "Z\" fill=\"%s\" fill-opacity=\"0.2\"/>\n", colour);
}
static void draw_line(FILE *f, const series *s, int len, int first, int xmax, double ymin, double ymax, const char *colour) {
    int q;
    fprintf(f, "<path d=\"M %f %f\n", px(first + 1, xmax), py(frac(s, first), ymin, ymax));
    for(q = first + 1; q <= len; q++) if(s_any(s, q)) fprintf(f, "  L %f %f\n", px(q + 1, xmax), py(frac(s, q), ymin, ymax));
    fprintf(f, "\" stroke=\"%s\" stroke-width=\"2\" fill-opacity=\"0\"/>\n", colour);
}
static void bound_marker(FILE *f, int x, int xmax, const char *colour) {
    fprintf(f, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 1\" stroke=\"%s\" stroke-width=\"1\" />\n", px(x, xmax), AREA + MARGIN, px(x, xmax), MARGIN, colour);
}

static const char *ABBREV[4] = {"OT", "OB", "CTOT", "CTOB"};
static const char *TITLE[4] = {"Original Top", "Original Bottom", "Complementary to the Original Top", "Complementary to the Original Bottom"};
static const char *COLOUR[2] = {"rgb(248,118,109)", "rgb(0,191,196)"};

/* number of positions of a strand: one past its last position with a call (strandMeth.l, MBias.c:210) */
static int strand_len(const md_mbias *h, int strand) {
    int q, c;
    for(q = h->len; q > 0; q--) for(c = 0; c < 4; c++) if(h->count[(size_t)(q - 1) * 16 + strand * 4 + c]) return q;
    return 0;
}

static int svg_of_strand(const char *opref, int strand, const series s[2], int len, int which, int sugg[4]) {
    char *name = xmalloc(strlen(opref) + 16); FILE *f; double ymin, ymax, span; int first[2] = {len, len}, has[2] = {0, 0}, xmax, q, r, j, n, step, labelled = 0;
    sprintf(name, "%s_%s.svg", opref, ABBREV[strand]);
    f = fopen(name, "w");
    free(name);
    if(!f) return -1;
    y_range(s, len, &ymin, &ymax);
    for(r = 0; r < 2; r++) for(q = 0; q < len; q++) if(s_any(&s[r], q)) { first[r] = q; has[r] = 1; break; }
    xmax = len;                                  /* the strand's last position has a call by construction ... */
    if(xmax % 5) xmax += 5 - (xmax % 5);         /* ... rounded up to a multiple of 5 (svg.c:93-107) */
    fprintf(f, "<svg height=\"%i\" width=\"%i\"\n", AREA + 2 * MARGIN, AREA + 2 * MARGIN);
    fputs("    xmlns=\"http://www.w3.org/2000/svg\"\n    xmlns:xlink=\"http://www.w3.org/1999/xlink\"\n    xmlns:ev=\"http://www.w3.org/2001/xml-events\">\n", f);
    fprintf(f, "<title>%s Strand</title>\n", TITLE[strand]);
    fprintf(f, "<rect x=\"0\" y=\"0\" width=\"%i\" height=\"%i\" fill=\"white\" />\n", AREA + 2 * MARGIN, AREA + 2 * MARGIN);
    fprintf(f, "<text x=\"%i\" y=\"%i\" text-anchor=\"middle\">%s Strand</text>\n", MARGIN + (AREA >> 1), 20, TITLE[strand]);
    fprintf(f, "<line x1=\"%i\" y1=\"%i\" x2=\"%i\" y2=\"%i\" stroke=\"black\" />\n", MARGIN, MARGIN, MARGIN, MARGIN + AREA);
    fprintf(f, "<line x1=\"%i\" y1=\"%i\" x2=\"%i\" y2=\"%i\" stroke=\"black\" />\n", MARGIN, MARGIN + AREA, MARGIN + AREA, MARGIN + AREA);
    /* axis titles */
    fprintf(f, "<text x=\"15\" y=\"%i\" transform=\"rotate(270 15, %i)\" text-anchor=\"middle\" dominant-baseline=\"text-before-edge\">", MARGIN + (AREA >> 1), MARGIN + (AREA >> 1));
    for(j = 0; j < 3; j++) if(which & (1 << j)) { fprintf(f, "%s%s", labelled ? "/" : "", j == 0 ? "CpG" : j == 1 ? "CHG" : "CHH"); labelled = 1; }
    if(labelled) fputc(' ', f);
    fputs("Methylation %</text>\n", f);
    fprintf(f, "<text x=\"%i\" y=\"%i\" text-anchor=\"middle\">Position along mapped read (5'->3' of + strand)</text>\n", MARGIN + (AREA >> 1), MARGIN + AREA + 40);
    /* x ticks every 5 positions, every 10 when that would be more than 7 (svg.c:109-149) */
    fprintf(f, "<line x1=\"%i\" y1=\"%i\" x2=\"%i\" y2=\"%i\" stroke=\"black\" />\n", MARGIN, MARGIN + AREA, MARGIN, MARGIN + AREA + 5);
    fprintf(f, "<text x=\"%i\" y=\"%i\" text-anchor=\"middle\">%i</text>\n", MARGIN, MARGIN + AREA + 20, 0);
    step = 5; n = xmax / 5; if(n > 7) { step = 10; n = xmax / 10; }
    for(j = 1; j <= n; j++) {
        double x = px(j * step, xmax);
        fprintf(f, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke-dasharray=\"5 5\" stroke=\"grey\" />\n", x, MARGIN, x, MARGIN + AREA);
        fprintf(f, "<line x1=\"%f\" y1=\"%i\" x2=\"%f\" y2=\"%i\" stroke=\"black\" />\n", x, MARGIN + AREA, x, MARGIN + AREA + 5);
        fprintf(f, "<text x=\"%f\" y=\"%i\" text-anchor=\"middle\">%i</text>\n", x, MARGIN + AREA + 20, j * step);
    }
    /* y ticks every 0.05 (svg.c:151-164) */
    span = ymax - ymin;
    n = (int)(1 + ceil(span / 0.05)); if(span < 0.05) n = 2;
    for(j = 0; j < n; j++) {
        double v = 0.05 * j + ymin, y = py(v, ymin, ymax);
        fprintf(f, "<line x1=\"%i\" y1=\"%f\" x2=\"%i\" y2=\"%f\" stroke=\"black\" />\n", MARGIN, y, MARGIN - 5, y);
        fprintf(f, "<text x=\"%i\" y=\"%f\" text-anchor=\"middle\" dominant-baseline=\"middle\">%4.2f</text>\n", MARGIN - 25, y, v);
    }
    for(r = 0; r < 2; r++) if(has[r]) draw_series(f, &s[r], len, first[r], xmax, ymin, ymax, COLOUR[r]);
    for(r = 0; r < 2; r++) if(has[r]) draw_line(f, &s[r], len, first[r], xmax, ymin, ymax, COLOUR[r]);
    suggest(&s[0], len, &sugg[0], &sugg[1]);
    suggest(&s[1], len, &sugg[2], &sugg[3]);
    if(sugg[0] + sugg[1] + sugg[2] + sugg[3]) {
        fprintf(f, "<text x=\"%i\" y=\"%i\" text-anchor=\"end\">--%s %i,%i,%i,%i</text>\n", 2 * MARGIN + AREA - 10, 2 * MARGIN + AREA - 10, ABBREV[strand], sugg[0], sugg[1], sugg[2], sugg[3]);
        for(j = 0; j < 4; j++) if(sugg[j]) bound_marker(f, sugg[j], xmax, COLOUR[j >> 1]);
    }
    for(r = 0; r < 2; r++) if(has[r]) {         /* legend */
        fprintf(f, "<rect x=\"%i\" y=\"%i\" width=\"20\" height=\"20\" fill=\"%s\" />\n", AREA + MARGIN + 10, (AREA >> 1) + MARGIN - 20 + 20 * r, COLOUR[r]);
        fprintf(f, "<text x=\"%i\" y=\"%i\" text-anchor=\"start\" dominant-baseline=\"middle\">#%i</text>\n", AREA + MARGIN + 35, (AREA >> 1) + MARGIN - 10 + 20 * r, r + 1);
    }
    fputs("</svg>\n", f);
    fclose(f);
    return 0;
}

int mdk_mbias_report(const md_mbias *h, const char *opref, int svg, int txt, int which) {
    int strand, q, r, printing = 0;
    if(!h || (h->len > 0 && !h->count) || (svg && !opref)) return -1;
    if(svg) {
        for(strand = 0; strand < 4; strand++) {
            int len = strand_len(h, strand), sugg[4]; series s[2];
            if(!len) continue;
            for(r = 0; r < 2; r++) { s[r].row = h->count; s[r].col = strand * 4 + 2 * r; s[r].len = h->len; }
            if(svg_of_strand(opref, strand, s, len, which, sugg)) { fprintf(stderr, "[mdk] cannot write %s_%s.svg\n", opref, ABBREV[strand]); return -3; }
            if(!printing) fprintf(stderr, "Suggested inclusion options:");
            fprintf(stderr, " --%s %i,%i,%i,%i", ABBREV[strand], sugg[0], sugg[1], sugg[2], sugg[3]);
            printing = 1;
        }
        if(printing) fputc('\n', stderr);
    }
    if(txt) {
        printf("Strand\tRead\tPosition\tnMethylated\tnUnmethylated\n");
        for(strand = 0; strand < 4; strand++) {
            int len = strand_len(h, strand);
            for(q = 0; q < len; q++) for(r = 0; r < 2; r++) {
                uint32_t m = h->count[(size_t)q * 16 + strand * 4 + 2 * r], u = h->count[(size_t)q * 16 + strand * 4 + 2 * r + 1];
                if(m || u) printf("%s\t%i\t%i\t%u\t%u\n", ABBREV[strand], r + 1, q + 1, m, u);
            }
        }
    }
    return 0;
}

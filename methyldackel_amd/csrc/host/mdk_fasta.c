/* mdk_fasta.c -- the reference genome in memory, as the reference's faidx_fetch_seq hands it out contig by contig (extract.c:368, common.c
 * uses of fai): per contig the printable characters of its lines (c > ' ' && c <= '~', case kept), names cut at the first blank.
 *
 * A human genome is 3.1 GB of text; a byte loop on one thread moves ~2 GB/s, which is 1.5 s before the first chunk can be piled up (and was
 * a quarter of the whole run on the 512 Mb bench input).  So a large file is loaded by several threads:
 *   1. the file is cut into as many stretches as threads; each thread reads its own (pread);
 *   2. the cuts move forward to the next line start; each thread lists the header lines of its stretch and counts the sequence bytes in
 *      front of and between them;
 *   3. one thread lays the contigs out back to back (names behind the sequences);
 *   4. each thread copies the bytes of its lines to where they belong.
 * The result equals the one-thread loader's byte for byte (tests/test_ranks_cpu.py runs the command both ways; MDK_FASTA_THREADS=n forces n). */
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "mdk_io.h"

static int fasta_load_serial(const char *fn, mdk_fasta *fa) {
    FILE *f = fopen(fn, "rb"); size_t sz, i, w; char *d; int cur = -1, cap = 0;
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
            if(fa->n == cap) { cap = cap ? cap * 2 : 64; fa->name = xrealloc(fa->name, sizeof(char *) * cap); fa->seq = xrealloc(fa->seq, sizeof(char *) * cap); fa->len = xrealloc(fa->len, sizeof(int64_t) * cap); }
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

/* ---- several threads ---- */
typedef struct { size_t line, end; uint64_t seq_after; } fa_hdr;      /* a header line [line, end) and the sequence bytes between it and the next header of the stretch */
typedef struct { fa_hdr *h; int n, cap; uint64_t lead, out0; int failed; } fa_part;     /* lead: sequence bytes in front of the stretch's first header; out0: where the stretch's first kept byte goes */
typedef struct {
    int fd, nth; size_t sz; char *raw, *pool; size_t *cut;             /* stretch t = [cut[t], cut[t+1]) once the cuts stand at line starts */
    fa_part *part; pthread_barrier_t bar; mdk_fasta *fa; int rc;
} fa_job;
typedef struct { fa_job *j; int t; } fa_arg;

/* how many of n bytes are kept (c > ' ' && c <= '~'), eight at a time: bit 7 of a byte of `bad` is set when the byte is >= 0x80, == 0x7f or < 0x21 */
static inline size_t fa_kept(const char *p, size_t n) {
    size_t k = 0, bad = 0;
    for(; k + 8 <= n; k += 8) {
        uint64_t x, y; memcpy(&x, p + k, 8); y = x & 0x7f7f7f7f7f7f7f7full;
        bad += (size_t)__builtin_popcountll((x | (y + 0x0101010101010101ull) | ~(y + 0x5f5f5f5f5f5f5f5full)) & 0x8080808080808080ull);
    }
    for(; k < n; k++) { const unsigned char c = (unsigned char)p[k]; bad += (size_t)((c <= ' ') | (c > '~')); }
    return n - bad;
}
static inline char *fa_copy_line(char *w, const char *p, size_t n) {
    if(fa_kept(p, n) == n) { memcpy(w, p, n); return w + n; }
    { size_t k; for(k = 0; k < n; k++) { const unsigned char c = (unsigned char)p[k]; if(c > ' ' && c <= '~') *w++ = (char)c; } }
    return w;
}
static void fa_layout(fa_job *j) {      /* one thread, between the counting and the copying */
    mdk_fasta *fa = j->fa; int t, k, n = 0, cur = -1; uint64_t w = 0, names = 0; char *np;
    for(t = 0; t < j->nth; t++) { if(j->part[t].failed) { j->rc = -1; return; } n += j->part[t].n; for(k = 0; k < j->part[t].n; k++) names += (uint64_t)(j->part[t].h[k].end - j->part[t].h[k].line) + 1; }
    fa->name = xrealloc(NULL, sizeof(char *) * (size_t)(n ? n : 1)); fa->seq = xrealloc(NULL, sizeof(char *) * (size_t)(n ? n : 1)); fa->len = xrealloc(NULL, sizeof(int64_t) * (size_t)(n ? n : 1));
    /* first the sizes: what stands in front of the file's first header belongs to nobody and is not kept */
    for(t = 0; t < j->nth; t++) {
        fa_part *P = &j->part[t];
        P->out0 = w; if(cur >= 0) { fa->len[cur] += (int64_t)P->lead; w += P->lead; }
        for(k = 0; k < P->n; k++) { cur = fa->n++; fa->len[cur] = (int64_t)P->h[k].seq_after; fa->seq[cur] = (char *)(uintptr_t)w; w += P->h[k].seq_after; }
    }
    j->pool = malloc((size_t)w + (size_t)names + 2);
    if(!j->pool) { j->rc = -1; return; }
    fa->pool = j->pool; np = j->pool + w; cur = 0;
    for(t = 0; t < j->nth; t++) for(k = 0; k < j->part[t].n; k++, cur++) {
        const fa_hdr *h = &j->part[t].h[k]; size_t s = h->line + 1, e = s;
        while(e < h->end && j->raw[e] != ' ' && j->raw[e] != '\t' && j->raw[e] != '\r') e++;
        fa->seq[cur] = j->pool + (uintptr_t)fa->seq[cur];
        memcpy(np, j->raw + s, e - s); fa->name[cur] = np; np += e - s; *np++ = 0;
    }
}
static double fa_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void *fa_main(void *arg) {
    fa_arg *a = arg; fa_job *j = a->j; const int t = a->t; fa_part *P = &j->part[t];
    const int prof = t == 0 && getenv("MDK_HOST_PROFILE") != NULL; double t0 = prof ? fa_now() : 0, t1, t2, t3;
    {   /* 1. this thread's stretch of the file */
        size_t o = j->cut[t]; const size_t e = j->cut[t + 1];
        while(o < e) { const ssize_t r = pread(j->fd, j->raw + o, e - o, (off_t)o); if(r <= 0) { P->failed = 1; break; } o += (size_t)r; }
    }
    pthread_barrier_wait(&j->bar);
    t1 = prof ? fa_now() : 0;
    /* 2. the cuts to line starts (thread 0, few of them), then the lines of the stretch counted */
    if(t == 0) { int q; for(q = 1; q < j->nth; q++) { size_t c = j->cut[q]; if(c < j->cut[q - 1]) c = j->cut[q - 1]; if(c > 0 && c < j->sz && j->raw[c - 1] != '\n') { const char *nl = memchr(j->raw + c, '\n', j->sz - c); c = nl ? (size_t)(nl - j->raw) + 1 : j->sz; } j->cut[q] = c; } }
    pthread_barrier_wait(&j->bar);
    {
        size_t i = j->cut[t]; const size_t end = j->cut[t + 1]; uint64_t run = 0;
        while(i < end) {          /* i stands at a line start.  The next header line: a '>' that follows a newline (or starts the file) */
            size_t h = i, e;
            for(;;) { const char *g = memchr(j->raw + h, '>', end - h); if(!g) { h = end; break; } h = (size_t)(g - j->raw); if(h == 0 || j->raw[h - 1] == '\n') break; h++; }
            run += fa_kept(j->raw + i, h - i);                 /* the lines in between: every byte of theirs that is kept (their newlines are not) */
            if(h >= end) break;
            e = (size_t)((const char *)memchr(j->raw + h, '\n', j->sz + 1 - h) - j->raw);                  /* raw[sz] is a newline */
            if(P->n) P->h[P->n - 1].seq_after = run; else P->lead = run;
            if(P->n == P->cap) { P->cap = P->cap ? P->cap * 2 : 16; P->h = xrealloc(P->h, sizeof(fa_hdr) * (size_t)P->cap); }
            P->h[P->n].line = h; P->h[P->n].end = e; P->h[P->n].seq_after = 0; P->n++; run = 0;
            i = e + 1;
        }
        if(P->n) P->h[P->n - 1].seq_after = run; else P->lead = run;
    }
    pthread_barrier_wait(&j->bar);
    t2 = prof ? fa_now() : 0;
    if(t == 0) fa_layout(j);
    pthread_barrier_wait(&j->bar);
    t3 = prof ? fa_now() : 0;
    if(j->rc == 0 && j->fa->n > 0) {
        /* 4. the stretch's lines to their places.  Bytes in front of the file's first header have none: a stretch that lies wholly in front of it
         * copies nothing, the one that holds it starts copying behind it */
        size_t i = j->cut[t]; const size_t end = j->cut[t + 1]; char *w = j->pool + P->out0; int q, headers_before = 0;
        for(q = 0; q < t; q++) headers_before += j->part[q].n;
        { int seen = headers_before > 0;
          while(i < end) {
            const char *nl = memchr(j->raw + i, '\n', j->sz + 1 - i); const size_t e = (size_t)(nl - j->raw);
            if(j->raw[i] == '>') seen = 1;
            else if(seen) w = fa_copy_line(w, j->raw + i, e - i);
            i = e + 1;
          } }
    }
    if(prof) { pthread_barrier_wait(&j->bar); fprintf(stderr, "[mdk host] reference text by %d threads: read %.3fs, lines counted %.3fs, layout %.3fs, copied %.3fs\n", j->nth, t1 - t0, t2 - t1, t3 - t2, fa_now() - t3); }
    else if(getenv("MDK_HOST_PROFILE")) pthread_barrier_wait(&j->bar);
    return NULL;
}
static int fasta_load_threads(const char *fn, mdk_fasta *fa, int nth, size_t sz) {
    fa_job J; pthread_t *th; fa_arg *args; int t, started = 0;
    memset(&J, 0, sizeof(J));
    J.fd = open(fn, O_RDONLY); if(J.fd < 0) return -1;
    J.nth = nth; J.sz = sz; J.fa = fa;
    J.raw = malloc(sz + 2); J.cut = calloc((size_t)nth + 1, sizeof(size_t)); J.part = calloc((size_t)nth, sizeof(fa_part));
    th = calloc((size_t)nth, sizeof(*th)); args = calloc((size_t)nth, sizeof(*args));
    if(!J.raw || !J.cut || !J.part || !th || !args || pthread_barrier_init(&J.bar, NULL, (unsigned)nth)) { close(J.fd); free(J.raw); free(J.cut); free(J.part); free(th); free(args); return -1; }
    J.raw[sz] = '\n'; J.raw[sz + 1] = 0;
    for(t = 0; t <= nth; t++) J.cut[t] = (size_t)((unsigned __int128)sz * (unsigned)t / (unsigned)nth);
    for(t = 0; t < nth; t++) { args[t].j = &J; args[t].t = t; if(pthread_create(&th[t], NULL, fa_main, &args[t])) break; started++; }
    if(started < nth) {      /* the threads that did start wait at a barrier for ones that never will: no way on from here but out */
        fprintf(stderr, "[mdk] cannot create the threads that load %s\n", fn); _exit(1);
    }
    for(t = 0; t < nth; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&J.bar);
    close(J.fd);
    for(t = 0; t < nth; t++) free(J.part[t].h);
    free(J.raw); free(J.cut); free(J.part); free(th); free(args);
    if(J.rc) { free(fa->pool); free(fa->name); free(fa->seq); free(fa->len); memset(fa, 0, sizeof(*fa)); return -1; }
    return 0;
}

int mdk_fasta_load(const char *fn, mdk_fasta *fa) {
    struct stat st; int nth = 1;
    memset(fa, 0, sizeof(*fa));
    if(stat(fn, &st) == 0 && S_ISREG(st.st_mode)) {
        const char *ev = getenv("MDK_FASTA_THREADS"); const long cores = sysconf(_SC_NPROCESSORS_ONLN);
        if(ev) nth = atoi(ev);
        else if(st.st_size >= (off_t)(48 << 20)) { nth = (int)(st.st_size >> 24); if(nth > 16) nth = 16; if(cores > 0 && nth > cores / 2) nth = (int)(cores / 2); }      /* a stretch of >= 16 MB each */
        if(nth > 64) nth = 64;
        if((off_t)nth > st.st_size) nth = (int)st.st_size;
    }
    if(nth > 1) return fasta_load_threads(fn, fa, nth, (size_t)st.st_size);
    return fasta_load_serial(fn, fa);
}
void mdk_fasta_free(mdk_fasta *fa) { free(fa->pool); free(fa->name); free(fa->seq); free(fa->len); memset(fa, 0, sizeof(*fa)); }
int mdk_fasta_find(const mdk_fasta *fa, const char *name) { int i; for(i = 0; i < fa->n; i++) if(!strcmp(fa->name[i], name)) return i; return -1; }

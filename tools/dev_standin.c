/* dev_standin.c -- TEST INFRASTRUCTURE: the whole device library (include/mdk_hip.h) without a device, as a library to LD_PRELOAD in front of
 * libmdk_hip.so, so that the COMMAND ITSELF -- extract_main's uploader / collector / reference threads (csrc/host/mdk_extract.c) and the
 * one-process-per-GPU driver (csrc/host/mdk_ranks.c: schedule sharding, the ring of chunks in flight on rank 0, the control and data
 * connections, ordered emission, a chunk handed back to the host) -- runs on a CPU-only box: tests/test_ranks_cpu.py.
 * "Device memory" is host memory; BGZF pieces are tools/piece_standin.c (zlib); and what a slot "computes" is looked up, by the slot's
 * contig and interval, in the per-column counters the oracle dumped for the same command line (MDK_ORACLE_DUMP -> MDK_STANDIN_DUMP): the
 * counting itself is what the GPU tests check, everything AROUND it is what runs here.  MDK_STANDIN_HANDBACK=k makes every k-th uploaded chunk
 * come back with MDK_ERR_PREP_HOST once, as a chunk with an over-long read-name chain does on the device; MDK_STANDIN_US_PER_KREC=t makes a
 * chunk take t microseconds per 1000 records on the "device" ("compute time" in the background, for the work balance between ranks).
 *   build: gcc -O2 -shared -fPIC -Iinclude -o tools/_build/libmdk_dev_standin.so tools/dev_standin.c -lz -lpthread */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>
#include "piece_standin.c"

typedef struct { int32_t tid; uint32_t pos, nm, nu, meta, noff, nvar; } row_t;
static row_t *g_row; static size_t g_nrow; static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static __thread char t_err[256];
static void load_dump(void) {
    const char *fn = getenv("MDK_STANDIN_DUMP"); FILE *f = fn ? fopen(fn, "r") : NULL; size_t cap = 0; int tid, pos, type, isg; unsigned a, b, c, d;
    if(!f) return;
    while(fscanf(f, "%d %d %d %d %u %u %u %u", &tid, &pos, &type, &isg, &a, &b, &c, &d) == 8) {
        if(g_nrow == cap) { cap = cap ? cap * 2 : 1 << 16; g_row = realloc(g_row, sizeof(row_t) * cap); if(!g_row) abort(); }
        g_row[g_nrow].tid = tid; g_row[g_nrow].pos = (uint32_t)pos; g_row[g_nrow].nm = a; g_row[g_nrow].nu = b; g_row[g_nrow].meta = (uint32_t)((isg ? 1 : 0) | (type << 1)); g_row[g_nrow].noff = c; g_row[g_nrow].nvar = d; g_nrow++;
    }
    fclose(f);
}
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
typedef struct { double ready_at; int used, launched, handed_back; int32_t tid; int64_t beg, end; uint8_t *raw; uint64_t raw_bytes, raw_cap; uint32_t *off; uint32_t n_rec, off_cap; md_site *site; md_site_var *var; int64_t cap; } sslot;
struct md_dev { md_dev_cfg cfg; int n_slots; sslot *slot; long n_up; int handback; pthread_mutex_t mu; double busy_until; int us_per_krec; };
const char *md_dev_last_error(void) { return t_err; }
int md_dev_count(void) { return 1; }
int md_dev_warm(int device) { (void)device; return 0; }
void md_dev_quiesce(void) { }
void md_dev_reserve_hint(uint64_t device_bytes) { (void)device_bytes; }
int md_dev_open(int device, const md_dev_cfg *cfg, md_dev **out) {
    md_dev *h = calloc(1, sizeof(*h)); (void)device;
    pthread_once(&g_once, load_dump);
    if(!h || !cfg) return MDK_ERR_ARG;
    if(!g_row && getenv("MDK_STANDIN_DUMP") == NULL) { snprintf(t_err, sizeof t_err, "dev_standin: MDK_STANDIN_DUMP is not set"); free(h); return MDK_ERR_NODEVICE; }
    h->cfg = *cfg; h->n_slots = cfg->n_slots > 0 ? cfg->n_slots : 2; h->slot = calloc((size_t)h->n_slots, sizeof(sslot)); pthread_mutex_init(&h->mu, NULL);
    h->handback = getenv("MDK_STANDIN_HANDBACK") ? atoi(getenv("MDK_STANDIN_HANDBACK")) : 0;
    h->us_per_krec = getenv("MDK_STANDIN_US_PER_KREC") ? atoi(getenv("MDK_STANDIN_US_PER_KREC")) : 0;
    *out = h; return h->slot ? 0 : MDK_ERR_NOMEM;
}
void md_dev_close(md_dev *h) { int i; if(!h) return; for(i = 0; i < h->n_slots; i++) { free(h->slot[i].raw); free(h->slot[i].off); free(h->slot[i].site); free(h->slot[i].var); } free(h->slot); free(h); }
int md_dev_tile(const md_dev *h) { (void)h; return 2048; }
int md_dev_reserve_contigs(md_dev *h, int32_t n) { (void)h; (void)n; return 0; }
int md_dev_set_reference(md_dev *h, int32_t tid, const char *seq, int64_t len) { (void)h; (void)tid; (void)seq; (void)len; return 0; }
int md_dev_set_regions(md_dev *h, int32_t tid, const md_region *runs, int64_t n) { (void)h; (void)tid; (void)runs; (void)n; return 0; }
int md_dev_set_mappability(md_dev *h, int32_t tid, const uint32_t *bits, int64_t n) { (void)h; (void)tid; (void)bits; (void)n; return 0; }
int md_dev_set_prep(md_dev *h, const md_prep_cfg *cfg) { (void)h; (void)cfg; return 0; }
int md_dev_pci_bus_id(const md_dev *h, char *buf, int cap) { (void)h; snprintf(buf, (size_t)cap, "standin:00.0"); return 0; }      /* every rank "on the same device": the site buffers travel over the ranks' TCP connections */
int md_dev_profile_text(char *buf, int cap) { if(buf && cap > 0) snprintf(buf, (size_t)cap, "device stand-in (tools/dev_standin.c)"); return 0; }
void *md_host_alloc(uint64_t bytes) { return malloc((size_t)bytes + 64); }
void md_host_free(void *p) { free(p); }
void md_host_set_pinned(int on) { (void)on; }
void md_host_profile(double *s, uint64_t *c, uint64_t *b) { if(s) *s = 0; if(c) *c = 0; if(b) *b = 0; }
int md_host_register_all(md_dev *h, int threads) { (void)h; (void)threads; return 0; }
static sslot *slot_of(md_dev *h, int slot) { if(!h || slot < 0 || slot >= h->n_slots) { snprintf(t_err, sizeof t_err, "bad slot"); return NULL; } return &h->slot[slot]; }
/* the records as the device would hold them: the ranges back to back, and every record's offset there (csrc/mdk_prep.hip copy_ranges) */
int md_dev_upload_raw(md_dev *h, int slot, const md_raw_batch *b) {
    sslot *s = slot_of(h, slot); uint64_t total = 0, o = 0; uint32_t idx = 0, hidx = 0; int i, any_tab = 0;
    if(!s || !b) return MDK_ERR_ARG;
    for(i = 0; i < b->n_ranges; i++) { total += b->range[i].bytes; if(b->range[i].d_rec_off || b->range[i].h_rec_off) any_tab = 1; }
    if(s->raw_cap < total + 64) { free(s->raw); s->raw_cap = total + total / 8 + 64; s->raw = malloc(s->raw_cap); }
    if(s->off_cap < (uint32_t)b->n_records + 1) { free(s->off); s->off_cap = (uint32_t)b->n_records + 1024; s->off = malloc(sizeof(uint32_t) * s->off_cap); }
    if(!s->raw || !s->off) return MDK_ERR_NOMEM;
    for(i = 0; i < b->n_ranges; i++) {
        const md_raw_range *r = &b->range[i]; uint32_t k;
        if(r->bytes) memcpy(s->raw + o, r->ptr, (size_t)r->bytes);
        if(r->d_rec_off || r->h_rec_off) { const uint32_t *t = r->d_rec_off ? r->d_rec_off : r->h_rec_off; for(k = 0; k < r->n_records; k++) s->off[idx + k] = t[k] - r->rec_delta + (uint32_t)o; idx += r->n_records; }
        else if(any_tab) { for(k = 0; k < r->n_records; k++) s->off[idx + k] = b->rec_off[hidx + k]; idx += r->n_records; hidx += r->n_records; }
        o += r->bytes;
    }
    if(!any_tab) { for(idx = 0; idx < (uint32_t)b->n_records; idx++) s->off[idx] = b->rec_off[idx]; }
    if(idx != (uint32_t)b->n_records) { snprintf(t_err, sizeof t_err, "dev_standin: %u record offsets for %d records", idx, b->n_records); return MDK_ERR_ARG; }
    for(idx = 0; idx < (uint32_t)b->n_records; idx++) {       /* every offset names a record inside the bytes, in order, back to back */
        const uint32_t at = s->off[idx]; uint32_t bs;
        if((uint64_t)at + 36 > total) { snprintf(t_err, sizeof t_err, "dev_standin: record %u at %u beyond %llu bytes", idx, at, (unsigned long long)total); return MDK_ERR_ARG; }
        memcpy(&bs, s->raw + at, 4);
        if(idx + 1 < (uint32_t)b->n_records && s->off[idx + 1] != at + 4 + bs) { snprintf(t_err, sizeof t_err, "dev_standin: record %u does not end where record %u begins", idx, idx + 1); return MDK_ERR_ARG; }
        if(idx + 1 == (uint32_t)b->n_records && (uint64_t)at + 4 + bs != total) { snprintf(t_err, sizeof t_err, "dev_standin: the last record does not end with the bytes"); return MDK_ERR_ARG; }
    }
    s->raw_bytes = total; s->n_rec = (uint32_t)b->n_records; s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->used = 1; s->launched = 0; s->handed_back = 0;
    pthread_mutex_lock(&h->mu); h->n_up++; if(h->handback > 0 && h->n_up % h->handback == 0) s->handed_back = 1; pthread_mutex_unlock(&h->mu);
    return 0;
}
int md_dev_upload_raw_inplace(md_dev *h, int slot, const md_raw_batch *b) { return md_dev_upload_raw(h, slot, b); }      /* (the stand-in always copies) */
int md_dev_upload_wait(md_dev *h, int slot) { return slot_of(h, slot) ? 0 : MDK_ERR_ARG; }
int md_dev_upload_done(md_dev *h, int slot) { static __thread unsigned n; return slot_of(h, slot) ? (int)(++n % 3 != 0) : MDK_ERR_ARG; }      /* "not yet" now and then: the caller's waiting path runs too */
int md_dev_upload(md_dev *h, int slot, const md_read_batch *b) { sslot *s = slot_of(h, slot); if(!s || !b) return MDK_ERR_ARG; s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->used = 1; s->launched = 0; s->handed_back = 0; return 0; }
/* "compute time": MDK_STANDIN_US_PER_KREC microseconds per 1000 records of the chunk, one chunk after the other, in the background -- a download
 * waits for what is left of it (tests of the work balance between ranks) */
int md_dev_launch(md_dev *h, int slot) {
    sslot *s = slot_of(h, slot); if(!s || !s->used) return MDK_ERR_ARG;
    if(h->us_per_krec > 0) { const double t = now_s(); pthread_mutex_lock(&h->mu); s->ready_at = (h->busy_until > t ? h->busy_until : t) + 1e-9 * h->us_per_krec * s->n_rec; h->busy_until = s->ready_at; pthread_mutex_unlock(&h->mu); }
    s->launched = 1; return 0;
}
int md_dev_launch_group(md_dev *h, const int *slots, int n) { int i; for(i = 0; i < n; i++) if(md_dev_launch(h, slots[i])) return MDK_ERR_ARG; return 0; }
int md_dev_group_max(void) { return 8; }
int md_dev_submit(md_dev *h, int slot, const md_read_batch *b) { int rc = md_dev_upload(h, slot, b); return rc ? rc : md_dev_launch(h, slot); }
int md_dev_submit_raw(md_dev *h, int slot, const md_raw_batch *b) { int rc = md_dev_upload_raw(h, slot, b); return rc ? rc : md_dev_launch(h, slot); }
int md_dev_read_raw(md_dev *h, int slot, uint8_t *bytes, uint64_t *n_bytes, uint32_t *rec_off, uint32_t *n_records) {
    sslot *s = slot_of(h, slot);
    if(!s || !s->raw || *n_bytes < s->raw_bytes || *n_records < s->n_rec) return MDK_ERR_ARG;
    memcpy(bytes, s->raw, (size_t)s->raw_bytes); memcpy(rec_off, s->off, sizeof(uint32_t) * s->n_rec); *n_bytes = s->raw_bytes; *n_records = s->n_rec;
    return 0;
}
int md_dev_download(md_dev *h, int slot, md_sites *out) {
    sslot *s = slot_of(h, slot); size_t a = 0, b = g_nrow, i; int64_t n = 0; const int variant = h && h->cfg.minOppositeDepth > 0;
    if(!s || !out || !s->launched) { snprintf(t_err, sizeof t_err, "dev_standin: slot not launched"); return MDK_ERR_ARG; }
    memset(out, 0, sizeof(*out));
    { const double t = now_s(); if(s->ready_at > t) usleep((useconds_t)((s->ready_at - t) * 1e6)); }      /* the "device" is still computing this chunk */
    if(s->handed_back) { s->handed_back = 0; snprintf(t_err, sizeof t_err, "dev_standin: this chunk goes back to the host preparation"); return MDK_ERR_PREP_HOST; }
    while(a < b) { const size_t m = (a + b) / 2; if(g_row[m].tid < s->tid || (g_row[m].tid == s->tid && (int64_t)g_row[m].pos < s->beg)) a = m + 1; else b = m; }
    for(i = a; i < g_nrow && g_row[i].tid == s->tid && (int64_t)g_row[i].pos < s->end; i++) n++;
    if(n > s->cap) { free(s->site); free(s->var); s->cap = n + 1024; s->site = malloc(sizeof(md_site) * (size_t)s->cap); s->var = malloc(sizeof(md_site_var) * (size_t)s->cap); if(!s->site || !s->var) return MDK_ERR_NOMEM; }
    for(i = a, n = 0; i < g_nrow && g_row[i].tid == s->tid && (int64_t)g_row[i].pos < s->end; i++) {
        if(!(g_row[i].nm + g_row[i].nu > 0 || (variant && g_row[i].noff > 0))) continue;
        s->site[n].pos = g_row[i].pos; s->site[n].nmeth = g_row[i].nm; s->site[n].nunmeth = g_row[i].nu; s->site[n].meta = g_row[i].meta; s->var[n].noff = g_row[i].noff; s->var[n].nvar = g_row[i].nvar; n++;
    }
    out->n_sites = n; out->site = s->site; out->var = variant ? s->var : NULL;
    return 0;
}
int md_dev_download_group(md_dev *h, const int *slots, int n, md_sites *out, int *rc) { int i; for(i = 0; i < n; i++) rc[i] = md_dev_download(h, slots[i], &out[i]); return 0; }
int md_dev_slot_sync(md_dev *h, int slot) { return slot_of(h, slot) ? 0 : MDK_ERR_ARG; }
int md_dev_sync(md_dev *h) { (void)h; return 0; }
/* the exchange between GPUs is not stood in for: ranks that "share a device" use the command's TCP connections */
int md_comm_unique_id(uint8_t *id) { (void)id; snprintf(t_err, sizeof t_err, "dev_standin: no RCCL"); return MDK_ERR_NODEVICE; }

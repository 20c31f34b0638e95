/* mdk_replicate.c -- TEST/BENCH INFRASTRUCTURE: a larger synthetic input from a smaller one, fast.  mdk_synth draws its reads one after the
 * other (3 Mb of genome per second); the end-to-end leg of bench.py that must outlast start-up and exit needs >= 512 Mb.  This tool writes K
 * copies of a one-contig sample as K contigs: the same bases under K names, the same records with refID / next_refID = k, compressed again
 * into BGZF members that end on record boundaries (as htslib writes them), all of it in parallel.  No index is written.
 *   usage: mdk_replicate IN_PREFIX OUT_PREFIX K      (IN_PREFIX.fa / IN_PREFIX.bam -> OUT_PREFIX.fa / OUT_PREFIX.bam) */
#define _GNU_SOURCE
#include <inttypes.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <zlib.h>
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void *xm(size_t n) { void *p = malloc(n ? n : 1); if(!p) { fprintf(stderr, "mdk_replicate: out of memory\n"); exit(2); } return p; }
typedef struct { size_t in_off, in_len, out_off; uint32_t isz; } mem_t;
static uint8_t *g_raw, *g_data; static mem_t *g_mem; static size_t g_nmem; static volatile size_t g_next; static int g_bad;
static void *inflate_main(void *a) {
    (void)a;
    for(;;) {
        size_t i = __sync_fetch_and_add(&g_next, 1); z_stream zs;
        if(i >= g_nmem) break;
        if(!g_mem[i].isz) continue;
        memset(&zs, 0, sizeof zs); zs.next_in = g_raw + g_mem[i].in_off; zs.avail_in = (uInt)g_mem[i].in_len; zs.next_out = g_data + g_mem[i].out_off; zs.avail_out = g_mem[i].isz;
        if(inflateInit2(&zs, -15) != Z_OK || inflate(&zs, Z_FINISH) != Z_STREAM_END) g_bad = 1;
        inflateEnd(&zs);
    }
    return NULL;
}
/* output members: block j of copy k (k = -1: the header block) */
typedef struct { size_t beg, end; } blk_t;
static blk_t *g_blk; static size_t g_nblk; static int g_K; static uint8_t **g_out; static uint32_t *g_outlen; static volatile size_t g_task; static size_t g_w0, g_w1;
#define WINDOW 16384         /* members compressed per round into buffers that are used again: fresh memory for all 8.7 GB of output costs more than the compression */
static const uint8_t *g_hdr; static size_t g_hdrlen;
static uint8_t *bgzf_member(const uint8_t *d, size_t n, uint32_t *len, uint8_t *o) {
    z_stream zs; uint32_t cl;
    if(!o) o = xm(n + n / 8 + 128);
    static const uint8_t H[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(o, H, 16);
    memset(&zs, 0, sizeof zs); zs.next_in = (Bytef *)d; zs.avail_in = (uInt)n; zs.next_out = o + 18; zs.avail_out = (uInt)(n + n / 8 + 64);
    if(deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK || deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "mdk_replicate: deflate failed\n"); exit(2); }
    cl = (uint32_t)zs.total_out; deflateEnd(&zs);
    if(cl + 26 > 65536) { fprintf(stderr, "mdk_replicate: a block does not fit a BGZF member\n"); exit(2); }
    o[16] = (uint8_t)((cl + 25) & 255); o[17] = (uint8_t)((cl + 25) >> 8);
    put32(o + 18 + cl, (uint32_t)crc32(0L, d, (uInt)n)); put32(o + 22 + cl, (uint32_t)n);
    *len = cl + 26;
    return o;
}
static void *deflate_main(void *a) {
    uint8_t *tmp = xm(65536 + 64);
    (void)a;
    for(;;) {
        size_t t = g_w0 + __sync_fetch_and_add(&g_task, 1), k, j, o, n;
        if(t >= g_w1) break;
        k = t / g_nblk; j = t % g_nblk; n = g_blk[j].end - g_blk[j].beg;
        memcpy(tmp, g_data + g_blk[j].beg, n);
        for(o = 0; o + 4 <= n;) { uint32_t bs = le32(tmp + o); put32(tmp + o + 4, (uint32_t)k); if((int32_t)le32(tmp + o + 4 + 20) >= 0) put32(tmp + o + 4 + 20, (uint32_t)k); o += 4 + bs; }
        (void)bgzf_member(tmp, n, &g_outlen[t - g_w0], g_out[t - g_w0]);
    }
    free(tmp);
    return NULL;
}
int main(int argc, char **argv) {
    char fn[4096]; FILE *f; size_t n, o, total = 0, mcap = 1 << 16, i; int K, nt = (int)sysconf(_SC_NPROCESSORS_ONLN), t; pthread_t th[256];
    uint32_t l_text, n_ref, l_name, ref_len; const char *ref_name; size_t rec0; uint8_t *hdr_member; uint32_t hdr_len; uint8_t *hb;
    if(argc != 4 || (K = atoi(argv[3])) < 1 || K > 1000) { fprintf(stderr, "usage: mdk_replicate IN_PREFIX OUT_PREFIX K\n"); return 1; }
    if(nt > 256) nt = 256;
    if(nt < 1) nt = 1;
    snprintf(fn, sizeof fn, "%s.bam", argv[1]); f = fopen(fn, "rb"); if(!f) { perror(fn); return 1; }
    fseek(f, 0, SEEK_END); n = (size_t)ftell(f); fseek(f, 0, SEEK_SET); g_raw = xm(n + 64); if(fread(g_raw, 1, n, f) != n) return 1; fclose(f);
    g_mem = xm(sizeof(mem_t) * mcap);
    for(o = 0; o + 18 <= n;) {
        uint32_t xlen = g_raw[o + 10] | (g_raw[o + 11] << 8), bs = (g_raw[o + 16] | (g_raw[o + 17] << 8)) + 1u;
        if(g_nmem == mcap) { mcap *= 2; g_mem = realloc(g_mem, sizeof(mem_t) * mcap); if(!g_mem) return 2; }
        g_mem[g_nmem].in_off = o + 12 + xlen; g_mem[g_nmem].in_len = bs - 12 - xlen - 8; g_mem[g_nmem].isz = le32(g_raw + o + bs - 4); g_mem[g_nmem].out_off = total; total += g_mem[g_nmem].isz; g_nmem++;
        o += bs;
    }
    g_data = xm(total + 64);
    for(t = 0; t < nt; t++) pthread_create(&th[t], NULL, inflate_main, NULL);
    for(t = 0; t < nt; t++) pthread_join(th[t], NULL);
    if(g_bad || total < 12 || memcmp(g_data, "BAM\1", 4)) { fprintf(stderr, "mdk_replicate: cannot read %s\n", fn); return 1; }
    free(g_raw);
    l_text = le32(g_data + 4); n_ref = le32(g_data + 8 + l_text);
    if(n_ref != 1) { fprintf(stderr, "mdk_replicate: the input must have one contig\n"); return 1; }
    l_name = le32(g_data + 12 + l_text); ref_name = (const char *)g_data + 16 + l_text; ref_len = le32(g_data + 16 + l_text + l_name); rec0 = 20 + (size_t)l_text + l_name;
    /* blocks of whole records, at most 65280 bytes (htslib's BGZF_BLOCK_SIZE) */
    { size_t cap = total / 60000 + 16, beg = rec0; g_blk = xm(sizeof(blk_t) * cap);
      for(o = rec0; o + 4 <= total;) { size_t rl = 4 + (size_t)le32(g_data + o); if(o + rl - beg > 65280 && o > beg) { g_blk[g_nblk].beg = beg; g_blk[g_nblk].end = o; g_nblk++; beg = o; if(g_nblk + 2 > cap) { cap *= 2; g_blk = realloc(g_blk, sizeof(blk_t) * cap); } } if(rl > 65280) { fprintf(stderr, "mdk_replicate: record larger than a BGZF member\n"); return 1; } o += rl; }
      if(o > beg) { g_blk[g_nblk].beg = beg; g_blk[g_nblk].end = o; g_nblk++; } }
    /* header: K contigs */
    { char txt[1 << 16]; int tl = 0, k; size_t hl;
      tl += snprintf(txt + tl, sizeof(txt) - (size_t)tl, "@HD\tVN:1.6\tSO:coordinate\n");
      for(k = 0; k < K; k++) tl += snprintf(txt + tl, sizeof(txt) - (size_t)tl, "@SQ\tSN:%s_%d\tLN:%" PRIu32 "\n", ref_name, k + 1, ref_len);
      tl += snprintf(txt + tl, sizeof(txt) - (size_t)tl, "@PG\tID:mdk_replicate\tPN:mdk_replicate\tCL:%d copies of %s\n", K, argv[1]);
      hb = xm((size_t)tl + 64 + (size_t)K * (l_name + 32)); memcpy(hb, "BAM\1", 4); put32(hb + 4, (uint32_t)tl); memcpy(hb + 8, txt, (size_t)tl); put32(hb + 8 + tl, (uint32_t)K); hl = 12 + (size_t)tl;
      for(k = 0; k < K; k++) { char nm[300]; int nl = snprintf(nm, sizeof nm, "%s_%d", ref_name, k + 1) + 1; put32(hb + hl, (uint32_t)nl); memcpy(hb + hl + 4, nm, (size_t)nl); put32(hb + hl + 4 + nl, ref_len); hl += 8 + (size_t)nl; }
      if(hl > 65280) { fprintf(stderr, "mdk_replicate: header too large\n"); return 1; }
      g_hdr = hb; g_hdrlen = hl; hdr_member = bgzf_member(g_hdr, g_hdrlen, &hdr_len, NULL); }
    g_K = K; g_out = xm(sizeof(uint8_t *) * WINDOW); g_outlen = xm(sizeof(uint32_t) * WINDOW);
    for(i = 0; i < WINDOW; i++) g_out[i] = xm(65536 + 8192 + 128);
    snprintf(fn, sizeof fn, "%s.bam", argv[2]); f = fopen(fn, "wb"); if(!f) { perror(fn); return 1; }
    { static const uint8_t EOFM[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t bytes = hdr_len;
      setvbuf(f, NULL, _IOFBF, 8u << 20);
      fwrite(hdr_member, 1, hdr_len, f);
      for(g_w0 = 0; g_w0 < (size_t)K * g_nblk; g_w0 = g_w1) {
          g_w1 = g_w0 + WINDOW < (size_t)K * g_nblk ? g_w0 + WINDOW : (size_t)K * g_nblk; g_task = 0;
          for(t = 0; t < nt; t++) pthread_create(&th[t], NULL, deflate_main, NULL);
          for(t = 0; t < nt; t++) pthread_join(th[t], NULL);
          for(i = 0; i < g_w1 - g_w0; i++) { fwrite(g_out[i], 1, g_outlen[i], f); bytes += g_outlen[i]; }
      }
      fwrite(EOFM, 1, 28, f); if(fclose(f)) { perror(fn); return 1; }
      printf("{\"contigs\": %d, \"contig_bp\": %" PRIu32 ", \"members\": %zu, \"bam_bytes\": %" PRIu64 ", \"inflated_bytes\": %zu}\n", K, ref_len, (size_t)K * g_nblk + 2, bytes + 28, g_hdrlen + (size_t)K * (total - rec0)); }
    /* FASTA: the same bases under K names */
    { char *fa; size_t fl, hl; int k; FILE *g; char *nlp;
      snprintf(fn, sizeof fn, "%s.fa", argv[1]); f = fopen(fn, "rb"); if(!f) { perror(fn); return 1; }
      fseek(f, 0, SEEK_END); fl = (size_t)ftell(f); fseek(f, 0, SEEK_SET); fa = xm(fl + 1); if(fread(fa, 1, fl, f) != fl) return 1; fclose(f);
      nlp = memchr(fa, '\n', fl); if(!nlp || fa[0] != '>') { fprintf(stderr, "mdk_replicate: %s is not a one-record FASTA\n", fn); return 1; }
      hl = (size_t)(nlp - fa) + 1;
      snprintf(fn, sizeof fn, "%s.fa", argv[2]); g = fopen(fn, "wb"); if(!g) { perror(fn); return 1; }
      for(k = 0; k < K; k++) { fprintf(g, ">%s_%d copy of %s\n", ref_name, k + 1, argv[1]); fwrite(fa + hl, 1, fl - hl, g); if(fa[fl - 1] != '\n') fputc('\n', g); }
      if(fclose(g)) { perror(fn); return 1; } }
    return 0;
}

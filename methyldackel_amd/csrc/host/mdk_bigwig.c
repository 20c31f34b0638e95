/* mdk_bigwig.c -- minimal bigWig reader (UCSC bbi format, zlib-compressed data blocks, B+ chromosome tree, R-tree
 * index) giving what the reference asks libBigWig for: the chromosome list (bw->cl, extract.c:1074,1100-1103) and one
 * float per base with NaN where there is no data (bwGetValues(..., includeNA=1), extract.c:1123).  libBigWig is not in
 * this image; the layout is restated from the published format. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "mdk_io.h"

static int rd(FILE *f, uint64_t off, void *dst, size_t n) { if(fseeko(f, (off_t)off, SEEK_SET)) return -1; return fread(dst, 1, n, f) == n ? 0 : -1; }
static uint16_t g16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t g32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t g64(const uint8_t *p) { return (uint64_t)g32(p) | ((uint64_t)g32(p + 4) << 32); }

static int chrom_walk(FILE *f, uint64_t off, uint32_t keySize, mdk_bigwig *bw) {
    uint8_t h[4], *item; uint16_t count, i; size_t isz = (size_t)keySize + 8;
    if(rd(f, off, h, 4)) return -1;
    count = g16(h + 2); item = xmalloc(isz * (count ? count : 1));
    if(!item || rd(f, off + 4, item, isz * count)) { free(item); return -1; }
    for(i = 0; i < count; i++) {
        const uint8_t *it = item + isz * i;
        if(h[0]) {      /* leaf: key, chromId, chromSize */
            uint32_t id = g32(it + keySize), len = g32(it + keySize + 4);
            if(bw->n == bw->cap) { bw->cap = bw->cap ? bw->cap * 2 : 64; bw->name = xrealloc(bw->name, sizeof(char *) * bw->cap); bw->len = xrealloc(bw->len, 4 * bw->cap); bw->id = xrealloc(bw->id, 4 * bw->cap); }
            bw->name[bw->n] = xcalloc((size_t)keySize + 1, 1); memcpy(bw->name[bw->n], it, keySize);
            bw->len[bw->n] = len; bw->id[bw->n] = id; bw->n++;
        } else if(chrom_walk(f, g64(it + keySize), keySize, bw)) { free(item); return -1; }
    }
    free(item);
    return 0;
}

mdk_bigwig *mdk_bigwig_open(const char *fn) {
    uint8_t h[64], t[32]; mdk_bigwig *bw; FILE *f = fopen(fn, "rb");
    if(!f) return NULL;
    if(fread(h, 1, 64, f) != 64 || g32(h) != 0x888FFC26u) { fclose(f); return NULL; }
    bw = xcalloc(1, sizeof(*bw)); bw->f = f;
    bw->chrom_tree = g64(h + 8); bw->index = g64(h + 24); bw->uncompress = g32(h + 52);
    if(rd(f, bw->chrom_tree, t, 32) || g32(t) != 0x78CA8C91u || chrom_walk(f, bw->chrom_tree + 32, g32(t + 8), bw)) { mdk_bigwig_close(bw); return NULL; }
    return bw;
}
void mdk_bigwig_close(mdk_bigwig *bw) { uint32_t i; if(!bw) return; if(bw->f) fclose(bw->f); for(i = 0; i < bw->n; i++) free(bw->name[i]); free(bw->name); free(bw->len); free(bw->id); free(bw); }

static int block_apply(mdk_bigwig *bw, uint64_t off, uint64_t size, uint32_t id, uint32_t clen, float *v) {
    uint8_t *raw = xmalloc(size), *buf = raw; size_t n = size, o; uint32_t i, start, step, span, cnt; int type;
    if(!raw || rd(bw->f, off, raw, size)) { free(raw); return -1; }
    if(bw->uncompress) {
        uLongf dl = bw->uncompress;
        buf = xmalloc(dl ? dl : 1);
        if(!buf || uncompress(buf, &dl, raw, size) != Z_OK) { free(raw); free(buf); return -1; }
        n = dl;
    }
    if(n >= 24 && g32(buf) == id) {
        start = g32(buf + 4); step = g32(buf + 12); span = g32(buf + 16); type = buf[20]; cnt = g16(buf + 22);
        for(i = 0, o = 24; i < cnt; i++) {
            uint32_t s, e, k; float val;
            if(type == 1) { if(o + 12 > n) break; s = g32(buf + o); e = g32(buf + o + 4); memcpy(&val, buf + o + 8, 4); o += 12; }
            else if(type == 2) { if(o + 8 > n) break; s = g32(buf + o); e = s + span; memcpy(&val, buf + o + 4, 4); o += 8; }
            else if(type == 3) { if(o + 4 > n) break; s = start + i * step; e = s + span; memcpy(&val, buf + o, 4); o += 4; }
            else break;
            if(e > clen) e = clen;
            for(k = s; k < e; k++) v[k] = val;
        }
    }
    if(buf != raw) free(buf);
    free(raw);
    return 0;
}
static int rtree_walk(mdk_bigwig *bw, uint64_t off, uint32_t id, uint32_t clen, float *v) {
    uint8_t h[4], *item; uint16_t count, i; size_t isz;
    if(rd(bw->f, off, h, 4)) return -1;
    count = g16(h + 2); isz = h[0] ? 32 : 24; item = xmalloc(isz * (count ? count : 1));
    if(!item || rd(bw->f, off + 4, item, isz * count)) { free(item); return -1; }
    for(i = 0; i < count; i++) {
        const uint8_t *it = item + isz * i; uint32_t c0 = g32(it), c1 = g32(it + 8);
        if(id < c0 || id > c1) continue;
        if(h[0]) { if(block_apply(bw, g64(it + 16), g64(it + 24), id, clen, v)) { free(item); return -1; } }
        else if(rtree_walk(bw, g64(it + 16), id, clen, v)) { free(item); return -1; }
    }
    free(item);
    return 0;
}
/* one float per base of chromosome k (NaN where the file has no data); caller frees */
float *mdk_bigwig_values(mdk_bigwig *bw, uint32_t k) {
    uint8_t t[48]; float *v; uint32_t i, clen;
    if(k >= bw->n) return NULL;
    clen = bw->len[k]; v = xmalloc(sizeof(float) * ((size_t)clen + 1));
    if(!v) return NULL;
    for(i = 0; i < clen; i++) v[i] = NAN;
    if(rd(bw->f, bw->index, t, 48) || g32(t) != 0x2468ACE0u || rtree_walk(bw, bw->index + 48, bw->id[k], clen, v)) { free(v); return NULL; }
    return v;
}

/* mdk_pipeline.c -- what happens to the records of a chunk on the host, and the reader/worker pipeline that does it (see mdk_plan.h). */
#include "mdk_plan.h"

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
MDK_LOCAL int ctx_code(const char *seq, int64_t len, int64_t i) {   /* 0 none, 1 CpG, 2 CHG, 3 CHH (sign = direction not needed here) */
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
/* realloc that keeps the old block (and its owner's pointer) when it fails */
static int grow(void **p, size_t bytes) { void *q = realloc(*p, bytes ? bytes : 1); if(!q) return -1; *p = q; return 0; }
static int bb_reserve(batchbuf *b, size_t more_reads, size_t more_blob, size_t more_qn, size_t more_cig) {
    if(b->n + more_reads > b->cap_ri) { size_t nc = (b->n + more_reads) * 2 + 1024; if(grow((void **)&b->ri, nc * sizeof(rinfo))) return -1; b->cap_ri = nc; }
    if(b->blob_len + more_blob > b->cap_blob) {
        size_t nc = (b->blob_len + more_blob) * 2 + (1 << 20); uint8_t *d = md_host_alloc(nc);
        if(!d) return -1;
        if(b->blob_len) memcpy(d, b->blob, b->blob_len);
        md_host_free(b->blob); b->blob = d; b->cap_blob = nc;
    }
    if(b->qn_len + more_qn > b->qn_cap) { size_t nc = (b->qn_len + more_qn) * 2 + 65536; if(grow((void **)&b->qn, nc)) return -1; b->qn_cap = nc; }
    if(b->cig_len + more_cig > b->cig_cap) { size_t nc = (b->cig_len + more_cig) * 2 + 4096; if(grow((void **)&b->cig, nc * 4)) return -1; b->cig_cap = nc; }
    return 0;
}
/* one-shot reservation at the start of a chunk (buffers are empty): no doubling, pinned memory is precious */
static int bb_reserve_exact(batchbuf *b, size_t reads, size_t blob, size_t qn, size_t cig, size_t segs) {
    if(reads > b->cap_ri) { size_t nc = reads + reads / 8; if(grow((void **)&b->ri, nc * sizeof(rinfo))) return -1; b->cap_ri = nc; }
    if(blob > b->cap_blob) { md_host_free(b->blob); b->cap_blob = 0; b->blob = md_host_alloc(blob + blob / 8); if(!b->blob) return -1; b->cap_blob = blob + blob / 8; }
    if(qn > b->qn_cap) { size_t nc = qn + qn / 8; if(grow((void **)&b->qn, nc)) return -1; b->qn_cap = nc; }
    if(cig > b->cig_cap) { size_t nc = cig + cig / 8; if(grow((void **)&b->cig, nc * 4)) return -1; b->cig_cap = nc; }
    if(segs > b->cap_seg) { md_host_free(b->seg); b->cap_seg = 0; b->seg = md_host_alloc((segs + segs / 8) * sizeof(md_seg)); if(!b->seg) return -1; b->cap_seg = segs + segs / 8; }
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
static int qt_prepare(size_t expect) {
    size_t want = 1024;
    while(want < expect + expect / 2 + 16) want <<= 1;
    if(want > t_qt_cap) { free(t_qt); t_qt_cap = 0; t_gen = 0; t_qt = calloc(want, sizeof(qent)); if(!t_qt) return -1; t_qt_cap = want; }
    if(++t_gen == 0x7fffffff) { size_t i; for(i = 0; i < t_qt_cap; i++) t_qt[i].used = 0; t_gen = 1; }     /* a new generation empties the table */
    t_side_n = 0;
    return 0;
}
static int pair_reads(batchbuf *b, int32_t tid) {
    size_t i, n = b->n; int32_t prev_pos = 0; int first = 1;
    if(qt_prepare(n)) return -1;
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
                if(t_side_n == t_side_cap) { size_t nc = t_side_cap ? t_side_cap * 2 : 1024; if(grow((void **)&t_side, sizeof(*t_side) * nc)) return -1; t_side_cap = nc; }
                t_side[t_side_n].end = end; t_side[t_side_n].next = e->more; e->more = (int32_t)++t_side_n;
            }
        }
        prev_pos = pos; first = 0;
    }
    return 0;
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
            if(l > 0) { if(n == *cap) { int nc = *cap ? *cap * 2 : 16; if(grow((void **)out, sizeof(run_t) * (size_t)nc)) return -1; *cap = nc; } (*out)[n].x = x; (*out)[n].y = y; (*out)[n].l = l; n++; }
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
    if(h->n == h->cap) { size_t nc = h->cap ? h->cap * 2 : 256; if(grow((void **)&h->v, nc * sizeof(md_seg))) return -1; h->cap = nc; }
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
        if(no < 0) return -1;
        /* only pairs whose strands agree in parity are resolved against each other (overlaps.c:63-65) */
        if(r->mate >= 0 && (((int)r->strand - (int)b->ri[r->mate].strand) & 1) == 0) {
            m = &b->ri[r->mate];
            nm = cigar_runs(b->cig + m->cig_off, m->ncig, m->pos, (int32_t)m->lq, &rm, &cm);
            if(nm < 0) return -1;
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
    if(need > *cap) { size_t nc = need * 2 + 65536; if(grow((void **)buf, nc)) return -1; *cap = nc; }
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
/* The records of a chunk are a list of RANGES in file order: records parsed in place in a slab a host team inflated (RG_HOST),
 * records copied into the slot's own buffer -- straddlers carried over from the previous chunk (RG_COPY) --, or whole members of a
 * slab inflated on the device (RG_DEV; the device filters the members' records against the chunk itself). */
enum { RG_HOST = 0, RG_COPY = 1, RG_DEV = 2 };
typedef struct { int kind; mdk_slab *slab; size_t beg, end; uint64_t cat0; uint32_t n_rec; int m0, m1; int tab; size_t tab0; } rrange;      /* cat0: offset of its first byte in the concatenation of the chunk's ranges; tab: a host range of whole members, its records are slab->off32[tab0 .. tab0 + n_rec) */
typedef struct pslot {
    int state; mdk_chunk c;
    uint8_t *raw; size_t raw_len, raw_cap;                /* copied records: [u32 len][record bytes]... */
    rrange *rg; int n_rg, cap_rg; uint64_t cat_len;       /* the ranges, in order; bytes in all of them */
    uint64_t n_stream;                                    /* records in the ranges (for up-front reservation) */
    const char *win; int64_t woff, wlen;
    batchbuf bb; int rc;
    /* device preparation: the same records described for md_dev_upload_raw with every host-resident record's offset in the
     * concatenation of the ranges; the slabs stay referenced until the chunk is recycled (the copies read them, and a chunk the
     * device gives up on is prepared from them by mdk_plan_host_prepare) */
    uint32_t *roff; size_t n_roff, cap_roff; md_raw_range *rr; int cap_rr; int hold_slabs, prepared, n_dev_rg;
    int released;                                         /* the slabs behind the ranges were given back early (mdk_plan_release_records) */
    int fallback;                                         /* the chunk is being prepared on the host after all (mdk_plan_host_prepare_from): its slabs are not given back early any more */
    int use_tab;                                          /* whole members of host slabs travel with the slab's own record table instead of an entry per record in roff (not perRead: its emitter looks the kept records up in roff) */
} pslot;

static int roff_push(pslot *sl, uint64_t off) {
    if(sl->n_roff == sl->cap_roff) { size_t nc = sl->cap_roff ? sl->cap_roff * 2 : 1u << 18; if(grow((void **)&sl->roff, nc * sizeof(uint32_t))) return -1; sl->cap_roff = nc; }
    sl->roff[sl->n_roff++] = (uint32_t)off;
    return 0;
}
static rrange *rg_new(mdk_plan *p, pslot *sl, int kind, mdk_slab *slab, size_t beg) {
    rrange *g;
    if(sl->n_rg == sl->cap_rg) { int nc = sl->cap_rg ? sl->cap_rg * 2 : 16; if(grow((void **)&sl->rg, sizeof(rrange) * (size_t)nc)) return NULL; sl->cap_rg = nc; }
    g = &sl->rg[sl->n_rg++]; memset(g, 0, sizeof(*g));
    g->kind = kind; g->slab = slab; g->beg = g->end = beg; g->cat0 = sl->cat_len;
    if(slab) mdk_slab_ref(p->bam, slab);
    return g;
}
static int raw_push(mdk_plan *p, pslot *sl, const mdk_rec *r) {
    size_t need = sl->raw_len + 4 + r->raw_len; rrange *g = sl->n_rg ? &sl->rg[sl->n_rg - 1] : NULL;
    if(need > sl->raw_cap) { size_t nc = need * 2 + (1 << 20); if(grow((void **)&sl->raw, nc)) return -1; sl->raw_cap = nc; }
    if(!(g && g->kind == RG_COPY && g->end == sl->raw_len)) { g = rg_new(p, sl, RG_COPY, NULL, sl->raw_len); if(!g) return -1; }
    if(sl->hold_slabs && roff_push(sl, sl->cat_len)) return -1;
    memcpy(sl->raw + sl->raw_len, &r->raw_len, 4); memcpy(sl->raw + sl->raw_len + 4, r->raw, r->raw_len); sl->raw_len = need;
    g->end = need; g->n_rec++; sl->cat_len += 4 + (uint64_t)r->raw_len;
    return 0;
}
/* one more member of a device slab */
static int dev_push(mdk_plan *p, pslot *sl, mdk_slab *ds, int mi) {
    const mdk_member *m = &ds->mem[mi]; rrange *g = sl->n_rg ? &sl->rg[sl->n_rg - 1] : NULL;
    if(!(g && g->kind == RG_DEV && g->slab == ds && g->m1 == mi)) { g = rg_new(p, sl, RG_DEV, ds, m->off); if(!g) return -1; g->m0 = mi; sl->n_dev_rg++; }
    g->m1 = mi + 1; g->end = (size_t)m->off + m->len; g->n_rec += m->n_sum; sl->cat_len += m->len; sl->n_stream += m->n_sum;
    return 0;
}
/* members of device slabs that the next chunk has to look at again: a list in file order, each with the number of bytes of the
 * host-side carry that precede it there */
static void dm_clear(mdk_plan *p) { int i; for(i = 0; i < p->n_dm; i++) mdk_slab_unref(p->bam, p->dm[i].slab); p->n_dm = 0; }
static int dm2_push(mdk_plan *p, mdk_slab *ds, int mi) {
    if(p->n_dm2 == p->cap_dm2) { int nc = p->cap_dm2 ? p->cap_dm2 * 2 : 64; if(grow((void **)&p->dm2, sizeof(*p->dm2) * (size_t)nc)) return -1; p->cap_dm2 = nc; }
    p->dm2[p->n_dm2].slab = ds; p->dm2[p->n_dm2].mi = mi; p->dm2[p->n_dm2].mark = p->carry2_len; p->n_dm2++;
    mdk_slab_ref(p->bam, ds);
    return 0;
}
/* a member of a device slab against the chunk [beg,end) of contig tid: into the chunk's ranges when one of its records may overlap
 * the chunk (the device tests every record), onto the next chunk's list when one of its records may reach or start beyond `end` */
static int dev_member(mdk_plan *p, pslot *sl, mdk_slab *ds, int mi, int32_t tid, uint32_t beg, uint32_t end, int collect) {
    const mdk_member *m = &ds->mem[mi]; const int32_t tN = m->tidN < 0 ? 0x7fffffff : m->tidN;      /* unplaced records sort behind every contig */
    const int multi = m->tid0 != m->tidN;
    if(m->n_sum == 0 || m->tid0 < 0) return 0;
    if(tN >= tid && (m->tid0 < tid || m->pos0 < (int32_t)end) && (multi || m->max_endp > (int32_t)beg)) { if(collect && dev_push(p, sl, ds, mi)) return -5; }
    if(tN > tid || (tN == tid && (m->max_endp > (int32_t)end || m->posN >= (int32_t)end))) { if(dm2_push(p, ds, mi)) return -5; }
    return 0;
}

/* schedule step + raw collection for one chunk; 1 = produced, 0 = schedule finished, <0 error */
static int reader_fill(mdk_plan *p, pslot *sl) {
    const opts_t *o = &p->o; mdk_bam *bam = p->bam; uint32_t tid, beg, end, tmp; int rc, fi, collect; mdk_rec r; size_t off; mdk_chunk *c = &sl->c;
    memset(c, 0, sizeof(*c)); sl->raw_len = 0; sl->n_rg = 0; sl->n_stream = 0; sl->win = NULL; sl->woff = sl->wlen = 0;
    sl->n_roff = 0; sl->cat_len = 0; sl->prepared = 0; sl->released = 0; sl->fallback = 0; sl->hold_slabs = p->dev_prep; sl->n_dev_rg = 0; sl->use_tab = p->dev_prep && !o->perread && !getenv("MDK_NO_RECTAB");
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
    if(p->claim) { if(!p->claim(p->claim_ctx, c->index)) c->skipped |= MDK_CHUNK_FOREIGN; }
    else if(p->shard_world > 1 && (int)(c->index % (uint32_t)p->shard_world) != p->shard_rank) c->skipped |= MDK_CHUNK_FOREIGN;
    if(p->bed_on && !bed_touches(p, (int32_t)tid, beg, end)) {      /* extract.c:352-369: the chunk is passed over before anything else happens */
        c->skipped |= MDK_CHUNK_BED;
        if(p->bai) { p->need_seek = 1; p->carry_len = 0; p->carry_tid = -1; dm_clear(p); return 1; }      /* do not even read its records */
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
        p->carry_len = 0; p->carry_tid = -1; dm_clear(p);
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
    p->carry2_len = 0; p->n_dm2 = 0;
    {   /* what the previous chunk left: copied records and members of device slabs, merged back into file order */
        int di = 0; const int same = p->carry_tid == (int32_t)tid; const size_t clen = same ? p->carry_len : 0;
        for(off = 0;;) {
            while(di < p->n_dm && (p->dm[di].mark <= off || off >= clen)) { int r2 = dev_member(p, sl, p->dm[di].slab, p->dm[di].mi, (int32_t)tid, beg, end, collect); if(r2) return r2; di++; }
            if(off >= clen) break;
            {
                uint32_t len; int32_t rlen, endp;
                memcpy(&len, p->carry + off, 4);
                if(mdk_rec_parse(p->carry + off + 4, len, &r) != 0) return -2;
                off += 4 + (size_t)len;
                rlen = cigar_ref_len(&r); endp = r.pos + (rlen > 0 ? rlen : 1);
                c->n_records_seen++;
                if(endp > (int32_t)beg && r.pos < (int32_t)end && collect) { if(raw_push(p, sl, &r)) return -5; }
                if((uint32_t)endp > end && carry_push(&p->carry2, &p->carry2_len, &p->carry2_cap, &r)) return -5;
            }
        }
        dm_clear(p);
    }
    for(;;) {
        mdk_rsum q; const uint8_t *raw;
        {   /* a slab inflated on the device is taken member by member, from the digests */
            mdk_slab *ds; int mi; int k = mdk_bam_at_device(bam, &ds, &mi);
            if(k < 0) { rc = k; break; }
            if(k == 1) {
                const mdk_member *m = &ds->mem[mi]; rc = 1;
                if(m->n_sum && m->tid0 >= 0) {
                    if(m->tid0 < p->last_tid || (m->tid0 == p->last_tid && m->pos0 < p->last_pos) || (!m->sorted && m->tidN >= 0)) { fprintf(stderr, "[mdk] %s is not coordinate sorted; `extract` needs sorted alignments\n", o->bam_name); return -2; }
                    if(m->tid0 > (int32_t)tid || (m->tid0 == (int32_t)tid && m->pos0 >= (int32_t)end)) break;
                    c->n_records_seen += m->n_sum;
                    { int r2 = dev_member(p, sl, ds, mi, (int32_t)tid, beg, end, collect); if(r2) return r2; }
                    if(m->tidN >= 0) { p->last_tid = m->tidN; p->last_pos = m->posN; }
                }
                mdk_bam_dev_advance(bam);
                continue;
            }
        }
        rc = mdk_bam_peek_sum(bam, &q, &raw);
        if(rc == 2) continue;
        if(rc != 1) break;
        {   /* a whole (rest of a) BGZF member at once, from the digest its inflating thread left: every record on this contig,
             * in order, starting before the chunk's end, reaching into the chunk, none reaching beyond it */
            const mdk_rsum *v; size_t n; const mdk_member *m;
            if(mdk_bam_member_run(bam, &v, &n, &m) && n > 1 && m->sorted && m->tid0 == (int32_t)tid && m->tidN == (int32_t)tid && m->posN < (int32_t)end &&
               m->min_endp > (int32_t)beg && m->max_endp <= (int32_t)end && !(v[0].tid < p->last_tid || (v[0].tid == p->last_tid && v[0].pos < p->last_pos))) {
                c->n_records_seen += n;
                if(collect) {
                    size_t roff; mdk_slab *cs = mdk_bam_cur_slab(bam, &roff); rrange *g = sl->n_rg ? &sl->rg[sl->n_rg - 1] : NULL; size_t k, rend = (size_t)v[n - 1].off + 4 + v[n - 1].len;
                    const size_t t0 = (size_t)(v - cs->sum); const int tab = sl->use_tab && cs->off32 != NULL;
                    if(!(g && g->kind == RG_HOST && g->slab == cs && g->end == roff && g->tab == tab && (!tab || g->tab0 + g->n_rec == t0))) { g = rg_new(p, sl, RG_HOST, cs, roff); if(!g) return -5; g->tab = tab; g->tab0 = t0; }
                    g->end = rend; g->n_rec += (uint32_t)n; sl->cat_len = g->cat0 + (g->end - g->beg);
                    sl->n_stream += n;
                    if(sl->hold_slabs && !tab) {
                        const uint64_t base = g->cat0 - g->beg;
                        if(sl->n_roff + n > sl->cap_roff) { size_t nc = (sl->n_roff + n) * 2 + (1u << 18); if(grow((void **)&sl->roff, nc * sizeof(uint32_t))) return -5; sl->cap_roff = nc; }
                        for(k = 0; k < n; k++) sl->roff[sl->n_roff + k] = (uint32_t)(base + v[k].off);
                        sl->n_roff += n;
                    }
                }
                p->last_tid = (int32_t)tid; p->last_pos = v[n - 1].pos;
                mdk_bam_advance_run(bam, n);
                continue;
            }
        }
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
                if(!(g && g->kind == RG_HOST && g->slab == cs && g->end == roff && !g->tab)) { g = rg_new(p, sl, RG_HOST, cs, roff); if(!g) return -5; }
                if(sl->hold_slabs && roff_push(sl, g->cat0 + (roff - g->beg))) return -5;
                g->end = roff + 4 + q.len; g->n_rec++; sl->cat_len = g->cat0 + (g->end - g->beg);
                sl->n_stream++;
            }
            if((uint32_t)q.endp > end) { r.raw = raw; r.raw_len = q.len; if(carry_push(&p->carry2, &p->carry2_len, &p->carry2_cap, &r)) return -5; }
        }
        mdk_bam_advance_sum(bam, &q);
    }
    if(rc < 0) { fprintf(stderr, "[mdk] error while reading %s: %s\n", o->bam_name, bam->err); return -2; }
    { uint8_t *t = p->carry; size_t tc = p->carry_cap; p->carry = p->carry2; p->carry_len = p->carry2_len; p->carry_cap = p->carry2_cap; p->carry2 = t; p->carry2_cap = tc; p->carry2_len = 0; p->carry_tid = (int32_t)tid; }
    { void *t = p->dm; int tc = p->cap_dm; p->dm = p->dm2; p->n_dm = p->n_dm2; p->cap_dm = p->cap_dm2; p->dm2 = t; p->cap_dm2 = tc; p->n_dm2 = 0; }
    return 1;
}

/* admission + packing + pairing + segments of one chunk */
static int worker_process(mdk_plan *p, pslot *sl) {
    batchbuf *b = &sl->bb; mdk_chunk *c = &sl->c; size_t off; mdk_rec r; double t0 = now_s(), t1, t2;
    int g; size_t bytes = 0; int32_t rl;
    b->n = 0; b->blob_len = 0; b->qn_len = 0; b->cig_len = 0; b->n_seg = 0; b->algo_bytes = 0;
    /* one reservation per chunk instead of growing (the blob is staging memory, which is expensive to allocate): the
     * payload, names and CIGARs of the admitted reads are all smaller than the raw records they come from */
    for(g = 0; g < sl->n_rg; g++) { if(sl->rg[g].kind == RG_DEV) { fprintf(stderr, "[mdk] internal: a chunk inflated on the device reached the host preparation without its bytes\n"); return -2; } bytes += sl->rg[g].end - sl->rg[g].beg; }
    if(bb_reserve_exact(b, (size_t)sl->n_stream + c->n_records_seen + 16, bytes - bytes / 8 + 65536, bytes / 4 + 4096, bytes / 16 + 4096, 2 * (size_t)sl->n_stream + 4096)) return -5;
    for(g = 0; g < sl->n_rg; g++) {
        const uint8_t *base = sl->rg[g].kind == RG_COPY ? sl->raw : sl->rg[g].slab->buf;
        for(off = sl->rg[g].beg; off < sl->rg[g].end;) {
            uint32_t len; memcpy(&len, base + off, 4);
            if(mdk_rec_parse(base + off + 4, len, &r) != 0) return -2;
            off += 4 + (size_t)len;
            rl = cigar_ref_len(&r);
            /* the region query of the chunk (sam_itr_queryi behind extract.c:379: this contig, pos < end, bam_endpos > beg).  Host ranges hold
             * exactly its records already; the records of a chunk handed back by the device (mdk_plan_host_prepare_from) are whole BGZF
             * members, which carry the neighbouring chunk's -- and at a contig's first or last member the neighbouring contig's -- records */
            if(!p->o.perread && (r.tid != c->tid || (int64_t)r.pos >= c->end || (int64_t)r.pos + (rl > 0 ? rl : 1) <= c->beg)) continue;
            if(admit_and_pack(p, b, &r, rl, sl->win, sl->woff, sl->wlen, c->beg, c->end) < 0) return -5;
        }
        if(!sl->hold_slabs && sl->rg[g].slab) mdk_slab_unref(p->bam, sl->rg[g].slab);
    }
    if(!sl->hold_slabs) sl->n_rg = 0;
    if(bb_reserve(b, 1, 16, 16, 1) || seg_reserve(b, 1)) return -5;          /* never hand out NULL arrays */
    t1 = now_s();
    if(p->o.perread) {        /* no pairing, no segments: the device walks each read's CIGAR itself */
        size_t i;
        if(b->cap_pr < b->n + 1) { free(b->pr); b->cap_pr = 0; b->pr = malloc(sizeof(md_pr_read) * (b->n + 1) * 2); if(!b->pr) return -5; b->cap_pr = (b->n + 1) * 2; }
        for(i = 0; i < b->n; i++) {
            const rinfo *ri = &b->ri[i]; md_pr_read *q = &b->pr[i];
            q->pos = ri->pos; q->off4 = ri->off4; q->l_qseq = ri->lq; q->cig_off = ri->cig_off; q->n_cigar = ri->ncig; q->strand = ri->strand; q->reserved = 0;
        }
        c->pr.tid = c->tid; c->pr.beg = c->beg; c->pr.end = c->end; c->pr.n_reads = (int32_t)b->n; c->pr.read = b->pr; c->pr.cigar = b->cig; c->pr.n_cigar = b->cig_len;
        c->pr.blob = b->blob; c->pr.blob_bytes = b->blob_len; c->host = b;
        pthread_mutex_lock(&p->mu); p->t_collect += t1 - t0; pthread_mutex_unlock(&p->mu);
        return 0;
    }
    if(!p->o.mbias && pair_reads(b, c->tid)) return -5;          /* mbias installs no overlap handler (MBias.c:158-161): every read counts on its own */
    t2 = now_s();
    if(build_segments(p, b, c->beg, c->end)) return -5;
    c->batch.tid = c->tid; c->batch.beg = c->beg; c->batch.end = c->end; c->batch.n_segs = (int32_t)b->n_seg; c->batch.seg = b->seg;
    c->batch.blob = b->blob; c->batch.blob_bytes = b->blob_len; c->batch.n_reads = (int32_t)b->n; c->batch.algo_bytes = b->algo_bytes;
    pthread_mutex_lock(&p->mu); p->t_collect += t1 - t0; p->t_pair += t2 - t1; p->t_segs += now_s() - t2; pthread_mutex_unlock(&p->mu);
    return 0;
}

/* describe the chunk's records for md_dev_upload_raw (no per-record work on the host) */
static int describe_raw(mdk_plan *p, pslot *sl) {
    mdk_chunk *c = &sl->c; int g; uint64_t nrec = 0;
    if(sl->n_rg + 1 > sl->cap_rr) { int nc = sl->n_rg + 8; if(grow((void **)&sl->rr, sizeof(md_raw_range) * (size_t)nc)) return -5; sl->cap_rr = nc; }
    for(g = 0; g < sl->n_rg; g++) {
        const rrange *q = &sl->rg[g]; md_raw_range *o = &sl->rr[g];
        o->bytes = q->end - q->beg; o->n_records = q->n_rec; o->d_rec_off = NULL; o->h_rec_off = NULL; o->rec_delta = 0;
        if(q->kind == RG_COPY) o->ptr = sl->raw + q->beg;
        else if(q->kind == RG_HOST) { o->ptr = q->slab->buf + q->beg; if(q->tab) { o->h_rec_off = q->slab->off32 + q->tab0; o->rec_delta = (uint32_t)q->beg; } }
        else { o->ptr = q->slab->d_buf + q->beg; o->d_rec_off = q->slab->d_rec_off + q->slab->mem[q->m0].sum0; o->rec_delta = (uint32_t)q->beg; }
        nrec += q->n_rec;
    }
    if(sl->cat_len >= 0xffffff00ull) { fprintf(stderr, "[mdk] a chunk holds more than 4 GiB of records; use a smaller --chunkSize\n"); return -2; }
    c->raw.tid = c->tid; c->raw.beg = c->beg; c->raw.end = c->end; c->raw.n_ranges = sl->n_rg; c->raw.range = sl->rr;
    c->raw.n_records = (int32_t)nrec; c->raw.rec_off = sl->roff; c->raw.woff = sl->woff; c->raw.wlen = sl->wlen;
    c->prep = 1;
    (void)p;
    return 0;
}
static void slot_release_slabs(mdk_plan *p, pslot *sl) {
    int g;
    if(!sl->hold_slabs) return;
    for(g = 0; g < sl->n_rg; g++) if(sl->rg[g].slab) mdk_slab_unref(p->bam, sl->rg[g].slab);
    sl->n_rg = 0;
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
        {   /* perRead lists the reads of a contig the FASTA lacks with zero calls (perRead.c:176): no device work, the host selects them */
            const int host_listing = p->o.perread && sl->c.skipped == MDK_CHUNK_NOREF;
            if(rc == 1 && p->dev_prep && !sl->c.skipped) { int r2 = describe_raw(p, sl); if(r2 < 0) rc = r2; }
            if(rc == 1 && p->dev_prep && !host_listing) { sl->rc = 0; sl->state = S_DONE; pthread_cond_broadcast(&p->cv_done); }     /* nothing to do per record on the host */
            else if(rc == 1) { sl->state = S_RAW; pthread_cond_signal(&p->cv_raw); }
            else { sl->state = S_FREE; if(rc < 0) p->pipe_rc = rc; p->reader_done = 1; pthread_cond_broadcast(&p->cv_raw); pthread_cond_broadcast(&p->cv_done); }
        }
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
static int slot_state_of(const mdk_plan *p, int k) { return p->slot[k].state; }
MDK_LOCAL int pipeline_start(mdk_plan *p) {
    int i;
    p->slot_state = slot_state_of;
    /* beyond a dozen workers the serial reader is the limit, and every slot pins ~1.2 bytes of host memory per raw BAM
     * byte of its chunk (expensive to allocate), so the pipeline depth is bounded; -@ still sizes the inflate pool */
    p->n_workers = p->o.n_threads < 1 ? 1 : p->o.n_threads;
    { int cap = getenv("MDK_WORKERS") ? atoi(getenv("MDK_WORKERS")) : 12; if(cap < 1) cap = 1; if(p->claim && cap > 2) cap = 2;      /* a rank that claims its chunks must not claim far ahead of what it computes */
      if(p->n_workers > cap) p->n_workers = cap; }
    if(p->n_hold < 2) p->n_hold = 2;
    p->n_slot = p->n_workers + 1 + p->n_hold;
    p->slot = calloc((size_t)p->n_slot, sizeof(pslot));
    p->worker_th = calloc((size_t)p->n_workers, sizeof(pthread_t));
    if(!p->slot || !p->worker_th) return -5;
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_free, NULL); pthread_cond_init(&p->cv_raw, NULL); pthread_cond_init(&p->cv_done, NULL);
    if(p->n_hold < 2) p->n_hold = 2;
    for(i = 0; i < p->n_hold; i++) p->held[i] = -1;
    p->next_out = 0;
    /* workers first: fewer than asked for is fine (they all take chunks from the same queue), none is not */
    for(i = 0; i < p->n_workers; i++) if(pthread_create(&p->worker_th[i], NULL, worker_main, p)) break;
    if(i < p->n_workers) { if(i == 0) { fprintf(stderr, "[mdk] cannot create a worker thread\n"); goto fail; } p->n_workers = i; }
    if(pthread_create(&p->reader_th, NULL, reader_main, p)) {
        fprintf(stderr, "[mdk] cannot create the reader thread\n");
        pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->cv_raw); pthread_mutex_unlock(&p->mu);
        for(i = 0; i < p->n_workers; i++) pthread_join(p->worker_th[i], NULL);
        goto fail;
    }
    p->started = 1;
    return 0;
fail:
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_free); pthread_cond_destroy(&p->cv_raw); pthread_cond_destroy(&p->cv_done);
    free(p->slot); free(p->worker_th); p->slot = NULL; p->worker_th = NULL;
    return -5;
}
MDK_LOCAL void pipeline_stop(mdk_plan *p) {
    int i;
    if(!p->started) return;
    pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->cv_free); pthread_cond_broadcast(&p->cv_raw); pthread_cond_broadcast(&p->cv_done); pthread_mutex_unlock(&p->mu);
    mdk_bam_abort(p->bam);          /* wake the reader if it is waiting for inflated data */
    pthread_join(p->reader_th, NULL);
    for(i = 0; i < p->n_workers; i++) pthread_join(p->worker_th[i], NULL);
    dm_clear(p); { int k; for(k = 0; k < p->n_dm2; k++) mdk_slab_unref(p->bam, p->dm2[k].slab); p->n_dm2 = 0; }
    for(i = 0; i < p->n_slot; i++) { int g; for(g = 0; g < p->slot[i].n_rg; g++) if(p->slot[i].rg[g].slab) mdk_slab_unref(p->bam, p->slot[i].rg[g].slab); bb_free(&p->slot[i].bb); free(p->slot[i].raw); free(p->slot[i].rg); free(p->slot[i].roff); free(p->slot[i].rr); }
    free(p->slot); free(p->worker_th); p->slot = NULL; p->started = 0;
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_free); pthread_cond_destroy(&p->cv_raw); pthread_cond_destroy(&p->cv_done);
}

static int next_chunk_ex(mdk_plan *p, mdk_chunk *c, int nonblock);
int mdk_plan_next_chunk(mdk_plan *p, mdk_chunk *c) { return next_chunk_ex(p, c, 0); }
/* the same, but 2 (and nothing handed out) when the next chunk of the schedule is not ready yet */
int mdk_plan_try_next_chunk(mdk_plan *p, mdk_chunk *c) { return next_chunk_ex(p, c, 1); }
static int next_chunk_ex(mdk_plan *p, mdk_chunk *c, int nonblock) {
    int i, found = -1, rc = 0;
    if(!p->started && pipeline_start(p)) return -5;
    pthread_mutex_lock(&p->mu);
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
        if(nonblock) { pthread_mutex_unlock(&p->mu); return 2; }
        pthread_cond_wait(&p->cv_done, &p->mu);
    }
    if(found >= 0 && p->slot[found].rc >= 0 && p->slot[found].c.skipped && !(p->o.perread && p->slot[found].c.skipped == MDK_CHUNK_NOREF)) {
        /* a chunk that was passed over (another rank's, no BED region, no reference) carries nothing the caller could refer to later: its slot
         * is free again at once and it does not push an older chunk out of the caller's hands -- a rank may see any number of other ranks'
         * chunks between two of its own */
        pslot *sl = &p->slot[found];
        *c = sl->c; rc = 1;
        slot_release_slabs(p, sl); sl->state = S_FREE; p->next_out++;
        pthread_cond_signal(&p->cv_free);
    } else
    if(found >= 0) {
        pslot *sl = &p->slot[found];
        {   /* the oldest chunk still held is no longer referenced by the caller: recycle its buffers */
            const int last = p->n_hold - 1;
            if(p->held[last] >= 0) { slot_release_slabs(p, &p->slot[p->held[last]]); p->slot[p->held[last]].state = S_FREE; pthread_cond_signal(&p->cv_free); }
            for(i = last; i > 0; i--) p->held[i] = p->held[i - 1];
            p->held[0] = -1;
        }
        if(sl->rc < 0) rc = sl->rc; else { *c = sl->c; rc = 1; }
        sl->state = S_HELD; p->held[0] = found; p->next_out++;
    }
    pthread_mutex_unlock(&p->mu);
    if(rc <= 0) memset(c, 0, sizeof(*c));
    return rc;
}


/* A chunk handed out for device preparation, prepared on the host after all (the device reported MDK_ERR_PREP_HOST): the
 * same admission, pairing and segments as a host-mode plan produces, from the records the slot still references. */
static pslot *held_slot(mdk_plan *p, const mdk_chunk *c) {      /* (another thread may be taking the next chunk meanwhile) */
    int i; pslot *sl = NULL;
    pthread_mutex_lock(&p->mu);
    for(i = 0; i < p->n_hold; i++) if(p->held[i] >= 0 && p->slot[p->held[i]].c.index == c->index) { sl = &p->slot[p->held[i]]; break; }
    pthread_mutex_unlock(&p->mu);
    return sl;
}
static pthread_mutex_t rel_mu = PTHREAD_MUTEX_INITIALIZER;      /* (what it is for: below, at mdk_plan_host_prepare_from) */
int mdk_plan_host_prepare(mdk_plan *p, mdk_chunk *c) {
    int rc, done; pslot *sl = NULL;
    if(!p || !c || !p->started) return -1;
    sl = held_slot(p, c);
    if(!sl || !sl->hold_slabs) return -1;
    pthread_mutex_lock(&rel_mu); done = sl->prepared; pthread_mutex_unlock(&rel_mu);      /* (mdk_plan_release_records looks at it from the uploader thread) */
    if(!done) { rc = worker_process(p, sl); if(rc < 0) return rc; pthread_mutex_lock(&rel_mu); sl->prepared = 1; pthread_mutex_unlock(&rel_mu); }
    c->batch = sl->c.batch;
    return 0;
}
/* the same for a chunk some of whose records were inflated on the device and so never were in host memory: the records as the
 * device slot holds them (the concatenation of the chunk's ranges) come back first and stand in for the ranges */
/* The uploader thread gives a chunk's slabs back as soon as its records have crossed the link (mdk_plan_release_records) while the collector
 * thread may be handed the same chunk back by the device (MDK_ERR_PREP_HOST): the two meet under rel_mu.  Whoever comes first decides --
 * released: the fallback reads the records back from the device slot; fallback first: the slabs stay referenced until the chunk is recycled
 * and are parsed where they lie. */
int mdk_plan_host_prepare_from(mdk_plan *p, mdk_chunk *c, md_dev *dev, int slot) {
    pslot *sl; int from_device;
    if(!p || !c || !p->started) return -1;
    pthread_mutex_lock(&rel_mu);
    sl = held_slot(p, c);
    if(!sl || !sl->hold_slabs) { pthread_mutex_unlock(&rel_mu); return -1; }
    sl->fallback = 1;
    from_device = !sl->prepared && (sl->n_dev_rg || sl->released);
    pthread_mutex_unlock(&rel_mu);
    if(from_device) {
        uint64_t nb = sl->cat_len + 64; uint32_t nr = (uint32_t)c->raw.n_records + 1; rrange *g;
        if(nb > sl->raw_cap) { if(grow((void **)&sl->raw, nb)) return -5; sl->raw_cap = nb; }
        if(nr > sl->cap_roff) { if(grow((void **)&sl->roff, sizeof(uint32_t) * (size_t)nr)) return -5; sl->cap_roff = nr; }
        if(md_dev_read_raw(dev, slot, sl->raw, &nb, sl->roff, &nr)) return -2;
        slot_release_slabs(p, sl);
        sl->raw_len = (size_t)nb; sl->n_roff = nr; sl->cat_len = 0; sl->n_dev_rg = 0; sl->released = 0;
        g = rg_new(p, sl, RG_COPY, NULL, 0); if(!g) return -5;
        g->end = (size_t)nb; g->n_rec = nr; sl->cat_len = nb;
    }
    return mdk_plan_host_prepare(p, c);
}

int mdk_plan_release_records(mdk_plan *p, const mdk_chunk *c) {
    pslot *sl; int rc = -1;
    if(!p || !c || !p->started) return -1;
    pthread_mutex_lock(&rel_mu);
    sl = held_slot(p, c);
    if(sl && sl->hold_slabs && !sl->prepared && !sl->fallback) {
        if(sl->n_rg) { slot_release_slabs(p, sl); sl->released = 1; }
        rc = 0;
    }
    pthread_mutex_unlock(&rel_mu);
    return rc;
}

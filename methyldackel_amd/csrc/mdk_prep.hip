// mdk_prep.hip -- chunk preparation on the device: from the inflated BAM records of a chunk, as they lie in the file, to
// the segment array k_pileup consumes.  What the reference does for this inside htslib's iterator and pileup buffer:
//
//   k_rec_scan   one lane per BAM record: the record's fields, its CIGAR (reference length, bam_cigar2rlen), the aux walk
//                for NH and XG (bam_aux_get), getStrand (common.c:84-116) and filter_func's admission tests in its order
//                (common.c:416-444: unmapped, MAPQ, -F, -R, duplicates, NH, mappability windows, singleton, discordant,
//                BED span, conversion efficiency); a 64-bit hash of the read name for the pairing.
//   k_compact    stream compaction of the admitted records, file order kept (it is the order bam_plp_push sees them in),
//                and insertion into a name-keyed hash table (what khash does in custom_overlap_constructor).
//   k_pair       one lane per name: the records of a name in file order go through the constructor/destructor state
//                machine of overlaps.c:121-147 *including* htslib's buffer eviction (a read leaves the pileup buffer once a
//                later read starts beyond its end, and its destructor erases the name): who is resolved against whom.
//   k_segments   one lane per admitted read: CIGAR -> gapless runs (calculate_positions, overlaps.c:27-52; htslib
//                resolve_cigar2), cut where the partner's runs begin and end, clipped to the chunk; counted, scanned,
//                written; every segment also widens the [first,last) run of the tiles it touches.
//
// Nothing is copied: segments address sequence and qualities inside the uploaded record bytes (MDK layout 1, see
// KParams::unit/packed in mdk_hip.hip).  A name with more records than a lane keeps in registers, or more live reads than
// its window holds, sets a flag and the host prepares that chunk the slow way (mdk_pipeline.c) -- same segments either way.
#include <algorithm>
#include "mdk_hip_internal.hpp"

#define PB 256                       // threads per block of the per-record kernels
#define MAXG 16                      // records of one name a lane sorts in registers
#define MAXLIVE 8                    // reads of one name alive in the pileup buffer at once

__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct PrepParams {
    const uint8_t *raw; uint64_t raw_bytes; const uint32_t *rec_off; int n_rec;
    md_prep_cfg cfg; int32_t tid; int64_t beg, end, woff, wlen;
    const char *ref; int64_t reflen;                 // contig letters (conversion efficiency)
    const uint32_t *mapbits; int64_t maplen;         // 1 bit per base, or NULL
    const md_region *runs; int64_t nruns; int bed_on;
    PrepRec *rec; uint64_t *hash;                    // per candidate record
    uint32_t *blockcnt, *blockoff; int nblocks;
    PrepRead *rd; int32_t *mate; uint8_t *second; uint32_t *aidx;    // per admitted read (aidx: its index among the candidate records)
    uint64_t *hkey; int32_t *hhead, *hnext; uint32_t hmask;
    uint32_t *segcnt; uint32_t *segblk, *segblkoff;  // per admitted read / per block
    md_seg *seg; int64_t cap_seg;
    TileEnt *tiles; int ntiles, tile;
    PrepCounters *cnt;
};

// ---- aux area: first NH and first XG, as bam_aux_get finds them; a malformed area ends the walk ----
__device__ void scan_aux(const uint8_t *s, const uint8_t *e, const uint8_t *&nh, const uint8_t *&xg) {
    nh = nullptr; xg = nullptr;
    while(e - s >= 3) {
        const uint8_t *ty = s + 2, *v = s + 3; size_t sz;
        const uint8_t t = *ty;
        if(t == 'A' || t == 'c' || t == 'C') sz = 1;
        else if(t == 's' || t == 'S') sz = 2;
        else if(t == 'i' || t == 'I' || t == 'f') sz = 4;
        else if(t == 'd') sz = 8;
        else if(t == 'Z' || t == 'H') { const uint8_t *z = v; while(z < e && *z) z++; if(z >= e) return; sz = (size_t)(z - v) + 1; }
        else if(t == 'B') {
            size_t es; if(e - v < 5) return;
            const uint8_t st = v[0];
            if(st == 'c' || st == 'C') es = 1; else if(st == 's' || st == 'S') es = 2; else if(st == 'i' || st == 'I' || st == 'f') es = 4; else return;
            sz = 5 + es * (size_t)ld32(v + 1);
        } else return;
        if((size_t)(e - v) < sz) return;
        if(s[0] == 'N' && s[1] == 'H' && !nh) nh = ty;
        else if(s[0] == 'X' && s[1] == 'G' && !xg) xg = ty;
        s = v + sz;
    }
}
__device__ __forceinline__ int64_t aux_int(const uint8_t *ty) {
    switch(*ty) {
    case 'c': return (int8_t)ty[1]; case 'C': return ty[1];
    case 's': return (int16_t)ld16(ty + 1); case 'S': return ld16(ty + 1);
    case 'i': return (int32_t)ld32(ty + 1); case 'I': return ld32(ty + 1);
    }
    return 0;
}
__device__ __forceinline__ int strand_of(uint32_t flag, const uint8_t *xg) {        // common.c:84-116
    int conv = 0;
    if(xg && (xg[1] == 'C' || xg[1] == 'G')) conv = xg[1];
    if(!conv) {
        if(!(flag & 0x1)) return (flag & 0x10) ? 2 : 1;
        if((flag & 0x50) == 0x50) return 2;
        if(flag & 0x40) return 1;
        if((flag & 0x90) == 0x90) return 1;
        if(flag & 0x80) return 2;
        return 0;
    }
    int fwdlike;
    if((flag & 0x51) == 0x41) fwdlike = 1;
    else if((flag & 0x51) == 0x51) fwdlike = 0;
    else if((flag & 0x91) == 0x81) fwdlike = 0;
    else if((flag & 0x91) == 0x91) fwdlike = 1;
    else fwdlike = !(flag & 0x10);
    if(conv == 'C') return fwdlike ? 1 : 3;
    return fwdlike ? 4 : 2;
}

// check_mappability (common.c:277-335) on a 1-bit-per-base track: does [start, start+l) hold at least `need` mappable bases,
// counted in a signed char as the reference does (a count that passes 127 wraps and never passes)
__device__ bool map_window_passes(const PrepParams &P, int64_t start, int l) {
    const int need = P.cfg.min_mappable;
    if(l <= 0) return false;
    if(need <= 0) return true;
    if(need > 127) return false;                       // the running count is a signed char: it never gets there
    if(start < 0 || !P.mapbits) return false;          // a negative start is a huge uint32 in the reference: past the array
    const int64_t nbits = ((P.maplen + 7) / 8) * 8;    // the track is stored in whole bytes; bits past it read as 0
    int64_t end = start + l; if(end > nbits) end = nbits;
    int n = 0;
    for(int64_t p = start; p < end;) {
        const int64_t w = p >> 5; const int b = (int)(p & 31); int take = 32 - b; if(take > end - p) take = (int)(end - p);
        uint32_t word = P.mapbits[w] >> b; if(take < 32) word &= (1u << take) - 1u;
        n += __popc(word); p += take;
    }
    return n >= need;
}

__device__ bool bed_touches(const PrepParams &P, int64_t beg, int64_t end) {     // any run overlapping [beg, end)
    int64_t a = 0, b = P.nruns;
    while(a < b) { const int64_t m = (a + b) >> 1; if((int64_t)P.runs[m].end <= beg) a = m + 1; else b = m; }
    return a < P.nruns && (int64_t)P.runs[a].start < end;
}

__device__ __forceinline__ int ctx_code_win(const char *win, int64_t len, int64_t i) {    // 0 none, 1 CpG, 2 CHG, 3 CHH, inside the chunk's window only
    const char c = win[i] & 0x5f;
    if(c == 'C') { if(i + 1 < len && (win[i + 1] & 0x5f) == 'G') return 1; if(i + 2 < len && (win[i + 2] & 0x5f) == 'G') return 2; return 3; }
    if(c == 'G') { if(i > 0 && (win[i - 1] & 0x5f) == 'C') return 1; if(i > 1 && (win[i - 2] & 0x5f) == 'C') return 2; return 3; }
    return 0;
}
// computeConversionEfficiency (common.c:338-404), including that `pos` is not advanced after an M run
__device__ float conv_efficiency(const PrepParams &P, const uint8_t *cig, int ncig, int32_t rpos, const uint8_t *seq, const uint8_t *qual, int lq, int strand, int *err) {
    unsigned nm = 0, nu = 0; int64_t pos = rpos; int q = 0;
    const char *win = P.ref + P.woff;
    for(int k = 0; k < ncig; k++) {
        const uint32_t c = ld32(cig + 4 * k); const int op = c & 15, len = (int)(c >> 4);
        if(op == 0 || op == 7 || op == 8) {
            for(int j = 0; j < len; j++, q++) {
                const int64_t wi = pos + j - P.woff;
                if(pos + j >= P.woff + P.wlen) goto done;
                if(wi < 0) continue;
                const int ctx = ctx_code_win(win, P.wlen, wi);
                if(ctx < 2) continue;
                if(strand == 0) { *err = 1; return 1.0f; }
                if(q >= lq || qual[q] < P.cfg.min_phred) continue;
                const int b = (seq[q >> 1] >> ((~q & 1) << 2)) & 15;
                if(strand & 1) { if(b == 2) nm++; else if(b == 8) nu++; }
                else { if(b == 4) nm++; else if(b == 1) nu++; }
            }
        } else if(op == 1 || op == 4) q += len;
        else if(op == 2 || op == 3) pos += len;
    }
done:
    if(nm + nu == 0) return 1.0f;
    return nu / ((float)(nm + nu));
}

__global__ __launch_bounds__(PB) void k_rec_scan(const PrepParams P) {
    __shared__ uint32_t wcnt[PB / 64];
    const int i = blockIdx.x * PB + threadIdx.x;
    int adm = 0;
    if(i < P.n_rec) {
        PrepRec R; memset(&R, 0, sizeof(R));
        const uint64_t o = P.rec_off[i];
        bool ok = o + 4 + 32 <= P.raw_bytes;
        uint32_t bs = 0;
        const uint8_t *r = P.raw + o + 4;
        if(ok) { bs = ld32(P.raw + o); ok = bs >= 32 && o + 4 + (uint64_t)bs <= P.raw_bytes; }
        if(ok) {
            const int32_t tid = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4);
            const uint32_t lqn = r[8], mapq = r[9], ncig = ld16(r + 12), flag = ld16(r + 14);
            const int32_t lq = (int32_t)ld32(r + 16), mpos = (int32_t)ld32(r + 24);
            const uint64_t need = 32ull + lqn + 4ull * ncig + (uint64_t)((lq > 0 ? lq : 0) + 1) / 2 + (uint64_t)(lq > 0 ? lq : 0);
            ok = lq >= 0 && need <= bs && lqn >= 1;
            if(ok) {
                const uint8_t *qn = r + 32, *cig = qn + lqn, *seq = cig + 4 * ncig, *qual = seq + (lq + 1) / 2, *aux = qual + lq, *end = r + bs;
                int32_t rlen = 0;
                for(uint32_t k = 0; k < ncig; k++) { const uint32_t c = ld32(cig + 4 * k); const int op = c & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int32_t)(c >> 4); }
                R.pos = pos; R.rend = pos + rlen; R.lq = (uint32_t)lq; R.ncig = (uint16_t)ncig; R.flag = (uint16_t)flag; R.lqname = (uint8_t)lqn;
                R.seq_off = (uint32_t)(seq - P.raw); R.cig_off = (uint32_t)(cig - P.raw); R.qn_off = (uint32_t)(qn - P.raw);
                const md_prep_cfg &c = P.cfg;
                if(c.perread) {          // perRead.c:178-183: alignments that start inside the chunk; flag masks and MAPQ only
                    const uint8_t *nh, *xg;
                    bool keepr = (int64_t)pos >= P.beg && (int64_t)pos < P.end;
                    keepr = keepr && !(c.require_flags && ((uint32_t)c.require_flags & flag) != (uint32_t)c.require_flags);
                    keepr = keepr && !(c.ignore_flags && ((uint32_t)c.ignore_flags & flag) != 0) && (int)mapq >= c.min_mapq;
                    if(keepr) { scan_aux(aux, end, nh, xg); R.strand = (uint8_t)strand_of(flag, xg); adm = 1; }
                    R.adm = (uint8_t)adm; P.rec[i] = R;
                    goto counted;
                }
                // filter_func, common.c:416-444 (the region query behind it: pos < end, bam_endpos > beg)
                bool keep = tid == P.tid && !(flag & 0x4) && (int64_t)pos < P.end && (int64_t)pos + (rlen > 0 ? rlen : 1) > P.beg;
                keep = keep && (int)mapq >= c.min_mapq && !(flag & (uint32_t)c.ignore_flags);
                keep = keep && !(c.require_flags && (flag & (uint32_t)c.require_flags) != (uint32_t)c.require_flags);
                keep = keep && !(!c.keep_dupes && (flag & 0x400));
                int strand = 0;
                if(keep) {
                    const uint8_t *nh, *xg;
                    scan_aux(aux, end, nh, xg);
                    if(!c.ignore_nh && nh && (int)aux_int(nh) > 1) keep = false;
                    strand = strand_of(flag, xg);
                }
                if(keep && c.map_on) {
                    int64_t s1, s2;
                    if((flag & 0x40) || ((flag & 0x10) && (flag & 0x80))) { s1 = pos; s2 = mpos; } else { s2 = pos; s1 = mpos; }
                    if(!map_window_passes(P, s1, lq) && !map_window_passes(P, s2, lq)) keep = false;
                }
                if(keep && !c.keep_singleton && (flag & 0x9) == 0x9) keep = false;
                if(keep && !c.keep_discordant && (flag & 0x3) == 0x1) keep = false;
                if(keep && P.bed_on && !bed_touches(P, pos, (int64_t)pos + (rlen > 0 ? rlen : 1))) keep = false;
                if(keep && c.min_conv_eff > 0.0f) {
                    int e = 0;
                    if(conv_efficiency(P, cig, (int)ncig, pos, seq, qual, lq, strand, &e) < c.min_conv_eff) keep = false;
                    if(e) atomicExch(&P.cnt->strand0, 1u);
                }
                R.strand = (uint8_t)strand;
                if(keep) {
                    adm = 1;
                    if(c.no_pairing) atomicMax(&P.cnt->max_lq, (uint32_t)lq);          // mbias: rows of the histogram
                    uint64_t h = 0xcbf29ce484222325ULL;
                    for(uint32_t k = 0; k + 1 < lqn && qn[k]; k++) h = (h ^ qn[k]) * 0x100000001b3ULL;
                    P.hash[i] = h ? h : 1;
                }
            }
        }
        if(!ok) atomicExch(&P.cnt->malformed, 1u);
        R.adm = (uint8_t)adm;
        P.rec[i] = R;
    }
counted:
    const unsigned long long m = __ballot(adm);
    if((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if(threadIdx.x == 0) { uint32_t s = 0; for(int w = 0; w < PB / 64; w++) s += wcnt[w]; P.blockcnt[blockIdx.x] = s; }
}

// exclusive scan of per-block counts (one workgroup; n is a few thousand at most); total -> *total.  Also resets the tiles.
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t *cnt, uint32_t *off, int n, uint32_t *total, TileEnt *tiles, int ntiles) {
    __shared__ uint32_t wsum[16]; __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if(tid == 0) carry = 0;
    if(tiles) for(int t = tid; t < ntiles; t += 1024) { tiles[t].first = 0x7fffffff; tiles[t].last = 0; }
    __syncthreads();
    for(int base = 0; base < n; base += 1024) {
        const int i = base + tid; const uint32_t v = i < n ? cnt[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for(int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if(lane >= d) incl += t; }
        if(lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t pre = carry; for(int w = 0; w < wave; w++) pre += wsum[w];
        if(i < n) off[i] = pre + incl - v;
        __syncthreads();
        if(tid == 1023) carry = pre + incl;
        __syncthreads();
    }
    if(tid == 0) *total = carry;
}

__global__ __launch_bounds__(PB) void k_compact(const PrepParams P) {
    __shared__ uint32_t wcnt[PB / 64];
    const int i = blockIdx.x * PB + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PrepRec R; R.adm = 0;
    if(i < P.n_rec) R = P.rec[i];
    const unsigned long long m = __ballot(R.adm);
    if(lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if(!R.adm) return;
    uint32_t a = P.blockoff[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for(int w = 0; w < wave; w++) a += wcnt[w];
    PrepRead D; D.pos = R.pos; D.rend = R.rend; D.seq_off = R.seq_off; D.lq = R.lq; D.cig_off = R.cig_off; D.qn_off = R.qn_off; D.ncig = R.ncig; D.flag = R.flag; D.strand = R.strand; D.lqname = R.lqname; D.pad = 0;
    P.rd[a] = D; P.mate[a] = -1; P.second[a] = 0;
    if(P.aidx) P.aidx[a] = (uint32_t)i;
    if(P.cfg.no_pairing || P.cfg.perread) return;
    // name table: open addressing on the 64-bit hash, members chained through hnext (order is restored by k_pair)
    const uint64_t h = P.hash[i]; uint32_t s = (uint32_t)(h ^ (h >> 32)) & P.hmask;
    for(;;) {
        const unsigned long long old = atomicCAS((unsigned long long *)&P.hkey[s], 0ull, (unsigned long long)h);
        if(old == 0ull || old == (unsigned long long)h) break;
        s = (s + 1) & P.hmask;
    }
    P.hnext[a] = atomicExch(&P.hhead[s], (int32_t)a);
}

__device__ __forceinline__ bool same_name(const PrepParams &P, const PrepRead &x, const PrepRead &y) {
    if(x.lqname != y.lqname) return false;
    const uint8_t *p = P.raw + x.qn_off, *q = P.raw + y.qn_off;
    for(int k = 0; k < x.lqname; k++) { if(p[k] != q[k]) return false; if(!p[k]) break; }
    return true;
}

// overlaps.c:121-147 + the pileup buffer's eviction, per read name (see pair_reads in csrc/host/mdk_pipeline.c for the host
// statement of the same rule): a read enters the buffer iff its end lies beyond the column about to be emitted (the start of
// the previously admitted read); entering, it first drops the name's reads that have been swept out (end < that column) --
// any such drop erases the name's pending entry --, then either becomes pending or is paired with the pending read.
__global__ __launch_bounds__(PB) void k_pair(const PrepParams P, const uint32_t *n_adm_p) {
    const uint32_t s = blockIdx.x * PB + threadIdx.x;
    if(s > P.hmask || P.hkey[s] == 0) return;
    int32_t idx[MAXG]; int k = 0;
    for(int32_t a = P.hhead[s]; a >= 0; a = P.hnext[a]) { if(k == MAXG) { atomicExch(&P.cnt->fallback, 1u); return; } idx[k++] = a; }
    for(int i = 1; i < k; i++) { const int32_t v = idx[i]; int j = i - 1; while(j >= 0 && idx[j] > v) { idx[j + 1] = idx[j]; j--; } idx[j + 1] = v; }
    uint32_t done = 0;                                    // records of other names that share the hash are handled as their own group
    for(int g = 0; g < k; g++) {
        if(done & (1u << g)) continue;
        const PrepRead lead = P.rd[idx[g]];
        int32_t pending = -1; int32_t live[MAXLIVE]; int nlive = 0;
        for(int i = g; i < k; i++) {
            if(done & (1u << i)) continue;
            const int32_t a = idx[i];
            const PrepRead x = P.rd[a];
            if(i != g && !same_name(P, lead, x)) continue;
            done |= 1u << i;
            const bool first = a == 0;
            const int32_t prev_pos = first ? 0 : P.rd[a - 1].pos;
            const bool inserted = first ? (P.tid > 0 || x.rend > 0) : (x.rend > prev_pos);
            if(!inserted) continue;
            bool evicted = false; int w = 0;
            for(int q = 0; q < nlive; q++) { if(!first && live[q] < prev_pos) evicted = true; else live[w++] = live[q]; }
            nlive = w;
            if(evicted) pending = -1;
            if((x.flag & 0x1) && !(x.flag & 12)) {
                if(pending < 0) pending = a;
                else { P.mate[pending] = a; P.mate[a] = pending; P.second[a] = 1; pending = -1; }
            }
            if(nlive == MAXLIVE) { atomicExch(&P.cnt->fallback, 1u); return; }
            live[nlive++] = x.rend;
        }
    }
    (void)n_adm_p;
}

// gapless runs of a CIGAR, one at a time (calculate_positions, overlaps.c:27-52)
struct RunIt {
    const uint8_t *cig; int n, k; int32_t x, y, lq;
    int32_t rx, ry, rl; bool valid;
    __device__ void init(const uint8_t *c, int ncig, int32_t pos, int32_t lq_) { cig = c; n = ncig; k = 0; x = pos; y = 0; lq = lq_; valid = false; next(); }
    __device__ void next() {
        valid = false;
        while(k < n) {
            const uint32_t c = ld32(cig + 4 * k); k++;
            const int op = c & 15; const int32_t len = (int32_t)(c >> 4);
            if(op == 0 || op == 7 || op == 8) {
                int32_t l = len; if(y + l > lq) l = lq - y;          // a CIGAR that consumes more bases than the record stores
                const int32_t sx = x, sy = y;
                x += len; y += len;
                if(l > 0) { rx = sx; ry = sy; rl = l; valid = true; return; }
            } else if(op == 1 || op == 4) y += len;
            else if(op == 2 || op == 3) x += len;
        }
    }
};

// lo/hi: reference extent of the pieces written (for the tile runs); untouched when nothing is emitted
template <bool WRITE>
__device__ __forceinline__ uint32_t read_segments(const PrepParams &P, uint32_t a, md_seg *out, uint32_t base, int64_t &lo, int64_t &hi) {
    const PrepRead r = P.rd[a];
    const int32_t mi = P.mate[a];
    PrepRead m; bool paired = false;
    if(mi >= 0) { m = P.rd[mi]; paired = (((int)r.strand - (int)m.strand) & 1) == 0; }       // overlaps.c:63-65
    RunIt own, oth;
    own.init(P.raw + r.cig_off, r.ncig, r.pos, (int32_t)r.lq);
    if(paired) oth.init(P.raw + m.cig_off, m.ncig, m.pos, (int32_t)m.lq); else oth.valid = false;
    const uint8_t sf = (uint8_t)((r.strand & 7) | ((r.flag & 0x80) ? MDK_SF_READ2 : 0) | (P.second[a] ? MDK_SF_SECOND : 0));
    const uint8_t msf = paired ? (uint8_t)((m.strand & 7) | ((m.flag & 0x80) ? MDK_SF_READ2 : 0)) : 0;
    uint32_t n = 0;
    for(; own.valid; own.next()) {
        int32_t cur = own.rx; const int32_t stop = own.rx + own.rl;
        while(cur < stop) {
            int32_t pe = stop; bool covered = false;
            while(oth.valid && oth.rx + oth.rl <= cur) oth.next();
            if(oth.valid) { if(oth.rx <= cur) { covered = true; if(oth.rx + oth.rl < pe) pe = oth.rx + oth.rl; } else if(oth.rx < pe) pe = oth.rx; }
            if(pe - cur > 65535) pe = cur + 65535;
            if((int64_t)pe > P.beg && (int64_t)cur < P.end) {
                if(WRITE) {
                    md_seg g; g.rpos = cur; g.off4 = r.seq_off; g.l_qseq = r.lq; g.q0 = (uint32_t)(own.ry + (cur - own.rx)); g.len = (uint16_t)(pe - cur);
                    g.sf = sf; g.msf = 0; g.m_off4 = 0; g.m_l_qseq = 0; g.m_q0 = 0;
                    if(covered) { g.sf |= MDK_SF_PARTNER; g.msf = msf; g.m_off4 = m.seq_off; g.m_l_qseq = m.lq; g.m_q0 = (uint32_t)(oth.ry + (cur - oth.rx)); }
                    const uint32_t o = base + n;
                    if((int64_t)o < P.cap_seg) {
                        out[o] = g;
                        if(cur < lo) lo = cur;
                        if(pe > hi) hi = pe;
                    }
                }
                n++;
            }
            cur = pe;
        }
    }
    return n;
}

__global__ __launch_bounds__(PB) void k_seg_count(const PrepParams P) {
    __shared__ uint32_t wsum[PB / 64]; __shared__ unsigned long long wbytes[PB / 64];
    const uint32_t a = blockIdx.x * PB + threadIdx.x, n_adm = P.cnt->n_adm;
    uint32_t n = 0; unsigned long long bytes = 0;
    if(a < n_adm) {
        int64_t lo = 0, hi = 0;
        n = read_segments<false>(P, a, nullptr, 0, lo, hi);
        P.segcnt[a] = n;
        const PrepRead r = P.rd[a];
        bytes = 16ull + 4ull * r.ncig + ((unsigned long long)r.lq + 1) / 2 + r.lq;        // SURVEY.md 8d, per admitted read
    }
    uint32_t s = n; unsigned long long b = bytes;
#pragma unroll
    for(int d = 32; d; d >>= 1) { s += __shfl_xor(s, d); b += __shfl_xor(b, d); }
    if((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wbytes[threadIdx.x >> 6] = b; }
    __syncthreads();
    if(threadIdx.x == 0) {
        uint32_t t = 0; unsigned long long tb = 0; for(int w = 0; w < PB / 64; w++) { t += wsum[w]; tb += wbytes[w]; }
        P.segblk[blockIdx.x] = t;
        if(tb) atomicAdd((unsigned long long *)&P.cnt->algo_bytes, tb);
    }
}

__global__ __launch_bounds__(PB) void k_seg_write(const PrepParams P) {
    __shared__ uint32_t wsum[PB / 64];
    const uint32_t a = blockIdx.x * PB + threadIdx.x, n_adm = P.cnt->n_adm; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = a < n_adm ? P.segcnt[a] : 0;
    uint32_t incl = n;
#pragma unroll
    for(int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if(lane >= d) incl += t; }
    if(lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = P.segblkoff[blockIdx.x] + incl - n;
    for(int w = 0; w < wave; w++) base += wsum[w];
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    if(n) (void)read_segments<true>(P, a, P.seg, base, lo, hi);
    // Tile runs: tile t's run [first, last) must cover every segment touching t.  A lane contributes [base, base + n) to every
    // tile its pieces reach -- a superset, which is all k_pileup needs.  Reads are in coordinate order, so the lanes of a wave
    // touching one tile are (nearly always) consecutive: only the first of them lowers `first`, only the last raises `last`,
    // instead of two contended atomics per segment (275 us -> a few us per 1 Mb chunk).
    int t0 = 0x7fffffff, t1 = -1;
    if(n && hi > lo && (int64_t)base < P.cap_seg) {
        if(lo < P.beg) lo = P.beg;
        if(hi > P.end) hi = P.end;
        t0 = (int)((lo - P.beg) / P.tile); t1 = (int)((hi - 1 - P.beg) / P.tile);
    }
    const int p0 = __shfl_up(t0, 1), p1 = __shfl_up(t1, 1), n0 = __shfl_down(t0, 1), n1 = __shfl_down(t1, 1);
    uint32_t top = base + n; if((int64_t)top > P.cap_seg) top = (uint32_t)P.cap_seg;
    for(int t = t0; t <= t1; t++) {
        if(lane == 0 || t < p0 || t > p1) atomicMin(&P.tiles[t].first, (int)base);
        if(lane == 63 || t < n0 || t > n1) atomicMax(&P.tiles[t].last, (int)top);
    }
}

// perRead over device-selected reads: the walk of k_perread on the records where they lie
__global__ __launch_bounds__(PB) void k_perread_raw(const PrepParams P, const uint8_t *ctxcode, int64_t wend, md_pr_count *out) {
    const uint32_t a = blockIdx.x * PB + threadIdx.x;
    if(a >= P.cnt->n_adm) return;
    const PrepRead r = P.rd[a];
    const uint8_t *seq = P.raw + r.seq_off, *qual = seq + ((r.lq + 1) >> 1), *cg = P.raw + r.cig_off;
    out[a] = perread_walk(seq, qual, r.lq, (int)r.ncig, r.pos, r.strand & 1, ctxcode, P.reflen, wend, P.cfg.min_phred, [cg](int k) { return ld32(cg + 4 * k); });
}

// record offsets of a device-resident range (offsets in the piece it was inflated in) -> offsets in the chunk's concatenation
__global__ __launch_bounds__(256) void k_rebase(uint32_t *dst, const uint32_t *src, uint32_t n, uint32_t add) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if(i < n) dst[i] = src[i] + add;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// the H2D (host ranges) or D2D (ranges of a piece inflated on the device) copies of a chunk's records and of its record table
static int copy_ranges(md_dev *h, Slot *s, const md_raw_batch *b) {
    bool any_dev = false; uint64_t nrec_sum = 0;
    for(int i = 0; i < b->n_ranges; i++) { if(b->range[i].d_rec_off) any_dev = true; nrec_sum += b->range[i].n_records; }
    { uint64_t host_rec = 0; for(int i = 0; i < b->n_ranges; i++) if(!b->range[i].d_rec_off) host_rec += any_dev ? b->range[i].n_records : 0; if((any_dev ? host_rec : (uint64_t)b->n_records) && !b->rec_off) return fail(MDK_ERR_ARG, "md_dev_upload_raw: null record table", hipSuccess); }
    if(any_dev && nrec_sum != (uint64_t)b->n_records) return fail(MDK_ERR_ARG, "md_dev_upload_raw: with a device-resident range every range must carry its record count", hipSuccess);
    uint64_t o = 0; uint32_t idx = 0, hidx = 0;
    for(int i = 0; i < b->n_ranges; i++) {
        const md_raw_range &r = b->range[i];
        if(r.d_rec_off) {
            if(r.bytes) HIPCHK(hipMemcpyAsync(s->d_raw.p + o, r.ptr, (size_t)r.bytes, hipMemcpyDeviceToDevice, s->stream));
            if(r.n_records) hipLaunchKernelGGL(k_rebase, dim3((r.n_records + 255) / 256), dim3(256), 0, s->stream, s->d_recoff.p + idx, r.d_rec_off, r.n_records, (uint32_t)o - r.rec_delta);
        } else {
            if(r.bytes) { host_block_ensure_registered(r.ptr); HIPCHK(hipMemcpyAsync(s->d_raw.p + o, r.ptr, (size_t)r.bytes, hipMemcpyHostToDevice, s->stream)); }
            if(any_dev && r.n_records) HIPCHK(hipMemcpyAsync(s->d_recoff.p + idx, b->rec_off + hidx, sizeof(uint32_t) * (size_t)r.n_records, hipMemcpyHostToDevice, s->stream));
            hidx += r.n_records;
        }
        idx += r.n_records; o += r.bytes;
    }
    if(!any_dev && b->n_records) { host_block_ensure_registered(b->rec_off); HIPCHK(hipMemcpyAsync(s->d_recoff.p, b->rec_off, sizeof(uint32_t) * (size_t)b->n_records, hipMemcpyHostToDevice, s->stream)); }
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int md_dev_read_raw(md_dev *h, int slot, uint8_t *bytes, uint64_t *n_bytes, uint32_t *rec_off, uint32_t *n_records) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->raw_layout || !bytes || !n_bytes || !rec_off || !n_records) return fail(MDK_ERR_ARG, "md_dev_read_raw: needs a slot uploaded with md_dev_upload_raw", hipSuccess);
    if(*n_bytes < s->raw_bytes || *n_records < (uint32_t)s->pr_nrec) { *n_bytes = s->raw_bytes; *n_records = (uint32_t)s->pr_nrec; return fail(MDK_ERR_ARG, "md_dev_read_raw: buffers too small", hipSuccess); }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if(s->raw_bytes) HIPCHK(hipMemcpy(bytes, s->d_raw.p, (size_t)s->raw_bytes, hipMemcpyDeviceToHost));
    if(s->pr_nrec) HIPCHK(hipMemcpy(rec_off, s->d_recoff.p, sizeof(uint32_t) * (size_t)s->pr_nrec, hipMemcpyDeviceToHost));
    *n_bytes = s->raw_bytes; *n_records = (uint32_t)s->pr_nrec;
    return 0;
}

extern "C" int md_dev_set_prep(md_dev *h, const md_prep_cfg *cfg) {
    if(!h || !cfg) return fail(MDK_ERR_ARG, "md_dev_set_prep", hipSuccess);
    h->prep = *cfg; h->prep_set = true;
    return 0;
}

// 1 bit per base of a contig (bit i of word i/32), for the -M/-B admission windows (common.c:277-335)
extern "C" int md_dev_set_mappability(md_dev *h, int32_t tid, const uint32_t *bits, int64_t n_bases) {
    if(!h || tid < 0 || n_bases < 0 || (n_bases && !bits)) return fail(MDK_ERR_ARG, "md_dev_set_mappability", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    if((size_t)tid >= h->mapbits.size()) { h->mapbits.resize(tid + 1, nullptr); h->maplen.resize(tid + 1, 0); }
    if(h->mapbits[tid]) { (void)hipFree(h->mapbits[tid]); h->mapbits[tid] = nullptr; h->maplen[tid] = 0; }
    const size_t words = (size_t)((n_bases + 31) / 32);
    uint32_t *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, (words + 2) * 4);
    if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc(mappability)", e);
    if(words) { e = hipMemcpy(d, bits, words * 4, hipMemcpyHostToDevice); if(e != hipSuccess) { (void)hipFree(d); return fail(MDK_ERR_HIP, "hipMemcpy(mappability)", e); } }
    h->mapbits[tid] = d; h->maplen[tid] = n_bases;
    return 0;
}

static uint32_t pow2_at_least(size_t n) { uint32_t p = 1024; while(p < n) p <<= 1; return p; }

static int enqueue_prep(md_dev *h, Slot *s) {
    PrepParams P; memset(&P, 0, sizeof(P));
    const int n = s->pr_nrec, nb = (n + PB - 1) / PB;
    P.raw = s->d_raw.p; P.raw_bytes = s->raw_bytes; P.rec_off = s->d_recoff.p; P.n_rec = n;
    P.cfg = h->prep; P.tid = s->tid; P.beg = s->beg; P.end = s->end; P.woff = s->woff; P.wlen = s->wlen;
    P.ref = h->ref[s->tid]; P.reflen = h->reflen[s->tid];
    if(h->prep.map_on && (size_t)s->tid < h->mapbits.size()) { P.mapbits = h->mapbits[s->tid]; P.maplen = h->maplen[s->tid]; }
    P.bed_on = (size_t)s->tid < h->d_runs.size() && h->has_runs[s->tid]; if(P.bed_on) { P.runs = h->d_runs[s->tid]; P.nruns = h->n_runs[s->tid]; }
    P.rec = s->d_prec.p; P.hash = s->d_hash.p; P.blockcnt = s->d_blk.p; P.blockoff = s->d_blk.p + nb; P.nblocks = nb;
    P.rd = s->d_prd.p; P.mate = s->d_mate.p; P.second = s->d_second.p;
    P.hkey = s->d_hkey.p; P.hhead = s->d_hhead.p; P.hnext = s->d_hnext.p; P.hmask = s->hmask;
    P.segcnt = s->d_segcnt.p; P.segblk = s->d_blk.p + 2 * nb; P.segblkoff = s->d_blk.p + 3 * nb;
    P.seg = s->d_seg_in.p; P.cap_seg = (int64_t)s->d_seg_in.cap;
    P.tiles = s->d_tiles.p; P.ntiles = s->ntiles; P.tile = s->tile;
    P.cnt = s->d_pcnt.p;
    hipStream_t st = s->stream;
    HIPCHK(hipMemsetAsync(s->d_pcnt.p, 0, sizeof(PrepCounters), st));
    if(!h->prep.no_pairing) {
        HIPCHK(hipMemsetAsync(s->d_hkey.p, 0, sizeof(uint64_t) * ((size_t)s->hmask + 1), st));
        HIPCHK(hipMemsetAsync(s->d_hhead.p, 0xff, sizeof(int32_t) * ((size_t)s->hmask + 1), st));
    }
    if(n > 0) hipLaunchKernelGGL(k_rec_scan, dim3(nb), dim3(PB), 0, st, P);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, (const uint32_t *)P.blockcnt, P.blockoff, nb, &s->d_pcnt.p->n_adm, s->d_tiles.p, s->ntiles);
    if(n > 0) {
        hipLaunchKernelGGL(k_compact, dim3(nb), dim3(PB), 0, st, P);
        if(!h->prep.no_pairing) hipLaunchKernelGGL(k_pair, dim3((s->hmask + PB) / PB), dim3(PB), 0, st, P, (const uint32_t *)&s->d_pcnt.p->n_adm);
        hipLaunchKernelGGL(k_seg_count, dim3(nb), dim3(PB), 0, st, P);
    } else HIPCHK(hipMemsetAsync(s->d_blk.p, 0, sizeof(uint32_t) * 4 * (size_t)(nb > 0 ? nb : 1), st));
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, (const uint32_t *)P.segblk, P.segblkoff, nb, &s->d_pcnt.p->n_segs, (TileEnt *)nullptr, 0);
    if(n > 0) hipLaunchKernelGGL(k_seg_write, dim3(nb), dim3(PB), 0, st, P);
    HIPCHK(hipGetLastError());
    return 0;
}

// H2D of the chunk's record bytes and record table, then the preparation kernels: the slot ends up "uploaded", with its
// segments and tile runs in device memory, exactly as after md_dev_upload of a host-built batch.
extern "C" int md_dev_upload_raw(md_dev *h, int slot, const md_raw_batch *b) {
    Slot *s = get_slot(h, slot);
    if(!s || !b || b->n_records < 0 || b->n_ranges < 0 || b->end < b->beg) return fail(MDK_ERR_ARG, "md_dev_upload_raw", hipSuccess);
    if(!h->prep_set) return fail(MDK_ERR_ARG, "md_dev_upload_raw: md_dev_set_prep was not called", hipSuccess);
    if(b->n_records && !b->range) return fail(MDK_ERR_ARG, "md_dev_upload_raw: null array", hipSuccess);
    if(b->tid < 0 || (size_t)b->tid >= h->ref.size() || !h->ref[b->tid]) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "reference for tid %d not uploaded", b->tid); return MDK_ERR_NOREF; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if(s->run && s->run != s->stream) HIPCHK(hipStreamSynchronize(s->run));
    s->fresh = true;
    uint64_t total = 0;
    for(int i = 0; i < b->n_ranges; i++) total += b->range[i].bytes;
    if(total >= (1ull << 32) - 64) return fail(MDK_ERR_ARG, "md_dev_upload_raw: more than 4 GiB of records in one chunk", hipSuccess);
    const int64_t span = b->end - b->beg; const int TILE = h->tile;
    const int ntiles = (int)((span + TILE - 1) / TILE), n = b->n_records, nb = (n + PB - 1) / PB;
    s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->woff = b->woff; s->wlen = b->wlen; s->uploaded = false; s->launched = false;
    s->tile = TILE; s->ntiles = ntiles; s->lds_bytes = TILE * ((h->variant ? 16 : 8) + 4);
    s->pr_nrec = n; s->raw_bytes = total; s->raw_layout = true; s->n_segs = -1; s->n_reads = -1; s->read_bytes = 0;
    const size_t nn = (size_t)n + 1, nt = (size_t)(ntiles > 0 ? ntiles : 1);
    const size_t segcap = std::max<size_t>(s->d_seg_in.cap, nn * 2 + 4096);
    s->hmask = pow2_at_least(nn * 2) - 1;
    if(s->d_raw.need((size_t)total + 64) || s->d_recoff.need(nn) || s->d_prec.need(nn) || s->d_hash.need(nn) || s->d_blk.need(4 * (size_t)(nb + 1)) ||
       s->d_prd.need(nn) || s->d_mate.need(nn) || s->d_second.need(nn) || s->d_segcnt.need(nn) || s->d_hnext.need(nn) ||
       s->d_hkey.need((size_t)s->hmask + 1) || s->d_hhead.need((size_t)s->hmask + 1) || s->d_seg_in.need(segcap) || s->d_tiles.need(nt) || s->d_seg.need(nt)) return MDK_ERR_NOMEM;
    if(!s->b_site) {
        if(s->d_site.need((size_t)span + 16)) return MDK_ERR_NOMEM;
        if(h->variant && s->d_var.need((size_t)span + 16)) return MDK_ERR_NOMEM;
    }
    { int rcc = copy_ranges(h, s, b); if(rcc) return rcc; }
    int rc = enqueue_prep(h, s); if(rc) return rc;
    s->uploaded = true;
    return 0;
}

extern "C" int md_dev_submit_raw(md_dev *h, int slot, const md_raw_batch *b) {
    int rc = md_dev_upload_raw(h, slot, b);
    if(rc) return rc;
    return md_dev_launch(h, slot);
}

extern "C" int md_dev_perread_submit_raw(md_dev *h, int slot, const md_raw_batch *b) {
    Slot *s = get_slot(h, slot);
    if(!s || !b || b->n_records < 0 || b->n_ranges < 0 || b->end < b->beg) return fail(MDK_ERR_ARG, "md_dev_perread_submit_raw", hipSuccess);
    if(!h->prep_set || !h->prep.perread) return fail(MDK_ERR_ARG, "md_dev_perread_submit_raw: md_dev_set_prep with perread first", hipSuccess);
    if(b->n_records && !b->range) return fail(MDK_ERR_ARG, "md_dev_perread_submit_raw: null array", hipSuccess);
    if(b->tid < 0 || (size_t)b->tid >= h->ref.size() || !h->ref[b->tid]) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "reference for tid %d not uploaded", b->tid); return MDK_ERR_NOREF; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->pr_n = -1; s->uploaded = false; s->launched = false;
    uint64_t total = 0;
    for(int i = 0; i < b->n_ranges; i++) total += b->range[i].bytes;
    if(total >= (1ull << 32) - 64) return fail(MDK_ERR_ARG, "md_dev_perread_submit_raw: more than 4 GiB of records in one chunk", hipSuccess);
    const int n = b->n_records, nb = (n + PB - 1) / PB; const size_t nn = (size_t)n + 1;
    s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->pr_nrec = n; s->raw_bytes = total; s->raw_layout = true;
    if(s->d_raw.need((size_t)total + 64) || s->d_recoff.need(nn) || s->d_prec.need(nn) || s->d_hash.need(nn) || s->d_blk.need(4 * (size_t)(nb + 1)) || s->d_prd.need(nn) ||
       s->d_mate.need(nn) || s->d_second.need(nn) || s->d_aidx.need(nn) || s->h_aidx.need(nn) || s->d_prc.need(nn) || s->h_prc.need(nn)) return MDK_ERR_NOMEM;
    { int rcc = copy_ranges(h, s, b); if(rcc) return rcc; }
    PrepParams P; memset(&P, 0, sizeof(P));
    P.raw = s->d_raw.p; P.raw_bytes = total; P.rec_off = s->d_recoff.p; P.n_rec = n; P.cfg = h->prep; P.tid = b->tid; P.beg = b->beg; P.end = b->end;
    P.ref = h->ref[b->tid]; P.reflen = h->reflen[b->tid];
    P.rec = s->d_prec.p; P.hash = s->d_hash.p; P.blockcnt = s->d_blk.p; P.blockoff = s->d_blk.p + nb; P.nblocks = nb;
    P.rd = s->d_prd.p; P.mate = s->d_mate.p; P.second = s->d_second.p; P.aidx = s->d_aidx.p; P.cnt = s->d_pcnt.p;
    HIPCHK(hipMemsetAsync(s->d_pcnt.p, 0, sizeof(PrepCounters), s->stream));
    if(n > 0) hipLaunchKernelGGL(k_rec_scan, dim3(nb), dim3(PB), 0, s->stream, P);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s->stream, (const uint32_t *)P.blockcnt, P.blockoff, nb, &s->d_pcnt.p->n_adm, (TileEnt *)nullptr, 0);
    if(n > 0) {
        hipLaunchKernelGGL(k_compact, dim3(nb), dim3(PB), 0, s->stream, P);
        int64_t wend = b->end + 10000; if(wend > P.reflen - 1) wend = P.reflen - 1;
        hipLaunchKernelGGL(k_perread_raw, dim3(nb), dim3(PB), 0, s->stream, P, (const uint8_t *)h->refcode[b->tid], wend, s->d_prc.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(s->h_aidx.p, s->d_aidx.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(s->h_prc.p, s->d_prc.p, sizeof(md_pr_count) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(hipMemcpyAsync(s->h_st.p, h->d_status.p + s->index, sizeof(SlotStatus), hipMemcpyDeviceToHost, s->stream));
    s->pr_n = n;
    return 0;
}

extern "C" int md_dev_perread_download_raw(md_dev *h, int slot, const uint32_t **kept, const md_pr_count **counts, int64_t *n) {
    Slot *s = get_slot(h, slot);
    if(!s || !kept || !counts || !n || s->pr_n < 0) return fail(MDK_ERR_ARG, "md_dev_perread_download_raw: nothing submitted on this slot", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if(s->h_st.p->pc.malformed) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "malformed BAM record in the chunk"); return MDK_ERR_ARG; }
    *kept = s->h_aidx.p; *counts = s->h_prc.p; *n = (int64_t)s->h_st.p->pc.n_adm;
    return 0;
}

// after the slot's stream has drained: what the preparation found.  MDK_ERR_PREP_REDO: the segment array was too small and
// has been enlarged -- the caller (finish_count) runs preparation + pileup again on the resident records.
MDK_HIDDEN int prep_outcome(md_dev *h, Slot *s) {
    const PrepCounters &c = s->h_st.p->pc;
    if(c.malformed) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "malformed BAM record in the chunk"); return MDK_ERR_ARG; }
    if(c.strand0) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "Can't determine the strand of a read!"); return MDK_ERR_STRAND0; }
    if(c.fallback) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "a read name with more than %d records or %d live reads: this chunk needs the host preparation", MAXG, MAXLIVE); return MDK_ERR_PREP_HOST; }
    s->n_reads = (int)c.n_adm; s->n_segs = (int)c.n_segs; s->read_bytes = c.algo_bytes;
    if((size_t)c.n_segs > s->d_seg_in.cap) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if(s->run && s->run != s->stream) HIPCHK(hipStreamSynchronize(s->run));
        if(s->d_seg_in.need((size_t)c.n_segs + 64)) return MDK_ERR_NOMEM;
        int rc = enqueue_prep(h, s); if(rc) return rc;
        s->fresh = true;
        return MDK_ERR_PREP_REDO;
    }
    return 0;
}

// the preparation kernels of an uploaded raw slot re-run `iters` times on the resident records, timed with HIP events on the
// slot's stream (bench.py: what the device spends per chunk before the pileup)
extern "C" int md_dev_bench_prep(md_dev *h, int slot, int warmup, int iters, float *ms_per_chunk) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded || !s->raw_layout || iters < 1 || !ms_per_chunk) return fail(MDK_ERR_ARG, "md_dev_bench_prep: needs a slot uploaded with md_dev_upload_raw", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    for(int i = 0; i < warmup; i++) { int rc = enqueue_prep(h, s); if(rc) return rc; }
    HIPCHK(hipEventRecord(s->k0, s->stream));
    for(int i = 0; i < iters; i++) { int rc = enqueue_prep(h, s); if(rc) return rc; }
    HIPCHK(hipEventRecord(s->k1, s->stream));
    HIPCHK(hipEventSynchronize(s->k1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, s->k0, s->k1));
    s->fresh = true;
    *ms_per_chunk = ms / (float)iters;
    return 0;
}

// test hook: the segments the preparation built (device order), and the admitted reads behind them
extern "C" int md_dev_debug_segments(md_dev *h, int slot, md_seg *out, int64_t cap, int64_t *n_segs, int64_t *n_reads) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded || !n_segs) return fail(MDK_ERR_ARG, "md_dev_debug_segments", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if(s->raw_layout) {
        HIPCHK(hipMemcpy(&s->h_st.p->pc, s->d_pcnt.p, sizeof(PrepCounters), hipMemcpyDeviceToHost));
        int rc = prep_outcome(h, s);
        if(rc == MDK_ERR_PREP_REDO) { HIPCHK(hipStreamSynchronize(s->stream)); HIPCHK(hipMemcpy(&s->h_st.p->pc, s->d_pcnt.p, sizeof(PrepCounters), hipMemcpyDeviceToHost)); rc = prep_outcome(h, s); }
        if(rc) return rc;
    }
    *n_segs = s->n_segs; if(n_reads) *n_reads = s->n_reads;
    if(out && s->n_segs > 0) { if(cap < s->n_segs) return fail(MDK_ERR_ARG, "md_dev_debug_segments: buffer too small", hipSuccess); HIPCHK(hipMemcpy(out, s->d_seg_in.p, sizeof(md_seg) * (size_t)s->n_segs, hipMemcpyDeviceToHost)); }
    return 0;
}

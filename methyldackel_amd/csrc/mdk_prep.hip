// mdk_prep.hip -- chunk preparation on the device: from the inflated BAM records of a chunk, as they lie in the file, to
// the segment array k_pileup consumes.  What the reference does for this inside htslib's iterator and pileup buffer, in TWO kernels
// per launch, each over up to 8 chunks at once (a 1 Mb chunk is ~780 workgroups: alone it leaves the machine half empty and every
// launch boundary is paid per chunk):
//
//   k_prep_scan  one lane per BAM record; a wavefront stages the stretch of the record stream that holds its 64 records in LDS with
//                coalesced loads and every lane takes ITS record apart from there: reference length (bam_cigar2rlen), the aux walk for
//                NH and XG (bam_aux_get), getStrand (common.c:84-116) and filter_func's admission tests in its order (common.c:416-444:
//                unmapped, MAPQ, -F, -R, duplicates, NH, mappability windows, singleton, discordant, BED span, conversion efficiency).
//                What later steps need of a record travels in a 64-byte PrepRead AT THE RECORD'S OWN INDEX (admitted or not, a flag
//                says which), so a workgroup depends on no other: which records it takes follows from its place in the launch, nothing
//                is compacted, nobody waits.  The start of the read admitted just before a read -- what htslib's pileup buffer evicts
//                against -- is found with a ballot inside the wavefront and an exchange inside the workgroup.  Each admitted read goes
//                into a name-keyed hash table (what khash does in custom_overlap_constructor) whose compare-and-swap returns the name's
//                previous read: the two are linked both ways (hnext, hfwd), which is all k_prep_segs follows -- it never sees the table.
//                (perRead's selection keeps the order-preserving compaction of rounds 2-4 in a kernel of its own, k_prep_scan_ordered.)
//   k_prep_segs  one lane per record: an admitted read and the other read its links lead to (nearly always its mate, a few dozen
//                records away) go through the constructor/destructor state machine of overlaps.c:121-147 *including* htslib's buffer
//                eviction (a read leaves the pileup buffer once a later read starts beyond its end, and its destructor erases the
//                name) in closed form; a name with more reads walks its chain and runs the machine step by step.  Then CIGAR ->
//                gapless runs (calculate_positions, overlaps.c:27-52; htslib resolve_cigar2), cut where the partner's runs begin and
//                end, clipped to the chunk: counted, placed IN FILE ORDER (a workgroup draws a ticket, publishes its count and adds up
//                the counts of the tickets before it -- they all belong to workgroups that are already running, so nobody waits for a
//                workgroup that has not started), written; every segment also widens the [first,last) run of the tiles it touches.
//
// Nothing is copied: segments address sequence and qualities inside the uploaded record bytes (MDK layout 1, see
// KParams::unit/packed in mdk_hip.hip).  A name with more records than a lane keeps in registers, or more live reads than
// its window holds, sets a flag and the host prepares that chunk the slow way (mdk_pipeline.c) -- same segments either way.
#include <algorithm>
#include <atomic>
#include <mutex>
#include "mdk_hip_internal.hpp"
#include "mdk_pair_rule.h"          // the pending/pairing machine of the overlap callbacks, shared with its host test

#ifndef PB
#define PB 256                       // threads per block of the per-record kernels
#endif
#define MAXG 16                      // records of one name a lane sorts in registers
#define CNT_READY 0x80000000u        // a workgroup's published count: this bit | count

__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }       // the hardware reads at any alignment
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint4 ld128(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
// A lane walks ITS record, so every load of a wave touches 64 different cache lines and costs what 64 lines cost, whatever its width:
// the kernels below fetch 16 (header, name, CIGAR) or 8 (aux fields) bytes at a time and take them apart in registers.  The record
// buffer is allocated 64 bytes longer than the records (md_dev_upload_raw), so a wide load that starts inside never leaves it.

struct PrepParams {
    const uint8_t *raw; uint64_t raw_bytes; const uint32_t *rec_off; int n_rec;
    md_prep_cfg cfg; int32_t tid; int64_t beg, end, woff, wlen;
    const char *ref; int64_t reflen;                 // contig letters (conversion efficiency)
    const uint32_t *mapbits; int64_t maplen;         // 1 bit per base, or NULL
    const md_region *runs; int64_t nruns; int bed_on;
    PrepRead *rd; uint32_t *aidx;    // per admitted read: the read, (perRead) its index among the candidate records
    unsigned long long *hent; int32_t *hnext, *hfwd; uint32_t hmask;      // name table: (high half of the name's hash) << 32 | (index + 1 of the name's latest read); 0 = empty.  hnext[i]: the read of i's name that was in the table before i; hfwd[i]: the one that came after
    uint32_t *cntA, *cntS, *ticket; int nblocks;     // per workgroup: published counts of admitted reads (perRead) / segments; two ticket counters
    md_seg *seg; int64_t cap_seg;
    TileEnt *tiles; int ntiles, tile;
    PrepCounters *cnt;
    uint8_t *zero; uint64_t zero_bytes;              // what a launch starts from zeroed: name table, counts, tickets
};
struct PrepMulti { int n; int bstart[MAXM + 1]; PrepParams P[MAXM]; };

// Which chunk of the launch a workgroup works for.  Workgroups are dealt to the 8 XCDs round robin and every XCD has its own 4 MB L2:
// workgroup b serves chunk (b mod 8) mod n, so that with eight chunks per launch an XCD sees one chunk only and that chunk's name table
// (2 MB) and ticket counters stay in its L2 instead of every L2 thrashing over all eight.  WHICH records of the chunk a workgroup
// takes is decided by the ticket it draws, not by its index; the launch holds at least as many workgroups per chunk as the chunk has
// tickets (enqueue_prep_group), and a workgroup that draws a ticket beyond them leaves.
#ifndef PREP_WIN_AUX
#define PREP_WIN_AUX 2                // cache policy of the window's loads (2 = nt: the record stream is read once and should not push the name table out of the caches)
#endif
#ifndef PREP_EXP_NOCAS
#define PREP_EXP_NOCAS 0
#endif
#ifndef PREP_EXP_SEGS
#define PREP_EXP_SEGS 0               // TIMING EXPERIMENTS ONLY (wrong results), bits: 1 no tile-run atomics, 2 no write pass, 4 no wait for the earlier tickets, 8 the ticket from blockIdx, 16 no byte tally
#endif
#ifndef PREP_EXP_SCAN
#define PREP_EXP_SCAN 0               // TIMING EXPERIMENTS ONLY (wrong results), bits: 1 the PrepReads are not written out
#endif
#ifndef PREP_LOCAL
#define PREP_LOCAL 1                  // k_prep_scan pairs the reads of a workgroup among themselves in LDS before the chunk's table is asked (0: every read asks the table)
#endif
#ifndef PREP_STAGE
#define PREP_STAGE 2                  // how k_prep_scan gets at the records: 2 = of every record the pieces the scan looks at, gathered in LDS (GatherView: 25 KB per workgroup); 1 = the wavefront's whole stretch of the record stream in LDS (rounds 4-5: 76 KB per workgroup, two workgroups per CU); 0 = every lane reads its record from HBM
#endif
__device__ __forceinline__ int chunk_of_block(const PrepMulti &M) { return (int)((blockIdx.x & 7u) % (unsigned)M.n); }
// What the workgroups of a chunk tell each other -- tickets, published counts, the name table's compare-and-swaps -- goes through agent-scope
// atomics.  (Round 4 tried workgroup-scope atomics with a chunk pinned to the XCD s_getreg(XCC_ID) names and workgroups that keep drawing
// tickets: the ISA is the same but for one cache bit, the gain was 4 %, and with three groups in flight on three streams the launches
// dead-locked on the 16-contig 8.7 GB input -- gpurun_out r04l; removed.)
__device__ __forceinline__ uint32_t sync_add(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ uint32_t sync_peek(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sync_set(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef PREP_CAS_SCOPE
#define PREP_CAS_SCOPE 0              // EXPERIMENT: 1 = the name table's compare-and-swaps at workgroup scope (resolved in the XCD's own L2; right only while every workgroup of a chunk runs on one XCD)
#endif
__device__ __forceinline__ unsigned long long sync_cas(unsigned long long *p, unsigned long long expected, unsigned long long desired) {
#if PREP_CAS_SCOPE
    (void)__hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return expected;
#else
    return atomicCAS(p, expected, desired);
#endif
}

// ---- a record through a view of its bytes; the aux area ----
struct AuxHit { bool nh, xg; int64_t nh_val; uint8_t xg1; };
__device__ __forceinline__ int first_zero_byte(uint64_t x, int n) {        // index of the first zero among the low n (<= 8) bytes, or n
    if(n < 8) x |= ~0ull << (8 * n);
    const uint64_t t = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
    return t ? (int)(__ffsll((unsigned long long)t) - 1) >> 3 : n;
}
// Where a lane reads ITS record from.  GlobalView: the record bytes in HBM, loads of any alignment (every load of a wavefront then touches 64
// different lines).  LdsView: the wavefront has staged the stretch of the record stream that holds its 64 records in LDS with coalesced
// 16-byte loads (k_prep_scan), and the fields are picked out of it: aligned dword reads + v_alignbyte, the record starts at any byte.
#ifndef PREP_GLOBAL_NT
#define PREP_GLOBAL_NT 0              // the lanes' own loads of their records non-temporal
#endif
typedef uint32_t __attribute__((aligned(1))) u32_any; typedef uint64_t __attribute__((aligned(1))) u64_any;
typedef uint32_t u32x4_any __attribute__((ext_vector_type(4), aligned(1)));
struct GlobalView {
    const uint8_t *r;                              // the record's block_size word
#if PREP_GLOBAL_NT
    __device__ __forceinline__ uint32_t u32(uint32_t x) const { return __builtin_nontemporal_load((const u32_any *)(r + x)); }
    __device__ __forceinline__ uint64_t u64(uint32_t x) const { return __builtin_nontemporal_load((const u64_any *)(r + x)); }
    __device__ __forceinline__ uint4 u128(uint32_t x) const { const u32x4_any v = __builtin_nontemporal_load((const u32x4_any *)(r + x)); return make_uint4(v.x, v.y, v.z, v.w); }
#else
    __device__ __forceinline__ uint32_t u32(uint32_t x) const { return ld32(r + x); }
    __device__ __forceinline__ uint64_t u64(uint32_t x) const { return ld64(r + x); }
    __device__ __forceinline__ uint4 u128(uint32_t x) const { return ld128(r + x); }
#endif
    __device__ __forceinline__ bool fits(uint32_t, uint32_t, uint32_t) { return true; }
};
struct LdsView {
    const uint32_t *w; uint32_t b;                 // window words; byte offset of the record's block_size word in the window
    __device__ __forceinline__ uint32_t u32(uint32_t x) const { const uint32_t a = b + x, i = a >> 2; return __builtin_amdgcn_alignbyte(w[i + 1], w[i], a & 3u); }
    __device__ __forceinline__ uint64_t u64(uint32_t x) const { const uint32_t a = b + x, i = a >> 2, k = a & 3u; const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2]; return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, k) | (uint64_t)__builtin_amdgcn_alignbyte(w2, w1, k) << 32; }
    __device__ __forceinline__ uint4 u128(uint32_t x) const { const uint32_t a = b + x, i = a >> 2, k = a & 3u; const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2], w3 = w[i + 3], w4 = w[i + 4];
        return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, k), __builtin_amdgcn_alignbyte(w2, w1, k), __builtin_amdgcn_alignbyte(w3, w2, k), __builtin_amdgcn_alignbyte(w4, w3, k)); }
    __device__ __forceinline__ bool fits(uint32_t, uint32_t, uint32_t) { return true; }
};
// GatherView: of every record the wavefront has staged only the stretch that the scan looks at -- GW_NP 16-byte pieces from GW_PRE pieces before
// the record's (16-byte aligned) start: the END of the record before it, i.e. that record's aux fields, then this record's header, name and CIGAR;
// three quarters of a record, its sequence and qualities, are never asked for (stage_gather).  A record's bytes are therefore in two places: its
// head (offsets below `thr`) in its own window, its aux area in the window of the record after it.  fits() says whether both hold what the scan
// will ask for: a record whose head its window does not hold (a long name, a long CIGAR) is read from HBM by its lane altogether, one whose
// aux area is longer than what the next window holds of it has its aux fields walked in HBM (`auxg`; scan_aux reads through u64 and u128).
#ifndef GW_PRE
#define GW_PRE 1
#endif
#ifndef GW_NP
#define GW_NP 6                       // 96 bytes: 16-31 of the record before, at least 65 of the record (36 + a name of up to 24 letters + one operation)
#endif
#ifndef GW_NP_WIDE
#define GW_NP_WIDE 8                  // 128 bytes: k_prep_scan_wide, for libraries with long read names (at least 97 bytes of the record: a name of up to 56 letters)
#endif
#define GW_LDS(NP) (65 * 16 * (NP) + 32)              // 64 records and the one behind them (+ slack for word reads at the end)
__device__ __forceinline__ uint32_t gw_start(uint32_t o) { return o >= 16u * GW_PRE ? (o & ~15u) - 16u * GW_PRE : 0u; }
struct GatherView {
    const uint32_t *w; const uint8_t *g; bool auxg; uint32_t c1, c2, thr, lim1, onext, ws2, o;      // g: the record in HBM; byte x of the record at c1 + x of the wavefront's LDS (x < thr) or at c2 + x; lim1: bytes of the record its own window holds; ws2: where the next window starts in the record stream
    __device__ __forceinline__ uint32_t at(uint32_t x) const { return x + (x < thr ? c1 : c2); }
    __device__ __forceinline__ uint32_t u32(uint32_t x) const { const uint32_t a = at(x), i = a >> 2; return __builtin_amdgcn_alignbyte(w[i + 1], w[i], a & 3u); }
    __device__ __forceinline__ uint64_t u64(uint32_t x) const {
        if(auxg && x >= thr) return ld64(g + x);
        const uint32_t a = at(x), i = a >> 2, k = a & 3u; const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2]; return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, k) | (uint64_t)__builtin_amdgcn_alignbyte(w2, w1, k) << 32; }
    __device__ __forceinline__ uint4 u128(uint32_t x) const {
        if(auxg && x >= thr) return ld128(g + x);
        const uint32_t a = at(x), i = a >> 2, k = a & 3u; const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2], w3 = w[i + 3], w4 = w[i + 4];
        return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, k), __builtin_amdgcn_alignbyte(w2, w1, k), __builtin_amdgcn_alignbyte(w3, w2, k), __builtin_amdgcn_alignbyte(w4, w3, k)); }
    // head: the record's bytes [0, hend) are wanted; aux area [aux, end)
    __device__ __forceinline__ bool fits(uint32_t hend, uint32_t aux, uint32_t end) {
        if(hend > lim1) return false;
        auxg = aux < end && (o + end != onext || o + aux < ws2);
        thr = aux;
        return true;
    }
};
// aux area [s, e) (offsets from the record's block_size word): first NH and first XG, as bam_aux_get finds them; a malformed area ends the walk.
// One 8-byte read per field: tag (2), type (1) and the first five value bytes -- all of a fixed-size value that NH or XG can have, and the first
// letter of a string, which is all getStrand looks at.  Only a string longer than four letters costs further reads.
template <typename V>
__device__ __forceinline__ AuxHit scan_aux(const V &v, uint32_t s, const uint32_t e) {
    AuxHit A; A.nh = false; A.xg = false; A.nh_val = 0; A.xg1 = 0;
    while(e >= s + 3) {
        const uint64_t w = v.u64(s);
        const uint8_t t = (uint8_t)(w >> 16); const uint32_t vo = s + 3; size_t sz;
        const int64_t avail = (int64_t)e - vo;
        if(t == 'A' || t == 'c' || t == 'C') sz = 1;
        else if(t == 's' || t == 'S') sz = 2;
        else if(t == 'i' || t == 'I' || t == 'f') sz = 4;
        else if(t == 'd') sz = 8;
        else if(t == 'Z' || t == 'H') {
            int n = avail < 5 ? (int)avail : 5, k = first_zero_byte(w >> 24, n);
            if(k < n) sz = (size_t)k + 1;
            else {
                uint32_t z = vo + 5; bool found = false;
                if(avail <= 5) return A;
                // a long string (a bisulfite aligner's XM:Z holds a letter per base): 32 letters per round trip, two 16-byte reads asked for together
                // (eight at a time cost a 150-letter string 19 dependent round trips, and the scan of such a library four times its time);
                // what is read past the area's end lies in the next record or the buffer's 64 spare bytes
                while(z < e && !found) {
                    const uint4 b0 = v.u128(z), b1 = v.u128(z + 16);
                    const uint64_t w4[4] = {(uint64_t)b0.x | (uint64_t)b0.y << 32, (uint64_t)b0.z | (uint64_t)b0.w << 32, (uint64_t)b1.x | (uint64_t)b1.y << 32, (uint64_t)b1.z | (uint64_t)b1.w << 32};
#pragma unroll
                    for(int i = 0; i < 4; i++) {
                        const uint32_t zi = z + 8u * (uint32_t)i;
                        if(!found && zi < e) { n = e - zi < 8 ? (int)(e - zi) : 8; k = first_zero_byte(w4[i], n); if(k < n) { z = zi + (uint32_t)k; found = true; } }
                    }
                    if(!found) z += 32;
                }
                if(!found) return A;
                sz = (size_t)(z - vo) + 1;
            }
        } else if(t == 'B') {
            size_t es; if(avail < 5) return A;
            const uint8_t st = (uint8_t)(w >> 24);
            if(st == 'c' || st == 'C') es = 1; else if(st == 's' || st == 'S') es = 2; else if(st == 'i' || st == 'I' || st == 'f') es = 4; else return A;
            sz = 5 + es * (size_t)(uint32_t)(w >> 32);
        } else return A;
        if((size_t)avail < sz) return A;
        const uint32_t tag = (uint32_t)w & 0xffffu;
        if(tag == ('N' | 'H' << 8) && !A.nh) {
            A.nh = true;
            const uint32_t val = (uint32_t)(w >> 24);
            switch(t) {                                      // bam_aux2i; other types read as 0
            case 'c': A.nh_val = (int8_t)val; break; case 'C': A.nh_val = (uint8_t)val; break;
            case 's': A.nh_val = (int16_t)val; break; case 'S': A.nh_val = (uint16_t)val; break;
            case 'i': A.nh_val = (int32_t)val; break; case 'I': A.nh_val = val; break;
            default: A.nh_val = 0;
            }
        } else if(tag == ('X' | 'G' << 8) && !A.xg) { A.xg = true; A.xg1 = (uint8_t)(w >> 24); }
        s = vo + (uint32_t)sz;
    }
    return A;
}
__device__ __forceinline__ int strand_of(uint32_t flag, const AuxHit &A) {        // common.c:84-116
    int conv = 0;
    if(A.xg && (A.xg1 == 'C' || A.xg1 == 'G')) conv = A.xg1;
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

// sum over the lanes of a workgroup (PB threads); every thread gets the result
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *red) {
#pragma unroll
    for(int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    if((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for(int w = 0; w < PB / 64; w++) t += red[w];
    __syncthreads();
    return t;
}
// what the workgroups holding earlier tickets counted, added up (they are running: a ticket is drawn by a workgroup that has started).
// Every thread asks for four of the counts at once -- a thousand tickets in one round trip -- and only then waits for the ones not yet published.
__device__ __forceinline__ uint32_t tickets_before(uint32_t *cnt, uint32_t tk, uint32_t *red) {
    uint32_t part = 0;
    for(uint32_t q0 = threadIdx.x; q0 < tk; q0 += 4 * PB) {
        uint32_t v[4];
#pragma unroll
        for(int u = 0; u < 4; u++) { const uint32_t q = q0 + (uint32_t)u * PB; v[u] = q < tk ? sync_peek(&cnt[q]) : CNT_READY; }
#pragma unroll
        for(int u = 0; u < 4; u++) {
            const uint32_t q = q0 + (uint32_t)u * PB;
            while(!(v[u] & CNT_READY)) { __builtin_amdgcn_s_sleep(1); v[u] = sync_peek(&cnt[q]); }
            part += v[u] & ~CNT_READY;
        }
    }
    return block_sum(part, red);
}

// everything a launch starts from zeroed, for all its chunks (one launch instead of two memsets per chunk)
__global__ __launch_bounds__(PB) void k_prep_zero(const PrepMulti M) {
    // workgroup b works for chunk b mod n with the b / n-th share of it (the grid is n times a chunk's share: the launches over eight
    // chunks have a grid of their own, which is how the profiles tell them from the one-chunk launches)
    const int j = (int)(blockIdx.x % (unsigned)M.n); const uint32_t b = blockIdx.x / (unsigned)M.n, nb = gridDim.x / (unsigned)M.n;
    const PrepParams &P = M.P[j];
    uint4 *z = (uint4 *)P.zero; const uint64_t n16 = P.zero_bytes >> 4;
    for(uint64_t i = (uint64_t)b * PB + threadIdx.x; i < n16; i += (uint64_t)nb * PB) z[i] = make_uint4(0, 0, 0, 0);
    for(int t = (int)(b * PB + threadIdx.x); t < P.ntiles; t += (int)(nb * PB)) { P.tiles[t].first = 0x7fffffff; P.tiles[t].last = 0; }      // the tile runs start empty
    if(P.hfwd) { uint4 *f = (uint4 *)P.hfwd; const int n16f = (P.n_rec + 3) >> 2; for(int i = (int)(b * PB + threadIdx.x); i < n16f; i += (int)(nb * PB)) f[i] = make_uint4(~0u, ~0u, ~0u, ~0u); }      // nobody came after anybody yet
    if(b == 0 && threadIdx.x < sizeof(PrepCounters) / 4) ((uint32_t *)P.cnt)[threadIdx.x] = 0;
}

// One record: fields, reference length, aux walk, strand, filter_func's tests in its order, name hash.  `v` reads the record's bytes (from HBM
// or from the wavefront's staged window), o = the record's place in the chunk's record bytes.  Returns whether the record is admitted; D and h
// are what the rest of the kernel needs of it.  ok = false: malformed.
template <typename V>
__device__ __forceinline__ int scan_record(const PrepParams &P, V v, const uint64_t o, PrepRead &D, uint64_t &h, bool &ok, bool &redo) {
    int adm = 0;
    const uint4 h0 = v.u128(0), h1 = v.u128(16);          // block_size refID pos (l_read_name mapq bin) | (n_cigar flag) l_seq next_refID next_pos
    const uint32_t bs = h0.x;
    const int32_t tid = (int32_t)h0.y, pos = (int32_t)h0.z;
    const uint32_t lqn = h0.w & 255u, mapq = (h0.w >> 8) & 255u, ncig = h1.x & 0xffffu, flag = h1.x >> 16;
    const int32_t lq = (int32_t)h1.y, mpos = (int32_t)h1.w;
    const uint64_t need = 32ull + lqn + 4ull * ncig + (uint64_t)((lq > 0 ? lq : 0) + 1) / 2 + (uint64_t)(lq > 0 ? lq : 0);
    ok = bs >= 32 && o + 4 + (uint64_t)bs <= P.raw_bytes && lq >= 0 && need <= bs && lqn >= 1;
    if(!ok) return 0;
    // offsets from the record's block_size word
    const uint32_t qn = 4 + 32, cig = qn + lqn, seq = cig + 4 * ncig, qual = seq + (uint32_t)(lq + 1) / 2, aux = qual + (uint32_t)lq, end = 4 + bs;
    if(!v.fits(seq, aux, end)) { redo = true; return 0; }       // (a view that holds a part of the record only: not this record's)
    const uint4 cb = v.u128(cig);                         // the first four CIGAR operations
    int32_t rlen = 0;
    for(uint32_t k = 0; k < ncig; k++) { const uint32_t c = k == 0 ? cb.x : k == 1 ? cb.y : k == 2 ? cb.z : k == 3 ? cb.w : v.u32(cig + 4 * k); const int op = c & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int32_t)(c >> 4); }
    D.pos = pos; D.rend = pos + rlen; D.lq = (uint32_t)lq; D.ncig = (uint16_t)ncig; D.flag = (uint16_t)flag;
    D.lqn = (uint8_t)lqn;
    {   // the first three CIGAR operations in 21 bits each (a length below 2^17 and the operation, as BAM packs them) + a flag that they are
        unsigned long long pk = 0; bool okp = true;
#pragma unroll
        for(int k = 0; k < 3; k++) { const uint32_t cc = k == 0 ? cb.x : k == 1 ? cb.y : cb.z; if((uint32_t)k < ncig) { okp = okp && cc < (1u << 21); pk |= (unsigned long long)(cc & 0x1fffffu) << (21 * k); } }
        if(okp) pk |= 1ull << 63;
        D.cig[0] = (uint32_t)pk; D.cig[1] = (uint32_t)(pk >> 32);
    }
    const md_prep_cfg &c = P.cfg;
    if(c.perread) {          // perRead.c:178-183: alignments that start inside the chunk; flag masks and MAPQ only
        bool keepr = (int64_t)pos >= P.beg && (int64_t)pos < P.end;
        keepr = keepr && !(c.require_flags && ((uint32_t)c.require_flags & flag) != (uint32_t)c.require_flags);
        keepr = keepr && !(c.ignore_flags && ((uint32_t)c.ignore_flags & flag) != 0) && (int)mapq >= c.min_mapq;
        if(keepr) { const AuxHit A = scan_aux(v, aux, end); D.strand = (uint8_t)strand_of(flag, A); adm = 1; }
        return adm;
    }
    // filter_func, common.c:416-444 (the region query behind it: pos < end, bam_endpos > beg)
    bool keep = tid == P.tid && !(flag & 0x4) && (int64_t)pos < P.end && (int64_t)pos + (rlen > 0 ? rlen : 1) > P.beg;
    keep = keep && (int)mapq >= c.min_mapq && !(flag & (uint32_t)c.ignore_flags);
    keep = keep && !(c.require_flags && (flag & (uint32_t)c.require_flags) != (uint32_t)c.require_flags);
    keep = keep && !(!c.keep_dupes && (flag & 0x400));
    int strand = 0;
    if(keep) {
        const AuxHit A = scan_aux(v, aux, end);
        if(!c.ignore_nh && A.nh && (int)A.nh_val > 1) keep = false;
        strand = strand_of(flag, A);
    }
    if(keep && c.map_on) {
        int64_t s1, s2;
        if((flag & 0x40) || ((flag & 0x10) && (flag & 0x80))) { s1 = pos; s2 = mpos; } else { s2 = pos; s1 = mpos; }
        if(!map_window_passes(P, s1, lq) && !map_window_passes(P, s2, lq)) keep = false;
    }
    if(keep && !c.keep_singleton && (flag & 0x9) == 0x9) keep = false;
    if(keep && !c.keep_discordant && (flag & 0x3) == 0x1) keep = false;
    if(keep && P.bed_on && !bed_touches(P, pos, (int64_t)pos + (rlen > 0 ? rlen : 1))) keep = false;
    if(keep && c.min_conv_eff > 0.0f) {                   // (a rare option: straight from the record bytes in HBM)
        int e = 0; const uint8_t *r = P.raw + o;
        if(conv_efficiency(P, r + cig, (int)ncig, pos, r + seq, r + qual, lq, strand, &e) < c.min_conv_eff) keep = false;
        if(e) atomicExch(&P.cnt->strand0, 1u);
    }
    D.strand = (uint8_t)strand;
    if(!keep) return 0;
    if(c.no_pairing) return 1;                            // mbias: no names
    // the name as strcmp sees it (its letters up to the first NUL, at most l_read_name - 1 of them), 16 bytes at a time: hashed, and its
    // first block kept in the read
    uint32_t nlen = 0; h = 0x9e3779b97f4a7c15ULL;
    for(int32_t left = (int32_t)lqn - 1, blk = 0; left > 0; left -= 16, blk++) {
        const uint4 nb = v.u128(qn + 16 * blk);
        const int lim = left < 16 ? left : 16;
        int z = first_zero_byte((uint64_t)nb.x | (uint64_t)nb.y << 32, 8);
        if(z == 8) z += first_zero_byte((uint64_t)nb.z | (uint64_t)nb.w << 32, 8);
        const int take = z < lim ? z : lim;
        uint32_t w[4] = {nb.x, nb.y, nb.z, nb.w};
#pragma unroll
        for(int d = 0; d < 4; d++) { const int kept = take - 4 * d; if(kept <= 0) w[d] = 0; else if(kept < 4) w[d] &= (1u << (8 * kept)) - 1u; }
#pragma unroll
        for(int d = 0; d < 4; d++) { h = (h ^ w[d]) * 0xff51afd7ed558ccdULL; h ^= h >> 29; }
        if(blk == 0) { D.name[0] = w[0]; D.name[1] = w[1]; D.name[2] = w[2]; D.name[3] = w[3]; }
        nlen += (uint32_t)take;
        if(take < 16) break;
    }
    D.nlen = (uint8_t)nlen;
    h = (h ^ nlen) * 0xc4ceb9fe1a85ec53ULL; h ^= h >> 32;
    if(!h) h = 1;
    return 1;
}

// bytes of the record stream a wavefront stages in LDS for its 64 records (a 2x150 library: 64 x 283 B = 18.1 KB); a record that does not lie
// inside the window whole is read from HBM by its lane
#ifndef RAWWIN
#define RAWWIN 18944
#endif
#define RAWWIN_LDS (RAWWIN + 32)                      // (+ slack for the word reads of a field that ends at the window's end)
// Where the lane's record starts and where the next one does: two coalesced loads, issued together, before anything depends on them -- the
// window's bounds are then lane 0's start and lane 63's end, and the lane needs no further load before it can take its record apart.
struct RecAt { uint32_t o, onext; };
__device__ __forceinline__ RecAt rec_at(const PrepParams &P, const int i) {
    RecAt A; A.o = 0; A.onext = 0;
    if(i < P.n_rec) { A.o = P.rec_off[i]; A.onext = i + 1 < P.n_rec ? P.rec_off[i + 1] : (uint32_t)P.raw_bytes; }
    return A;
}
// the wavefront's stretch of the record stream -> LDS, 1 KiB per instruction, straight from HBM (global_load_lds: no registers in between)
__device__ __forceinline__ void stage_window(const PrepParams &P, const int i0, const RecAt &A, const int lane, uint8_t *const win, uint32_t &wbase, uint32_t &wlen) {
    wbase = 0; wlen = 0;
#if PREP_STAGE
    const uint32_t first = (uint32_t)__shfl((int)A.o, 0), last = (uint32_t)__shfl((int)A.onext, 63);       // (every lane takes part)
    if(i0 < P.n_rec) {
        const uint64_t b0 = first, b1 = i0 + 64 <= P.n_rec ? (uint64_t)last : P.raw_bytes;
        wbase = (uint32_t)(b0 & ~15ull);
        uint64_t l = b1 > wbase ? ((b1 - wbase) + 15) & ~15ull : 0; if(l > RAWWIN) l = RAWWIN;
        if((uint64_t)wbase + l > ((P.raw_bytes + 15) & ~15ull)) l = (uint64_t)wbase < ((P.raw_bytes + 15) & ~15ull) ? ((P.raw_bytes + 15) & ~15ull) - wbase : 0;       // (the buffer is 64 bytes longer than the records)
        wlen = (uint32_t)l;
        const uint8_t *src = P.raw + wbase + 16 * lane;
        for(uint32_t k = 0; k < wlen; k += 1024) if(k + 16 * lane < wlen) __builtin_amdgcn_global_load_lds((const void *)(src + k), (__attribute__((address_space(3))) void *)(win + k), 16, 0, PREP_WIN_AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_wave_barrier();
#endif
}
// PREP_STAGE 2: of each of the wavefront's 64 records, and of the record behind them, NP pieces (GatherView) -> LDS, window w at w * 16 NP.
// A lane brings piece p = 64 j + lane in round j: the pieces of a window are neighbours in the instruction and fall into one or two lines.
template <int NP>
__device__ __forceinline__ void stage_gather(const PrepParams &P, const int i, const RecAt &A, const int lane, uint8_t *const win) {
    // the start of window `lane`: the lane's record, or -- the lane after the chunk's last record -- the end of the records
    const uint32_t ow = i < P.n_rec ? A.o : i == P.n_rec ? (uint32_t)P.raw_bytes : 0xffffffffu;
    const uint32_t o64 = (uint32_t)__shfl((int)(i + 1 < P.n_rec ? A.onext : i + 1 == P.n_rec ? (uint32_t)P.raw_bytes : 0xffffffffu), 63);
    const uint32_t lim = (uint32_t)((P.raw_bytes + 15) & ~15ull);          // (the buffer is 64 bytes longer than the records)
#pragma unroll
    for(int j = 0; j < (65 * NP + 63) / 64; j++) {
        const uint32_t p = 64u * j + (uint32_t)lane, wdw = p / NP, k = p - wdw * NP;
        const uint32_t os = (uint32_t)__shfl((int)ow, (int)(wdw & 63u));
        const uint32_t oo = wdw < 64u ? os : o64;
        const uint32_t src = gw_start(oo) + 16u * k;
        if(wdw <= 64u && oo != 0xffffffffu && src < lim)
            __builtin_amdgcn_global_load_lds((const void *)(P.raw + src), (__attribute__((address_space(3))) void *)(win + 1024 * j), 16, 0, PREP_WIN_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
template <int NP>
__device__ __forceinline__ int scan_one_gather(const PrepParams &P, const int i, const RecAt &A, const int lane, const uint8_t *const win, PrepRead &D, uint64_t &h, bool &far) {
    constexpr uint32_t GW_BYTES = 16 * NP;
    int adm = 0;
    if(i < P.n_rec) {
        const uint64_t o = A.o;
        bool ok = o + 4 + 32 <= P.raw_bytes, redo = false;
        if(ok) {
            const uint32_t ws1 = gw_start(A.o), ws2 = gw_start(A.onext);
            GatherView v; v.w = (const uint32_t *)win; v.g = P.raw + o; v.auxg = false; v.o = A.o; v.onext = A.onext; v.ws2 = ws2;
            v.c1 = (uint32_t)lane * GW_BYTES + (A.o - ws1); v.lim1 = GW_BYTES - (A.o - ws1);
            v.c2 = ((uint32_t)lane + 1u) * GW_BYTES + A.o - ws2;         // (mod 2^32: added to an offset >= ws2 - o it is a place in the next window)
            v.thr = 0xffffffffu;
            adm = scan_record(P, v, o, D, h, ok, redo);
            if(redo) { far = true; redo = false; ok = true; adm = scan_record(P, GlobalView{P.raw + o}, o, D, h, ok, redo); }
        }
        if(!ok) atomicExch(&P.cnt->malformed, 1u);
    }
    return adm;
}
// record i through the window if it lies inside whole, from HBM otherwise
__device__ __forceinline__ int scan_one(const PrepParams &P, const int i, const RecAt &A, const uint8_t *const win, const uint32_t wbase, const uint32_t wlen, PrepRead &D, uint64_t &h) {
    int adm = 0;
    if(i < P.n_rec) {
        const uint64_t o = A.o;
        bool ok = o + 4 + 32 <= P.raw_bytes;
        if(ok) {
            uint32_t bs = 0;
            const bool inwin = o >= wbase && o + 8 <= (uint64_t)wbase + wlen && (bs = LdsView{(const uint32_t *)win, (uint32_t)(o - wbase)}.u32(0), o + 4 + (uint64_t)bs <= (uint64_t)wbase + wlen);
            bool redo = false;
            if(inwin) adm = scan_record(P, LdsView{(const uint32_t *)win, (uint32_t)(o - wbase)}, o, D, h, ok, redo);
            else adm = scan_record(P, GlobalView{P.raw + o}, o, D, h, ok, redo);
        }
        if(!ok) atomicExch(&P.cnt->malformed, 1u);
    }
    return adm;
}
#if PREP_STAGE == 2
#define SCAN_WAVE_LDS(NP) GW_LDS(NP)
#else
#define SCAN_WAVE_LDS(NP) RAWWIN_LDS
#endif
// the wavefront's 64 records from i0: staged as PREP_STAGE says, then every lane takes its own apart
template <int NP>
__device__ __forceinline__ int scan_wave(const PrepParams &P, const int i, const int i0, const int lane, uint8_t *const win, PrepRead &D, uint64_t &h) {
    const RecAt at = rec_at(P, i);
#if PREP_STAGE == 2
    (void)i0;
    stage_gather<NP>(P, i, at, lane, win);
    bool far = false;
    const int adm = scan_one_gather<NP>(P, i, at, lane, win, D, h, far);
    const unsigned long long fm = __ballot(far);          // records whose head the window did not hold: the host widens the windows when they are many (prep_outcome)
    if(fm && lane == 0) atomicAdd(&P.cnt->far, (uint32_t)__popcll(fm));
    return adm;
#else
    uint32_t wbase, wlen;
    stage_window(P, i0, at, lane, win, wbase, wlen);
    return scan_one(P, i, at, win, wbase, wlen, D, h);
#endif
}
#define RDQ 3                         // quads of a PrepRead
static_assert(sizeof(PrepRead) == RDQ * sizeof(uint4), "PrepRead in quads");
__device__ __forceinline__ void stage_read(uint4 *st, const PrepRead &D, const int adm) {
    st[0] = make_uint4((uint32_t)D.pos, (uint32_t)D.rend, (uint32_t)D.ncig | (uint32_t)D.flag << 16, (uint32_t)D.strand | (uint32_t)D.nlen << 8 | (uint32_t)(adm ? 1u : 0u) << 16 | (uint32_t)D.lqn << 24);
    st[1] = make_uint4(D.name[0], D.name[1], D.name[2], D.name[3]);
    st[2] = make_uint4(D.lq, (uint32_t)D.prev, D.cig[0], D.cig[1]);
}

// barrier that orders LDS traffic only (does not drain this wavefront's outstanding global loads, stores and atomics)
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the tk-th of the workgroups that work for chunk cj (chunk_of_block): XCDs cj, cj + n, ... (< 8) serve it
__device__ __forceinline__ uint32_t static_ticket(const PrepMulti &M, int cj) {
    const uint32_t xcd = blockIdx.x & 7u, per = (7u - (uint32_t)cj) / (uint32_t)M.n + 1u;
    return (blockIdx.x >> 3) * per + xcd / (uint32_t)M.n;
}
// A group of reads of one name (its first read i, `mine` = the name's key and the group's last read) enters the chunk's table at slot sl;
// `old` is what its first compare-and-swap against an empty entry saw.  The read that headed the name before is linked to it both ways.
__device__ __forceinline__ void table_insert(const PrepParams &P, int i, uint32_t sl, unsigned long long key, unsigned long long mine, unsigned long long old) {
    int32_t before = -1;
    for(;;) {
        if(old == 0ull) break;
        if((old >> 32) == (key >> 32)) {
            for(;;) { const unsigned long long seen = sync_cas(&P.hent[sl], old, mine); if(seen == old) break; old = seen; }       // (only the reads of this very name compete here)
            before = (int32_t)(uint32_t)old - 1;
            break;
        }
        sl = (sl + 1) & P.hmask;
        old = sync_cas(&P.hent[sl], 0ull, mine);
    }
    P.hnext[i] = before;
    if(before >= 0) P.hfwd[before] = i;
}

// extract / mbias: every record's PrepRead at the record's own index; no workgroup waits for another.
// Workgroup b works for chunk (b mod 8) mod n (chunk_of_block) and is the tk-th of the workgroups that do: tk follows from b alone.
template <int NP>
__device__ __forceinline__ void prep_scan_body(const PrepMulti &M) {
    __shared__ uint32_t wcnt[PB / 64]; __shared__ int32_t wlast[PB / 64];
    extern __shared__ __align__(16) uint4 dyn[];          // PB/64 windows of RAWWIN_LDS bytes; afterwards the workgroup's PrepReads on their way out (64 bytes each)
    uint4 *const stage = dyn;
    const int cj = chunk_of_block(M);
    const PrepParams &P = M.P[cj];
    const uint32_t tk = static_ticket(M, cj);
    if(tk >= (uint32_t)P.nblocks) return;                 // more workgroups than this chunk needs
    const int i = (int)(tk * PB + threadIdx.x), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PrepRead D; memset(&D, 0, sizeof(D)); uint64_t h = 0;
    uint8_t *const win = (uint8_t *)dyn + (size_t)wave * SCAN_WAVE_LDS(NP);
    const int adm = scan_wave<NP>(P, i, (int)(tk * PB) + 64 * wave, lane, win, D, h);
    // the start of the read admitted just before this one: the nearest admitted lane below in the wavefront, else the last admitted read of
    // the nearest wavefront below in the workgroup, else (not the chunk's first workgroup) left to k_prep_segs
    const unsigned long long m = __ballot(adm), below = m & ((1ull << lane) - 1ull);
    const int32_t pv_wave = __shfl(D.pos, below ? 63 - __clzll((long long)below) : 0), top = __shfl(D.pos, m ? 63 - __clzll((long long)m) : 0);
    if(lane == 0) { wcnt[wave] = (uint32_t)__popcll(m); wlast[wave] = top; }
    if(P.cfg.no_pairing) {                                // mbias: the longest admitted read sizes the histogram rows kept in LDS
        uint32_t v = adm ? D.lq : 0u;
#pragma unroll
        for(int d = 32; d; d >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, d); v = t > v ? t : v; }
        if(lane == 0 && v) atomicMax(&P.cnt->max_lq, v);
    }
    // Name table: open addressing; an entry is (high half of the name's hash, the name's latest read).  One 8-byte word per name, so an
    // insertion touches one line of a 2 MB table: the first read of a name takes an empty entry with one compare-and-swap, a later one
    // replaces the head with a second -- and learns who was there: the two are linked both ways (hnext, hfwd).
    // A device-scope atomic is 64 bytes of memory traffic and a round trip beyond the L2 (the scan without them: 132 us per launch instead
    // of 187, profiles/r05e_prep_variants.txt), and a read's mate is nearly always a few dozen records away -- in this very workgroup.  So
    // the workgroup first groups its own reads by name hash in LDS (a small table, LDS atomics) and links each group among itself; ONE
    // lane per group then inserts the whole group into the chunk's table as if its reads had arrived one after the other: 0.65 atomics
    // per read instead of 1.5.
    const bool ins = adm && !P.cfg.no_pairing;
    lds_only_barrier();                                   // (every wavefront has parsed its records: the windows' memory is free)
    if(adm) {
        int32_t pv = pv_wave;
        if(!below) {
            pv = tk == 0 ? PREP_PREV_NONE : PREP_PREV_UNKNOWN;
            for(int w = wave - 1; w >= 0; w--) if(wcnt[w]) { pv = wlast[w]; break; }
        }
        D.prev = pv;
    }
    if(threadIdx.x == 0) {
        uint32_t t = 0; int32_t lastpos = PREP_PREV_NONE;
        for(int w = 0; w < PB / 64; w++) { t += wcnt[w]; if(wcnt[w]) lastpos = wlast[w]; }
        if(t) atomicAdd(&P.cnt->n_adm, t);
        P.cntA[tk] = (uint32_t)lastpos;                    // the start of this workgroup's last admitted read: what the first one of the next workgroup was admitted after (k_prep_segs block_prev)
    }
    const int i0 = (int)(tk * PB);
    int32_t lprev = -2;                                   // the read of this workgroup linked in front of this one; -1: none, this lane inserts the group; -2: does not take part
    uint32_t ghead = threadIdx.x;                         // the group's last read (what the chunk's table will name)
#if PREP_LOCAL
    {
        // (behind the PrepRead stage in the windows' memory) lh: every lane's hash; lt: open addressing, 2 PB entries of lane + 1; lhead: per
        // group -- filed under the lane that took the entry -- the lane that joined last
        unsigned long long *const lh = (unsigned long long *)(stage + RDQ * PB); uint32_t *const lt = (uint32_t *)(lh + PB), *const lhead = lt + 2 * PB;
        lh[threadIdx.x] = ins ? h : 0ull; lt[threadIdx.x] = 0u; lt[threadIdx.x + PB] = 0u; lhead[threadIdx.x] = 0xffffffffu;
        lds_only_barrier();
        uint32_t rep = threadIdx.x;
        if(ins) {
            uint32_t e = (uint32_t)(h >> 20) & (2u * PB - 1u);          // (other bits than the chunk table's slot)
            for(;;) {
                const uint32_t was = atomicCAS(&lt[e], 0u, threadIdx.x + 1u);
                if(was == 0u) break;
                if(lh[was - 1u] == h) { rep = was - 1u; break; }
                e = (e + 1u) & (2u * PB - 1u);
            }
            lprev = (int32_t)atomicExch(&lhead[rep], threadIdx.x);        // 0xffffffff = -1: the first of its group to get here
        }
        lds_only_barrier();
        if(ins && lprev == -1) ghead = lhead[rep];
    }
#else
    if(ins) lprev = -1;
#endif
    // the group's compare-and-swap is on its way from here; its answer is looked at only after the PrepReads have left
    const bool gins = ins && lprev == -1;
    const unsigned long long key = (unsigned long long)(uint32_t)(h >> 32) << 32, mine = key | (unsigned long long)((uint32_t)i0 + ghead + 1u);
    uint32_t sl = (uint32_t)h & P.hmask;
    unsigned long long old = 0ull;
#if PREP_EXP_NOCAS                                        // TIMING EXPERIMENT ONLY (wrong results): a plain store where the compare-and-swap is
    if(gins) P.hent[sl] = mine;
#else
    if(gins) old = sync_cas(&P.hent[sl], 0ull, mine);
#endif
    if(ins && lprev >= 0) { P.hnext[i] = i0 + lprev; P.hfwd[i0 + lprev] = i; }
    // the PrepReads go to the workgroup's stage in LDS first and from there to rd[] as whole lines: written straight from the lanes, the four
    // quads would leave in four store instructions of 16 bytes per 64 -- partial lines, which the memory side does not merge
    stage_read(stage + RDQ * threadIdx.x, D, adm);
    lds_only_barrier();
    {
        const int cnt = P.n_rec - i0 < PB ? P.n_rec - i0 : PB;
        uint4 *out = (uint4 *)(P.rd + i0);
#if !(PREP_EXP_SCAN & 1)
        for(int q = threadIdx.x; q < RDQ * cnt; q += PB) out[q] = stage[q];
#else
        if(cnt < 0) out[threadIdx.x] = stage[threadIdx.x];
#endif
    }
    if(gins) table_insert(P, i, sl, key, mine, old);
}

__global__ __launch_bounds__(PB) void k_prep_scan(const PrepMulti M) { prep_scan_body<GW_NP>(M); }
__global__ __launch_bounds__(PB) void k_prep_scan_wide(const PrepMulti M) { prep_scan_body<GW_NP_WIDE>(M); }       // (the same with wider windows: enqueue_prep_group)

// perRead: the selected reads compacted IN FILE ORDER (rd[a], aidx[a] = the a-th kept record): a workgroup draws a ticket, publishes its
// count, and adds up the counts of the tickets before it
__global__ __launch_bounds__(PB) void k_prep_scan_ordered(const PrepMulti M) {
    constexpr int NP = GW_NP;
    __shared__ uint32_t s_tk, wcnt[PB / 64], red[PB / 64];
    extern __shared__ __align__(16) uint4 dyn[];
    uint4 *const stage = dyn;
    const PrepParams &P = M.P[chunk_of_block(M)];
    if(threadIdx.x == 0) s_tk = sync_add(&P.ticket[0], 1u);
    __syncthreads();
    const uint32_t tk = s_tk; const int i = (int)(tk * PB + threadIdx.x), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if(tk >= (uint32_t)P.nblocks) return;                 // more workgroups than tickets for this chunk
    PrepRead D; memset(&D, 0, sizeof(D)); uint64_t h = 0;
    uint8_t *const win = (uint8_t *)dyn + (size_t)wave * SCAN_WAVE_LDS(NP);
    const int adm = scan_wave<NP>(P, i, (int)(tk * PB) + 64 * wave, lane, win, D, h);
    const unsigned long long m = __ballot(adm);
    if(lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), total = 0;
    for(int w = 0; w < PB / 64; w++) { if(w < wave) rank += wcnt[w]; total += wcnt[w]; }
    if(threadIdx.x == 0) sync_set(&P.cntA[tk], total | CNT_READY);
    const uint32_t base = tickets_before(P.cntA, tk, red);
    if((int)tk == P.nblocks - 1 && threadIdx.x == 0) P.cnt->n_adm = base + total;
    if(adm) {
        if(P.aidx) P.aidx[base + rank] = (uint32_t)i;
        D.prev = PREP_PREV_UNKNOWN;
        stage_read(stage + RDQ * rank, D, 1);
    }
    __syncthreads();
    {
        uint4 *out = (uint4 *)(P.rd + base);
        for(uint32_t q = threadIdx.x; q < RDQ * total; q += PB) out[q] = stage[q];
    }
}

// A PrepRead in registers: its four quads as they are loaded, fields picked out with shifts where they are used.  (As a struct copied and
// zeroed whole, the compiler splits it into BYTES -- one register per byte of every word that holds a byte field -- and k_prep_segs
// needed 86 VGPRs for two of them.)
struct RdRegs {
    uint4 q0, q1, q2; uint32_t o;                      // its three quads; o: where its record lies (rec_off)
    __device__ __forceinline__ int32_t pos() const { return (int32_t)q0.x; }
    __device__ __forceinline__ int32_t rend() const { return (int32_t)q0.y; }
    __device__ __forceinline__ uint32_t ncig() const { return q0.z & 0xffffu; }
    __device__ __forceinline__ uint32_t flag() const { return q0.z >> 16; }
    __device__ __forceinline__ uint32_t strand() const { return q0.w & 255u; }
    __device__ __forceinline__ uint32_t nlen() const { return (q0.w >> 8) & 255u; }
    __device__ __forceinline__ bool adm() const { return (q0.w >> 16) & 1u; }
    __device__ __forceinline__ uint32_t qn_off() const { return o + 36u; }
    __device__ __forceinline__ uint32_t cig_off() const { return o + 36u + (q0.w >> 24); }
    __device__ __forceinline__ uint32_t seq_off() const { return o + 36u + (q0.w >> 24) + 4u * (q0.z & 0xffffu); }
    __device__ __forceinline__ uint32_t lq() const { return q2.x; }
    __device__ __forceinline__ int32_t prev() const { return (int32_t)q2.y; }
};
__device__ __forceinline__ RdRegs rd_zero() { RdRegs R; R.q0 = R.q1 = R.q2 = make_uint4(0, 0, 0, 0); R.o = 0; return R; }
// rd[x] with the offset of ITS record: record x's (extract, mbias: a PrepRead per record), or -- perRead: rd[] holds the kept reads only -- record aidx[x]'s
__device__ __forceinline__ RdRegs rd_load(const PrepParams &P, const uint32_t x) { const uint4 *q = (const uint4 *)&P.rd[x]; RdRegs R; R.q0 = q[0]; R.q1 = q[1]; R.q2 = q[2]; R.o = P.rec_off[P.aidx ? P.aidx[x] : x]; return R; }

// what pairing looks at in another read of the name: quad 0 (pos rend ncig|flag strand|nlen|adm), quad 1 (the name's first 16 bytes) and the
// last word (the start of the read admitted before it) of its PrepRead, taken apart in registers (a PrepRead filled through a pointer would
// live in scratch)
struct OtherRead { int32_t rend, prev; uint32_t flag, nlen; uint4 name; };
__device__ __forceinline__ OtherRead other_read(const PrepRead *rd, const int32_t x) {
    const uint4 *q = (const uint4 *)&rd[x]; const uint4 q0 = q[0], q1 = q[1]; const uint32_t pv = ((const uint32_t *)q)[9];
    OtherRead O; O.rend = (int32_t)q0.y; O.flag = q0.z >> 16; O.nlen = (q0.w >> 8) & 255u; O.name = q1; O.prev = (int32_t)pv;
    return O;
}
__device__ __forceinline__ bool same_name(const uint8_t *raw, const uint32_t xnlen, const uint4 xname, const uint32_t xqn_off, const uint32_t ynlen, const uint4 yname, const uint32_t yqn_off) {       // strcmp == 0
    if(xnlen != ynlen || xname.x != yname.x || xname.y != yname.y || xname.z != yname.z || xname.w != yname.w) return false;
    if(xnlen <= 16) return true;
    const uint8_t *p = raw + xqn_off, *q = raw + yqn_off;
    for(uint32_t k = 16; k < xnlen; k += 8) {             // eight letters at a time (a sequencer's names are ~40 letters: every pair comes through here; the reads past the name stay inside the record, or the buffer's 64 spare bytes)
        uint64_t a = ld64(p + k), b = ld64(q + k);
        if(xnlen - k < 8) { const uint64_t m = (1ull << (8 * (xnlen - k))) - 1ull; a &= m; b &= m; }
        if(a != b) return false;
    }
    return true;
}
// The start of the read admitted just before the first admitted read of block `blk` (the records one workgroup of k_prep_scan took): the
// last admitted start of the nearest block below that admitted anything.  v = last[blk - 1], asked for ahead of time by the caller.
__device__ __forceinline__ int32_t block_prev(const uint32_t *last, int32_t blk, int32_t v) {
    if(blk <= 0) return PREP_PREV_NONE;
    while(v == PREP_PREV_NONE && --blk > 0) v = (int32_t)last[blk - 1];
    return v;
}
// the same by walking the records (the rare path's way): x is the first admitted read of its block, so the one before it is the nearest
// admitted record below that block's first
__device__ __forceinline__ int32_t prev_of(const PrepRead *rd, const int32_t x, const int32_t prev) {
    if(prev != PREP_PREV_UNKNOWN) return prev;
    for(int32_t j = (x & ~(PB - 1)) - 1; j >= 0; j--) {
        const uint4 q0 = *(const uint4 *)&rd[j];
        if((q0.w >> 16) & 1u) return (int32_t)q0.x;
    }
    return PREP_PREV_NONE;
}
// The reads of a's name, all of them (more than two, or two names with one hash): the chain from its newest read back, taken in file order
// (the smallest index not done yet, found by walking the chain again: a handful of reads, and no array of them), through the machine step
// by step.  Returns the read `a` is resolved against, | PAIR_SECOND when `a` is the later of the two; -1: none; -2: the host must prepare
// this chunk.  (As a real call it cost the kernel 52 bytes of scratch per lane -- and every launch ~30 us at each kernel boundary while the
// runtime found room for it, profiles/r05c_prep_variants.txt; inlined, its state lives in LDS and the kernel needs 57 VGPRs.)
#define PAIR_SECOND 0x40000000
struct PairCtx { const int32_t *hnext, *hfwd; const PrepRead *rd; const uint32_t *rec_off; const uint8_t *raw; int32_t tid; };
__device__ __forceinline__ int32_t pair_of_many(const PairCtx X, MdkPairState &S, const uint32_t a, const int32_t a_rend, const int32_t a_prev, const uint32_t a_flag, const uint32_t a_nlen, const uint4 a_name, const uint32_t a_qn_off) {
    int32_t head = (int32_t)a; int k = 0;
    for(int guard = 0; X.hfwd[head] >= 0; head = X.hfwd[head]) if(++guard > MAXG) return -2;
    for(int32_t x = head; x >= 0; x = X.hnext[x]) if(++k > MAXG) return -2;
    mdk_pair_init(S);
    for(int32_t last = -1;;) {
        int32_t x = 0x7fffffff;
        for(int32_t y = head; y >= 0; y = X.hnext[y]) if(y > last && y < x) x = y;
        if(x == 0x7fffffff) break;
        last = x;
        int32_t rend = a_rend, prev = a_prev; uint32_t flag = a_flag;
        if((uint32_t)x != a) { const OtherRead O = other_read(X.rd, x); if(!same_name(X.raw, a_nlen, a_name, a_qn_off, O.nlen, O.name, X.rec_off[x] + 36u)) continue; rend = O.rend; flag = O.flag; prev = O.prev; }      // (continue: another name with the same hash)
        prev = prev_of(X.rd, x, prev);
        mdk_pair_step(S, X.tid, a, x, flag, rend, prev == PREP_PREV_NONE, prev);
        if(S.overflow) return -2;
    }
    return S.mate < 0 ? -1 : (S.mate | (S.second ? PAIR_SECOND : 0));
}

// gapless runs of a CIGAR, one at a time (calculate_positions, overlaps.c:27-52)
struct RunIt {
    uint32_t cig_off; unsigned long long pk; int n, k; int32_t x, y, lq;      // the first three operations travel with the read (PrepRead::cig, 21 bits each); the others are read where they lie
    int32_t rx, ry, rl;                                               // the run at hand: rl > 0 while there is one
    __device__ __forceinline__ bool valid() const { return rl > 0; }
    __device__ void init(const uint8_t *raw, const RdRegs &r) { cig_off = r.cig_off(); pk = (unsigned long long)r.q2.z | (unsigned long long)r.q2.w << 32; n = (int)r.ncig(); k = 0; x = r.pos(); y = 0; lq = (int32_t)r.lq(); rx = ry = rl = 0; next(raw); }
    __device__ void stop() { rl = 0; }
    __device__ void next(const uint8_t *raw) {
        rl = 0;
        while(k < n) {
            const uint32_t c = (k < 3 && (long long)pk < 0) ? (uint32_t)(pk >> (21 * k)) & 0x1fffffu : ld32(raw + cig_off + 4 * k); k++;
            const int op = c & 15; const int32_t len = (int32_t)(c >> 4);
            if(op == 0 || op == 7 || op == 8) {
                int32_t l = len; if(y + l > lq) l = lq - y;          // a CIGAR that consumes more bases than the record stores
                const int32_t sx = x, sy = y;
                x += len; y += len;
                if(l > 0) { rx = sx; ry = sy; rl = l; return; }
            } else if(op == 1 || op == 4) y += len;
            else if(op == 2 || op == 3) x += len;
        }
    }
};

// lo/hi: reference extent of the pieces written (for the tile runs); untouched when nothing is emitted.  A piece leaves as the two quads of
// its md_seg through `sink(index, quad0, quad1)` -- into the workgroup's stage in LDS, or (a workgroup with more pieces than the stage holds)
// straight into the segment array.
struct NoSink { __device__ __forceinline__ void operator()(uint32_t, uint4, uint4) const {} };
template <bool WRITE, typename Sink>
__device__ __forceinline__ uint32_t read_segments(const PrepParams &P, const RdRegs &r, const bool has_mate, const RdRegs &m, const bool is_second, const Sink sink, const uint32_t base, const uint32_t limit, int32_t &lo, int32_t &hi) {
    const bool paired = has_mate && (((int)r.strand() - (int)m.strand()) & 1) == 0;       // overlaps.c:63-65
    RunIt own, oth;
    own.init(P.raw, r);
    if(paired) oth.init(P.raw, m); else oth.stop();
    const uint32_t sf = (r.strand() & 7) | ((r.flag() & 0x80) ? MDK_SF_READ2 : 0) | (is_second ? MDK_SF_SECOND : 0);
    const uint32_t msf = paired ? ((m.strand() & 7) | ((m.flag() & 0x80) ? MDK_SF_READ2 : 0)) : 0;
    uint32_t n = 0;
    for(; own.valid(); own.next(P.raw)) {
        int32_t cur = own.rx; const int32_t stop = own.rx + own.rl;
        while(cur < stop) {
            int32_t pe = stop; bool covered = false;
            while(oth.valid() && oth.rx + oth.rl <= cur) oth.next(P.raw);
            if(oth.valid()) { if(oth.rx <= cur) { covered = true; if(oth.rx + oth.rl < pe) pe = oth.rx + oth.rl; } else if(oth.rx < pe) pe = oth.rx; }
            if(pe - cur > 65535) pe = cur + 65535;
            if((int64_t)pe > P.beg && (int64_t)cur < P.end) {
                if(WRITE) {
                    // md_seg: rpos off4 l_qseq q0 | len sf msf  m_off4 m_l_qseq m_q0
                    const uint4 g0 = make_uint4((uint32_t)cur, r.seq_off(), r.lq(), (uint32_t)(own.ry + (cur - own.rx)));
                    uint4 g1 = make_uint4((uint32_t)(pe - cur) & 0xffffu | sf << 16, 0u, 0u, 0u);
                    if(covered) g1 = make_uint4((uint32_t)(pe - cur) & 0xffffu | (sf | MDK_SF_PARTNER) << 16 | msf << 24, m.seq_off(), m.lq(), (uint32_t)(oth.ry + (cur - oth.rx)));
                    const uint32_t o = base + n;
                    if(o < limit) sink(o, g0, g1);
                    if(cur < lo) lo = cur;
                    if(pe > hi) hi = pe;
                }
                n++;
            }
            cur = pe;
        }
    }
    return n;
}
static_assert(sizeof(md_seg) == 32 && offsetof(md_seg, len) == 16 && offsetof(md_seg, sf) == 18 && offsetof(md_seg, msf) == 19 && offsetof(md_seg, m_off4) == 20, "md_seg layout (read_segments builds it as two quads)");

#define SEG_STAGE 512                 // segments of one workgroup staged in LDS (16 KB)
#define SEG_TSPAN 32                  // tiles a workgroup's reads may reach for their runs to be merged in LDS
#ifndef SEGS_SGPRS
#define SEGS_SGPRS 96                 // cap on the kernel's scalar registers: 104 of them cost the eighth wavefront per SIMD; the excess spills into lanes of a VGPR (measured: 172.6 -> 162.6 us per launch, profiles/r05d_prep_variants.txt).  0: no cap
#endif
#if SEGS_SGPRS
__global__ __launch_bounds__(PB, 8) __attribute__((amdgpu_num_sgpr(SEGS_SGPRS))) void k_prep_segs(const PrepMulti M) {
#else
__global__ __launch_bounds__(PB, 8) void k_prep_segs(const PrepMulti M) {
#endif
    __shared__ uint32_t s_tk, wsum[PB / 64], red[PB / 64]; __shared__ unsigned long long wbytes[PB / 64];
    // the workgroup's segments on their way out (32 bytes each; a workgroup of 256 records makes ~350); before that, the rare path's state
    __shared__ __align__(16) uint4 sstage[2 * SEG_STAGE];
    __shared__ int s_tlo, s_thi, tfirst[SEG_TSPAN], tlast[SEG_TSPAN];      // the tiles the workgroup's segments reach, and per tile the run of them (relative to the workgroup's first)
    static_assert(sizeof(sstage) >= PB * sizeof(MdkPairState), "the rare path keeps a state per thread in the stage's memory");
    const PrepParams &P = M.P[chunk_of_block(M)];
#if PREP_EXP_SEGS & 8
    if(threadIdx.x == 0) s_tk = (blockIdx.x >> 3) * ((7u - (uint32_t)chunk_of_block(M)) / (uint32_t)M.n + 1u) + (blockIdx.x & 7u) / (uint32_t)M.n;
#else
    if(threadIdx.x == 0) s_tk = sync_add(&P.ticket[1], 1u);
#endif
    if(threadIdx.x == 0) { s_tlo = 0x7fffffff; s_thi = -1; }
    if(threadIdx.x < SEG_TSPAN) { tfirst[threadIdx.x] = 0x7fffffff; tlast[threadIdx.x] = 0; }
    __syncthreads();
    const uint32_t tk = s_tk, n_rec = (uint32_t)P.n_rec;
    if(tk * PB >= n_rec) return;                          // (a workgroup that leaves here is never waited for: every ticket before an active one is active)
    const uint32_t a = tk * PB + threadIdx.x; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RdRegs r = rd_zero(), m = rd_zero(); bool has_mate = false, is_second = false; int32_t mate = -1;
    int32_t lnext = -1, lfwd = -1;
    const int32_t mylast = tk > 0 ? (int32_t)P.cntA[tk - 1] : PREP_PREV_NONE;      // (one scalar load, on its way with the rest)
    if(a < n_rec) { r = rd_load(P, a); if(!P.cfg.no_pairing) { lnext = P.hnext[a]; lfwd = P.hfwd[a]; } }      // (hnext of a record that was not admitted is whatever the buffer held: looked at only under adm)
    const bool active = a < n_rec && r.adm();
    uint32_t n = 0; unsigned long long bytes = 0;
    if(active) {
        if(!P.cfg.no_pairing && mdk_pairs(r.flag())) {    // (a read that cannot pair never becomes pending; it still occupies the buffer for the others)
            const int32_t c = lnext >= 0 ? lnext : lfwd;  // the read that was in the table before this one, else the one that came after
            if(c >= 0) {
                // exactly two reads under the name unless the chain goes on at either end: this read has both neighbours, or the other
                // read has one on its far side
                const int32_t cn = P.hnext[c], cf = P.hfwd[c];
                m = rd_load(P, (uint32_t)c);
                const int32_t cblk = c / PB, clast = cblk > 0 ? (int32_t)P.cntA[cblk - 1] : PREP_PREV_NONE;      // (asked for with the rest, used if c turns out to be its block's first admitted read)
                const bool more = lnext >= 0 ? (lfwd >= 0 || cn >= 0) : cf >= 0;
                const int32_t rprev = r.prev() == PREP_PREV_UNKNOWN ? block_prev(P.cntA, (int32_t)tk, mylast) : r.prev();
                const int32_t mprev = m.prev() == PREP_PREV_UNKNOWN ? block_prev(P.cntA, cblk, clast) : m.prev();
                int32_t mi = -1;
                if(!more) {
                    if(same_name(P.raw, r.nlen(), r.q1, r.qn_off(), m.nlen(), m.q1, m.qn_off())) {       // (else: two names with one hash, each alone)
                        // two reads f < s, one of them this one: the rule in closed form (mdk_pair_two)
                        const bool a_first = a < (uint32_t)c;
                        const int32_t f = a_first ? (int32_t)a : c, sx = a_first ? c : (int32_t)a;
                        const int32_t prev_f = a_first ? rprev : mprev, prev_s = a_first ? mprev : rprev;
                        mi = mdk_pair_two(P.tid, a, f, sx, a_first ? r.flag() : m.flag(), a_first ? r.rend() : m.rend(), prev_f == PREP_PREV_NONE, prev_f,
                                          a_first ? m.flag() : r.flag(), a_first ? m.rend() : r.rend(), prev_s, is_second);
                    }
                } else {
                    PairCtx X; X.hnext = P.hnext; X.hfwd = P.hfwd; X.rd = P.rd; X.rec_off = P.rec_off; X.raw = P.raw; X.tid = P.tid;
                    mi = pair_of_many(X, ((MdkPairState *)sstage)[threadIdx.x], a, r.rend(), r.prev(), r.flag(), r.nlen(), r.q1, r.qn_off());
                    if(mi == -2) { atomicExch(&P.cnt->fallback, 1u); mi = -1; }
                    if(mi >= 0) { is_second = (mi & PAIR_SECOND) != 0; mi &= ~PAIR_SECOND; m = rd_load(P, (uint32_t)mi); }
                }
                has_mate = mi >= 0; mate = mi;
            }
        }
        int32_t lo = 0, hi = 0;
        n = read_segments<false>(P, r, has_mate, m, is_second, NoSink(), 0u, 0u, lo, hi);
        bytes = 16ull + 4ull * r.ncig() + ((unsigned long long)r.lq() + 1) / 2 + r.lq();        // SURVEY.md 8d, per admitted read
    }
    // where this read's segments go: scan inside the workgroup (wpos: its place among the workgroup's), plus what the earlier tickets counted
    uint32_t incl = n;
#pragma unroll
    for(int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if(lane >= d) incl += t; }
    if(lane == 63) wsum[wave] = incl;
    { unsigned long long b = bytes;                       // the workgroup's share of the byte tally: one atomic per workgroup (one per wavefront -- 25,000 on one address per launch -- cost the kernel 4 us)
#pragma unroll
      for(int d = 32; d; d >>= 1) b += __shfl_xor(b, d);
      if(lane == 0) wbytes[wave] = b; }
    __syncthreads();                                      // (also: nobody is in the rare path any more -- the stage's memory is the stage's)
    uint32_t wpos = incl - n, total = 0;
    for(int w = 0; w < PB / 64; w++) { if(w < wave) wpos += wsum[w]; total += wsum[w]; }
    if(threadIdx.x == 0) sync_set(&P.cntS[tk], total | CNT_READY);
    if(!(PREP_EXP_SEGS & 16) && threadIdx.x == 64) { unsigned long long b = 0; for(int w = 0; w < PB / 64; w++) b += wbytes[w]; if(b) atomicAdd((unsigned long long *)&P.cnt->algo_bytes, b); }
    // The segments are made while the earlier tickets' counts are on their way: into the stage in LDS, at their place among the workgroup's
    // (straight from the lanes they would leave as 16-byte stores 32 bytes apart -- partial lines, which the memory side does not merge: the
    // write pass was 92 of the kernel's 177 us, profiles/r05f_prep_variants.txt), and leave it as whole lines.  A workgroup with more
    // segments than the stage holds (long CIGARs) goes round again for the next stage-full.
    int32_t lo = INT32_MAX, hi = INT32_MIN;             // (reference positions are 32 bits: BAM's pos)
    int t0 = 0x7fffffff, t1 = -1;
    uint32_t before = 0;
    const uint32_t cap = P.cap_seg > 0xffffffffll ? 0xffffffffu : (uint32_t)P.cap_seg;
    uint4 *const st = sstage;
    if(n && !(PREP_EXP_SEGS & 2))                        // (every lane with segments: lo and hi are wanted of all of them, the stage takes what fits)
        (void)read_segments<true>(P, r, has_mate, m, is_second, [st](uint32_t o, uint4 g0, uint4 g1) { st[2 * o] = g0; st[2 * o + 1] = g1; }, wpos, (uint32_t)SEG_STAGE, lo, hi);
    if(n && hi > lo) {
        const int64_t l = lo < P.beg ? P.beg : (int64_t)lo, u = hi > P.end ? P.end : (int64_t)hi;
        t0 = (int)((uint32_t)(l - P.beg) / (uint32_t)P.tile); t1 = (int)((uint32_t)(u - 1 - P.beg) / (uint32_t)P.tile);       // (a chunk spans less than 2^31 positions)
        atomicMin(&s_tlo, t0); atomicMax(&s_thi, t1);
    }
    before = (PREP_EXP_SEGS & 4) ? tk * 352u : tickets_before(P.cntS, tk, red);       // (two barriers inside: the stage is complete behind it)
    if((tk + 1) * PB >= n_rec && threadIdx.x == 0) P.cnt->n_segs = before + total;
    {
        const uint32_t left = total < SEG_STAGE ? total : SEG_STAGE;
        const uint32_t avail = cap > before ? (left < cap - before ? left : cap - before) : 0u;
        uint4 *out = (uint4 *)(P.seg + before);
        for(uint32_t q = threadIdx.x; q < 2 * avail; q += PB) out[q] = sstage[q];
    }
    if(total > SEG_STAGE && n && wpos + n > SEG_STAGE && !(PREP_EXP_SEGS & 2)) {
        // (rare: what did not fit the stage leaves from the lanes; the reads are loaded again rather than kept in registers across the wait)
        const RdRegs r2 = rd_load(P, a), m2 = has_mate ? rd_load(P, (uint32_t)mate) : rd_zero();
        md_seg *const sg = P.seg + before; int32_t lo2 = 0, hi2 = 0;
        (void)read_segments<true>(P, r2, has_mate, m2, is_second, [sg](uint32_t o, uint4 g0, uint4 g1) { if(o >= SEG_STAGE) { uint4 *q = (uint4 *)(sg + o); q[0] = g0; q[1] = g1; } }, wpos, cap > before ? cap - before : 0u, lo2, hi2);
    }
    // Tile runs: tile t's run [first, last) must cover every segment touching t.  A lane contributes [its first, its last + 1) to every
    // tile its pieces reach -- a superset, which is all k_pileup needs.  Reads are in coordinate order, so the lanes of a wave
    // touching one tile are (nearly always) consecutive: only the first of them lowers `first`, only the last raises `last`.
    // The workgroup's reads reach a tile or two: their runs are merged in LDS and a lane per tile tells the chunk (two device-scope
    // atomics per tile and workgroup instead of two per tile and wavefront).
    const int tlo = s_tlo, thi = s_thi;
    if(thi >= tlo && !(PREP_EXP_SEGS & 1)) {
        if(thi - tlo < SEG_TSPAN) {
            const int p0 = __shfl_up(t0, 1), p1 = __shfl_up(t1, 1), n0 = __shfl_down(t0, 1), n1 = __shfl_down(t1, 1);
            for(int t = t0; t <= t1; t++) {
                if(lane == 0 || t < p0 || t > p1) atomicMin(&tfirst[t - tlo], (int)wpos);
                if(lane == 63 || t < n0 || t > n1) atomicMax(&tlast[t - tlo], (int)(wpos + n));
            }
            __syncthreads();
            const int k = (int)threadIdx.x;
            if(k <= thi - tlo && tlast[k] > tfirst[k] && before + (uint32_t)tfirst[k] < cap) {
                const uint32_t top = before + (uint32_t)tlast[k];
                atomicMin(&P.tiles[tlo + k].first, (int)(before + (uint32_t)tfirst[k]));
                atomicMax(&P.tiles[tlo + k].last, (int)(top < cap ? top : cap));
            }
        } else {                                          // (reads that skip far: N operations)
            const uint32_t base = before + wpos; uint32_t top = base + n; if(top > cap) top = cap;
            if(base < cap) for(int t = t0; t <= t1; t++) { atomicMin(&P.tiles[t].first, (int)base); atomicMax(&P.tiles[t].last, (int)top); }
        }
    }
}

// perRead over device-selected reads: the walk of k_perread on the records where they lie
__global__ __launch_bounds__(PB) void k_perread_raw(const PrepParams P, const uint8_t *ctxcode, int64_t wend, md_pr_count *out) {
    const uint32_t a = blockIdx.x * PB + threadIdx.x;
    if(a >= P.cnt->n_adm) return;
    const RdRegs r = rd_load(P, a);
    const uint8_t *seq = P.raw + r.seq_off(), *qual = seq + ((r.lq() + 1) >> 1), *cg = P.raw + r.cig_off();
    const unsigned long long pk = (unsigned long long)r.q2.z | (unsigned long long)r.q2.w << 32;
    out[a] = perread_walk(seq, qual, r.lq(), (int)r.ncig(), r.pos(), r.strand() & 1, ctxcode, P.reflen, wend, P.cfg.min_phred, [cg, pk](int k) { return (k < 3 && (long long)pk < 0) ? (uint32_t)(pk >> (21 * k)) & 0x1fffffu : ld32(cg + 4 * k); });
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
    bool any_tab = false; uint64_t nrec_sum = 0, loose = 0;
    for(int i = 0; i < b->n_ranges; i++) { const md_raw_range &r = b->range[i]; if(r.d_rec_off || r.h_rec_off) any_tab = true; else loose += r.n_records; nrec_sum += r.n_records; }
    if((any_tab ? loose : (uint64_t)b->n_records) && !b->rec_off) return fail(MDK_ERR_ARG, "md_dev_upload_raw: null record table", hipSuccess);
    if(any_tab && nrec_sum != (uint64_t)b->n_records) return fail(MDK_ERR_ARG, "md_dev_upload_raw: with a range that has its own record table every range must carry its record count", hipSuccess);
    // The host's record tables (a slab's own, or the slot's list) are ordinary memory: copied from there, every table would be a staged copy that
    // returns when it has been made -- after everything queued on the stream before it, the chunk's 57 MB of records included (the thread that
    // uploads stood 5-35 ms in such a call, a dozen times per run: profiles/r06pf_*).  They go through a pinned table of the slot instead.
    if(b->n_records && s->h_rectab.need((size_t)b->n_records + 1)) return MDK_ERR_NOMEM;
    uint64_t o = 0; uint32_t idx = 0, hidx = 0;
    for(int i = 0; i < b->n_ranges; i++) {
        const md_raw_range &r = b->range[i];
        if(r.d_rec_off) {
            if(r.bytes) HIPCHK(hipMemcpyAsync(s->d_raw.p + o, r.ptr, (size_t)r.bytes, hipMemcpyDeviceToDevice, s->stream));
            if(r.n_records) hipLaunchKernelGGL(k_rebase, dim3((r.n_records + 255) / 256), dim3(256), 0, s->stream, s->d_recoff.p + idx, r.d_rec_off, r.n_records, (uint32_t)o - r.rec_delta);
        } else {
            if(r.bytes) { host_block_ensure_registered(r.ptr); HIPCHK(hipMemcpyAsync(s->d_raw.p + o, r.ptr, (size_t)r.bytes, hipMemcpyHostToDevice, s->stream)); }
            if(r.h_rec_off) {          // the range's own table: as it is, then re-based where it lands
                if(r.n_records) {
                    memcpy(s->h_rectab.p + idx, r.h_rec_off, sizeof(uint32_t) * (size_t)r.n_records);
                    HIPCHK(hipMemcpyAsync(s->d_recoff.p + idx, s->h_rectab.p + idx, sizeof(uint32_t) * (size_t)r.n_records, hipMemcpyHostToDevice, s->stream));
                    hipLaunchKernelGGL(k_rebase, dim3((r.n_records + 255) / 256), dim3(256), 0, s->stream, s->d_recoff.p + idx, (const uint32_t *)(s->d_recoff.p + idx), r.n_records, (uint32_t)o - r.rec_delta);
                }
            } else {
                if(any_tab && r.n_records) {
                    memcpy(s->h_rectab.p + idx, b->rec_off + hidx, sizeof(uint32_t) * (size_t)r.n_records);
                    HIPCHK(hipMemcpyAsync(s->d_recoff.p + idx, s->h_rectab.p + idx, sizeof(uint32_t) * (size_t)r.n_records, hipMemcpyHostToDevice, s->stream));
                }
                hidx += r.n_records;
            }
        }
        idx += r.n_records; o += r.bytes;
    }
    if(!any_tab && b->n_records) {
        memcpy(s->h_rectab.p, b->rec_off, sizeof(uint32_t) * (size_t)b->n_records);
        HIPCHK(hipMemcpyAsync(s->d_recoff.p, s->h_rectab.p, sizeof(uint32_t) * (size_t)b->n_records, hipMemcpyHostToDevice, s->stream));
    }
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int md_dev_read_raw(md_dev *h, int slot, uint8_t *bytes, uint64_t *n_bytes, uint32_t *rec_off, uint32_t *n_records) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->raw_layout || !bytes || !n_bytes || !rec_off || !n_records) return fail(MDK_ERR_ARG, "md_dev_read_raw: needs a slot uploaded with md_dev_upload_raw", hipSuccess);
    if(*n_bytes < s->raw_bytes || *n_records < (uint32_t)s->pr_nrec) { *n_bytes = s->raw_bytes; *n_records = (uint32_t)s->pr_nrec; return fail(MDK_ERR_ARG, "md_dev_read_raw: buffers too small", hipSuccess); }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if(s->raw_bytes) HIPCHK(hipMemcpy(bytes, s->raw_at + s->inplace_delta, (size_t)s->raw_bytes, hipMemcpyDeviceToHost));
    if(s->pr_nrec) HIPCHK(hipMemcpy(rec_off, s->rec_at, sizeof(uint32_t) * (size_t)s->pr_nrec, hipMemcpyDeviceToHost));
    if(s->inplace_delta) for(int i = 0; i < s->pr_nrec; i++) rec_off[i] -= s->inplace_delta;      // (read in place: the table counts from the piece's first byte)
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

#define PREP_SCAN_LDS_MIN ((size_t)(RDQ * PB * sizeof(uint4) + PB * 8 + 2 * PB * 4 + PB * 4))
#if PREP_STAGE
#define PREP_SCAN_LDS_NP(NP) (((size_t)(PB / 64) * SCAN_WAVE_LDS(NP)) > PREP_SCAN_LDS_MIN ? ((size_t)(PB / 64) * SCAN_WAVE_LDS(NP)) : PREP_SCAN_LDS_MIN)
#else
#define PREP_SCAN_LDS_NP(NP) PREP_SCAN_LDS_MIN       /* no windows: the PrepRead stage and the name grouping only */
#endif
#define PREP_SCAN_LDS PREP_SCAN_LDS_NP(GW_NP)
#define PREP_SCAN_LDS_WIDE PREP_SCAN_LDS_NP(GW_NP_WIDE)
static_assert(PREP_SCAN_LDS >= RDQ * PB * sizeof(uint4) + PB * 8 + 2 * PB * 4 + PB * 4, "the PrepRead stage and the workgroup's name grouping live in the windows' memory");
MDK_HIDDEN int prep_kernels_init() {        // more dynamic LDS than the default window: once per process
    static std::once_flag once; static int rc = 1;
    std::call_once(once, [] { rc = (hipFuncSetAttribute((const void *)k_prep_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_SCAN_LDS) == hipSuccess &&
                                    hipFuncSetAttribute((const void *)k_prep_scan_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_SCAN_LDS_WIDE) == hipSuccess &&
                                    hipFuncSetAttribute((const void *)k_prep_scan_ordered, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_SCAN_LDS) == hipSuccess) ? 0 : 1; });
    return rc;
}
static uint32_t pow2_at_least(size_t n) { uint32_t p = 1024; while(p < n) p <<= 1; return p; }

// layout of a slot's zeroed region: hent[H] (8 bytes each), cntA[nb], cntS[nb], ticket[2]; sizes rounded to 16 bytes
static size_t zero_bytes_for(uint32_t hmask, int nb) { return (((size_t)hmask + 1) * 8 + (size_t)(nb > 0 ? nb : 1) * 8 + 8 + 15) & ~(size_t)15; }
static void fill_prep(md_dev *h, Slot *s, PrepParams &P) {
    memset(&P, 0, sizeof(P));
    const int n = s->pr_nrec, nb = (n + PB - 1) / PB; const size_t H = (size_t)s->hmask + 1;
    P.raw = s->raw_at; P.raw_bytes = s->raw_span; P.rec_off = s->rec_at; P.n_rec = n;
    P.cfg = h->prep; P.tid = s->tid; P.beg = s->beg; P.end = s->end; P.woff = s->woff; P.wlen = s->wlen;
    P.ref = h->ref[s->tid]; P.reflen = h->reflen[s->tid];
    if(h->prep.map_on && (size_t)s->tid < h->mapbits.size()) { P.mapbits = h->mapbits[s->tid]; P.maplen = h->maplen[s->tid]; }
    P.bed_on = (size_t)s->tid < h->d_runs.size() && h->has_runs[s->tid]; if(P.bed_on) { P.runs = h->d_runs[s->tid]; P.nruns = h->n_runs[s->tid]; }
    P.rd = s->d_prd.p; P.aidx = h->prep.perread ? s->d_aidx.p : nullptr; P.hmask = s->hmask;
    const bool links = !h->prep.perread && !h->prep.no_pairing;      // the reads of a name linked both ways: hnext[n_rec], hfwd[n_rec] (16-byte aligned: k_prep_zero fills it with -1)
    P.hnext = links ? s->d_hnext.p : nullptr; P.hfwd = links ? s->d_hnext.p + (((size_t)n + 4) & ~(size_t)3) : nullptr;
    uint8_t *z = s->d_zero.p;
    P.hent = (unsigned long long *)z; P.cntA = (uint32_t *)(z + H * 8); P.cntS = P.cntA + (nb > 0 ? nb : 1); P.ticket = P.cntS + (nb > 0 ? nb : 1);
    P.nblocks = nb; P.zero = z; P.zero_bytes = zero_bytes_for(s->hmask, nb);
    P.seg = s->d_seg_in.p; P.cap_seg = (int64_t)s->d_seg_in.cap;
    P.tiles = s->d_tiles.p; P.ntiles = s->ntiles; P.tile = s->tile;
    P.cnt = s->d_pcnt.p;
}
// the preparation of up to MAXM uploaded raw slots on stream `st`: zero, scan (+ compaction, name table), segments -- each kernel
// once for all of them.  The caller has ordered `st` behind the slots' uploads.
int enqueue_prep_group(md_dev *h, Slot *const *ss, int n, hipStream_t st) {
    if(n < 1 || n > MAXM) return fail(MDK_ERR_ARG, "enqueue_prep_group", hipSuccess);
    for(int i = 0; i < n; i++) if((size_t)ss[i]->tid >= h->ref.size() || !h->ref[ss[i]->tid]) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "reference for tid %d not uploaded", ss[i]->tid); return MDK_ERR_NOREF; }
    if(prep_kernels_init()) return fail(MDK_ERR_HIP, "hipFuncSetAttribute(k_prep_scan)", hipGetLastError());
    static_assert(sizeof(PrepMulti) <= 4096, "kernel arguments are limited to 4 KiB");
    PrepMulti M; memset(&M, 0, sizeof(M));
    int total = 0; size_t zmax = 0;
    for(int i = 0; i < n; i++) { fill_prep(h, ss[i], M.P[i]); M.bstart[i] = total; total += M.P[i].nblocks; zmax = std::max<size_t>(zmax, (size_t)M.P[i].zero_bytes); ss[i]->prep_pending = false; }
    M.n = n; M.bstart[n] = total;
    int zgrid = (int)std::min<size_t>(256, (zmax / 16 + PB - 1) / PB); if(zgrid < 1) zgrid = 1;
    hipLaunchKernelGGL(k_prep_zero, dim3(zgrid * n), dim3(PB), 0, st, M);
    if(total > 0) {
        int grid = total;
        {   // chunk j is served by the workgroups b with (b mod 8) mod n == j (chunk_of_block): enough of them for its nblocks tickets
            int per8 = 1;
            for(int j = 0; j < n; j++) { int xcds = 0; for(int x = 0; x < 8; x++) if(x % n == j) xcds++; if(xcds) per8 = std::max(per8, (M.P[j].nblocks + xcds - 1) / xcds); else per8 = std::max(per8, M.P[j].nblocks); }
            grid = 8 * per8;
        }
        static_assert(MAXM <= 8, "chunk_of_block deals the chunks of a launch to 8 XCDs");
        if(h->prep.perread) hipLaunchKernelGGL(k_prep_scan_ordered, dim3(grid), dim3(PB), PREP_SCAN_LDS, st, M);
        else {
            if(h->scan_wide.load(std::memory_order_relaxed)) hipLaunchKernelGGL(k_prep_scan_wide, dim3(grid), dim3(PB), PREP_SCAN_LDS_WIDE, st, M);
            else hipLaunchKernelGGL(k_prep_scan, dim3(grid), dim3(PB), PREP_SCAN_LDS, st, M);
            hipLaunchKernelGGL(k_prep_segs, dim3(grid), dim3(PB), 0, st, M);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}
static int enqueue_prep(md_dev *h, Slot *s, hipStream_t st = nullptr) { Slot *one[1] = {s}; return enqueue_prep_group(h, one, 1, st ? st : s->stream); }

// H2D of the chunk's record bytes and record table, then the preparation kernels: the slot ends up "uploaded", with its
// segments and tile runs in device memory, exactly as after md_dev_upload of a host-built batch.
static int upload_raw(md_dev *h, int slot, const md_raw_batch *b, bool may_stay) {
    Slot *s = get_slot(h, slot);
    if(!s || !b || b->n_records < 0 || b->n_ranges < 0 || b->end < b->beg) return fail(MDK_ERR_ARG, "md_dev_upload_raw", hipSuccess);
    if(!h->prep_set) return fail(MDK_ERR_ARG, "md_dev_upload_raw: md_dev_set_prep was not called", hipSuccess);
    if(b->n_records && !b->range) return fail(MDK_ERR_ARG, "md_dev_upload_raw: null array", hipSuccess);
    if(b->tid < 0) return fail(MDK_ERR_ARG, "md_dev_upload_raw: contig", hipSuccess);      /* (the contig's reference is needed when the slot is launched, not yet here) */
    HIPCHK(hipSetDevice(h->device));
    struct SlowCall { double t0, t_sync = 0, t_alloc = 0; const md_raw_batch *b; bool on; SlowCall(const md_raw_batch *bb) : t0(mdk_prof_on() ? mdk_now() : 0), b(bb), on(mdk_prof_on()) {}
        ~SlowCall() { if(on) { const double t = mdk_now(); if(t - t0 > 0.004) { char w[256]; snprintf(w, sizeof(w), "slow upload_raw: %.1f ms (slot wait %.1f, buffers %.1f, copies %.1f), %d ranges, %d records, first range %s", (t - t0) * 1e3, (t_sync - t0) * 1e3, (t_alloc - t_sync) * 1e3, (t - t_alloc) * 1e3, b->n_ranges, b->n_records, b->n_ranges ? (b->range[0].d_rec_off ? "device" : b->range[0].h_rec_off ? "host+table" : "host") : "-"); mdk_marks_dump(w); } } } } slow(b);
    if(s->busy) {                                    // work of the slot's previous chunk may still be running (its results were not collected)
        ProfScope pf(PF_UP_SYNC);
        HIPCHK(hipStreamSynchronize(s->stream));
        if(s->run && s->run != s->stream) HIPCHK(hipStreamSynchronize(s->run));
    }
    slow.t_sync = slow.on ? mdk_now() : 0;
    s->busy = true;
    s->fresh = true;
    uint64_t total = 0;
    for(int i = 0; i < b->n_ranges; i++) total += b->range[i].bytes;
    if(total >= (1ull << 32) - 64) return fail(MDK_ERR_ARG, "md_dev_upload_raw: more than 4 GiB of records in one chunk", hipSuccess);
    const int64_t span = b->end - b->beg; const int TILE = h->tile;
    const int ntiles = (int)((span + TILE - 1) / TILE), n = b->n_records, nb = (n + PB - 1) / PB;
    s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->woff = b->woff; s->wlen = b->wlen; s->uploaded = false; s->launched = false;
    s->tile = TILE; s->ntiles = ntiles; s->lds_bytes = TILE * ((h->variant ? 16 : 8) + 4);
    s->pr_nrec = n; s->raw_bytes = total; s->raw_layout = true; s->n_segs = -1; s->n_reads = -1; s->read_bytes = 0;
    // one device-resident range with its own record table, and a caller that keeps it: nothing is copied
    static const bool no_inplace = getenv("MDK_NO_INPLACE") != nullptr;
    const bool inplace = may_stay && !no_inplace && b->n_ranges == 1 && b->range[0].d_rec_off && (uint32_t)n == b->range[0].n_records && n > 0 && b->range[0].bytes + (uint64_t)b->range[0].rec_delta < (1ull << 32) - 64;
    const size_t nn = (size_t)n + 1, nt = (size_t)(ntiles > 0 ? ntiles : 1);
    const size_t segcap = std::max<size_t>(s->d_seg_in.cap, nn * 2 + 4096);
    s->hmask = pow2_at_least(nn + nn / 4) - 1;          // names are at most the records: load factor <= 0.8, ~0.4 for pairs; 2 MB for a 1 Mb chunk at 30x
    static std::atomic<int> first_call{1}; const bool first = mdk_prof_on() && first_call.exchange(0); const double tf0 = first ? mdk_now() : 0; double tf1 = 0;
    {
        ProfScope pf(PF_UP_ALLOC);
        if((!inplace && (s->d_raw.need((size_t)total + 64) || s->d_recoff.need(nn))) || s->d_prd.need(nn) || s->d_hnext.need(2 * nn + 8) || s->d_zero.need(zero_bytes_for(s->hmask, nb)) ||
           s->d_seg_in.need(segcap) || s->d_tiles.need(nt) || s->d_seg.need(nt)) return MDK_ERR_NOMEM;
        if(!s->b_site) {
            if(s->d_site.need((size_t)span + 16)) return MDK_ERR_NOMEM;
            if(h->variant && s->d_var.need((size_t)span + 16)) return MDK_ERR_NOMEM;
        }
    }
    if(first) tf1 = mdk_now();
    slow.t_alloc = slow.on ? mdk_now() : 0;
    if(inplace) { const md_raw_range &r = b->range[0]; s->inplace = true; s->inplace_delta = r.rec_delta; s->raw_at = r.ptr - r.rec_delta; s->rec_at = r.d_rec_off; s->raw_span = (uint64_t)r.rec_delta + r.bytes; HIPCHK(hipEventRecord(s->e1, s->stream)); }
    else { ProfScope pf(PF_UP_COPY); s->inplace = false; s->inplace_delta = 0; s->raw_at = s->d_raw.p; s->rec_at = s->d_recoff.p; s->raw_span = total; int rcc = copy_ranges(h, s, b); if(rcc) return rcc; HIPCHK(hipEventRecord(s->e1, s->stream)); }
    if(first) fprintf(stderr, "[mdk hip] the first chunk's upload: device buffers %.3fs, registration + copies queued %.3fs\n", tf1 - tf0, mdk_now() - tf1);
    s->prep_pending = true;            // the preparation kernels are queued with the launch: alone (md_dev_launch) or with up to seven other chunks (md_dev_launch_group)
    s->uploaded = true;
    return inplace ? 1 : 0;
}
extern "C" int md_dev_upload_raw(md_dev *h, int slot, const md_raw_batch *b) { return upload_raw(h, slot, b, false); }
extern "C" int md_dev_upload_raw_inplace(md_dev *h, int slot, const md_raw_batch *b) { return upload_raw(h, slot, b, true); }

extern "C" int md_dev_upload_wait(md_dev *h, int slot) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded) return fail(MDK_ERR_ARG, "md_dev_upload_wait: slot not uploaded", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    if(s->raw_layout) HIPCHK(hipEventSynchronize(s->e1)); else HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int md_dev_upload_done(md_dev *h, int slot) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded) return fail(MDK_ERR_ARG, "md_dev_upload_done: slot not uploaded", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    const hipError_t e = s->raw_layout ? hipEventQuery(s->e1) : hipStreamQuery(s->stream);
    if(e == hipSuccess) return 1;
    if(e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(MDK_ERR_HIP, "hipEventQuery", e);
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
    s->busy = true;
    s->pr_n = -1; s->uploaded = false; s->launched = false;
    uint64_t total = 0;
    for(int i = 0; i < b->n_ranges; i++) total += b->range[i].bytes;
    if(total >= (1ull << 32) - 64) return fail(MDK_ERR_ARG, "md_dev_perread_submit_raw: more than 4 GiB of records in one chunk", hipSuccess);
    const int n = b->n_records, nb = (n + PB - 1) / PB; const size_t nn = (size_t)n + 1;
    s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->pr_nrec = n; s->raw_bytes = total; s->raw_layout = true; s->hmask = 1023; s->ntiles = 0; s->tile = h->tile; s->inplace = false; s->inplace_delta = 0; s->raw_span = total;
    if(s->d_raw.need((size_t)total + 64) || s->d_recoff.need(nn) || s->d_prd.need(nn) || s->d_hnext.need(2 * nn + 8) || s->d_zero.need(zero_bytes_for(s->hmask, nb)) ||
       s->d_aidx.need(nn) || s->h_aidx.need(nn) || s->d_prc.need(nn) || s->h_prc.need(nn)) return MDK_ERR_NOMEM;
    s->raw_at = s->d_raw.p; s->rec_at = s->d_recoff.p;
    { int rcc = copy_ranges(h, s, b); if(rcc) return rcc; }
    { int rc = enqueue_prep(h, s); if(rc) return rc; }                 // perread mode: selection + file-order compaction only (k_prep_scan)
    if(n > 0) {
        PrepParams P; fill_prep(h, s, P);
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
    if(c.fallback) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "a read name with more than %d records or %d live reads: this chunk needs the host preparation", MAXG, MDK_MAXLIVE); return MDK_ERR_PREP_HOST; }
    s->n_reads = (int)c.n_adm; s->n_segs = (int)c.n_segs; s->read_bytes = c.algo_bytes;
    // a library whose records' heads (36 bytes, the name, the CIGAR) do not fit k_prep_scan's windows -- a quarter of a chunk's records read from HBM by
    // their lanes -- gets the wide windows from here on (same results either way; MDK_SCAN_WIDE=0/1 decides beforehand)
    if(!h->scan_wide_fixed && (uint64_t)c.far * 4 > (uint64_t)(s->pr_nrec > 0 ? s->pr_nrec : 0) && !h->scan_wide.exchange(true, std::memory_order_relaxed) && getenv("MDK_HOST_PROFILE"))
        fprintf(stderr, "[mdk hip] k_prep_scan: %u of a chunk's %d records did not fit the windows (long read names): wide windows from here on\n", c.far, s->pr_nrec);
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

// the same for `n` uploaded raw slots holding different intervals, `per_launch` of them per launch of each preparation kernel,
// round robin on one stream, `iters` launches in all: what the preparation costs per launch when its inputs stream from HBM
extern "C" int md_dev_bench_prep_rotate(md_dev *h, const int *slots, int n, int per_launch, int warmup, int iters, float *ms_per_launch) {
    if(!h || !slots || n < 1 || iters < 1 || !ms_per_launch || per_launch < 1 || per_launch > MAXM || n % per_launch) return fail(MDK_ERR_ARG, "md_dev_bench_prep_rotate", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    std::vector<Slot *> ss((size_t)n);
    for(int i = 0; i < n; i++) { ss[i] = get_slot(h, slots[i]); if(!ss[i] || !ss[i]->uploaded || !ss[i]->raw_layout) return fail(MDK_ERR_ARG, "md_dev_bench_prep_rotate: needs slots uploaded with md_dev_upload_raw", hipSuccess); HIPCHK(hipStreamSynchronize(ss[i]->stream)); if(ss[i]->run) HIPCHK(hipStreamSynchronize(ss[i]->run)); }
    Slot *s0 = ss[0]; const int groups = n / per_launch;
    for(int i = 0; i < warmup; i++) { int rc = enqueue_prep_group(h, ss.data() + (i % groups) * per_launch, per_launch, s0->stream); if(rc) return rc; }
    HIPCHK(hipEventRecord(s0->k0, s0->stream));
    for(int i = 0; i < iters; i++) { int rc = enqueue_prep_group(h, ss.data() + (i % groups) * per_launch, per_launch, s0->stream); if(rc) return rc; }
    HIPCHK(hipEventRecord(s0->k1, s0->stream));
    HIPCHK(hipEventSynchronize(s0->k1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, s0->k0, s0->k1));
    for(int i = 0; i < n; i++) ss[i]->fresh = true;
    *ms_per_launch = ms / (float)iters;
    return 0;
}

// test hook: the segments the preparation built (device order), and the admitted reads behind them
extern "C" int md_dev_debug_segments(md_dev *h, int slot, md_seg *out, int64_t cap, int64_t *n_segs, int64_t *n_reads) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded || !n_segs) return fail(MDK_ERR_ARG, "md_dev_debug_segments", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    if(s->raw_layout && s->prep_pending) { int rc = enqueue_prep(h, s); if(rc) return rc; }
    HIPCHK(hipStreamSynchronize(s->stream));
    if(s->run && s->run != s->stream) HIPCHK(hipStreamSynchronize(s->run));
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

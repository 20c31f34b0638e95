// mdk_hip.hip -- MI355X (gfx950 / CDNA4) device library for the `MethylDackel extract` hot path.
//
// The reference sweeps a pileup buffer column by column (htslib bam_mplp64_auto driven from
// extract.c:399-493) and, per column, loops over the reads covering it.  Every per-position output is a plain
// sum over (read, aligned base) pairs, so on the GPU the same result is computed READ-parallel, per tile of
// reference positions, entirely inside one kernel launch per interval:
//
//   k_classify  (once per contig) context code of every reference position (common.c:49-82, precedence
//               extract.c:407-418), 1 byte per base, resident in HBM next to the bases.
//   k_pileup    one workgroup per tile.
//                 1. the tile's context codes -> two sorted LDS lists (C positions, G positions) via wave
//                    ballots; the tile's counters are zeroed in LDS;
//                 2. every LANE owns one SEGMENT -- a gapless run of aligned bases; the host expands CIGARs once per
//                    read (include/mdk_hip.h md_seg), so there is no CIGAR walk and no dependent pointer chase on the
//                    device: a coalesced 32-byte record load, then bases.  The lane walks the slice of the C (OT/CTOT)
//                    or G (OB/CTOB) list its segment covers, K positions at a time: all seq/qual bytes of the read AND
//                    of its overlap partner are requested together, then used: trimming (common.c:137-208) is a
//                    predicate on the query index, the mate overlap is resolved on the fly against the partner's base
//                    at the same reference position (overlaps.c:54-119; nothing is written back, a launch is
//                    idempotent), then the getStrand/updateMetrics/isVariant arithmetic (common.c:118-134,
//                    extract.c:225-239,420-441) and an LDS atomic;
//                 3. the tile is compacted from LDS with wave ballots and written as 16-byte site records into
//                    a segment reserved with one atomic per tile.
//               The workgroup keeps 256 reads in flight and needs ~13 KiB of LDS, so 8 workgroups (32 wavefronts)
//               fit a CU and the per-read latency chain (header -> bases) is hidden by occupancy.
//               (Measured and rejected, see DESIGN.md: a wavefront per read -- 192 us on S1, latency bound; staging
//               each tile's read payload in LDS with LDS-DMA -- 43-62 us, 70 KiB per workgroup kills occupancy.)
//
// Integer/byte work, HBM-bound: no MFMA anywhere (DESIGN.md has the roofline accounting).
#include <sys/mman.h>
#include <atomic>
#include <mutex>
#include <algorithm>
#include <type_traits>
#include <condition_variable>
#include <thread>
#include <chrono>
#include "mdk_hip_internal.hpp"

#define DEFAULT_TILE 2048
#define PERMAX 4                   // reference positions owned by one thread: tile = WG * per, per <= PERMAX
#define MAX_TILE (WG * PERMAX)
#define LDS_LIMIT 163840          // 160 KiB per CU / per workgroup on gfx950

static thread_local char g_err[MDK_ERR_BYTES] = "";
char *mdk_err_buf() { return g_err; }
int fail(int code, const char *what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    return code;
}

// ---- MDK_HOST_PROFILE: seconds the calling threads spend inside the library, per site ----
static std::atomic<uint64_t> g_prof_ns[PF_N], g_prof_calls[PF_N];
bool mdk_prof_on() { static const bool on = getenv("MDK_HOST_PROFILE") != nullptr; return on; }
double mdk_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void mdk_prof_add(int site, double seconds) { g_prof_ns[site].fetch_add((uint64_t)(seconds * 1e9), std::memory_order_relaxed); g_prof_calls[site].fetch_add(1, std::memory_order_relaxed); }
extern "C" int md_dev_profile_text(char *buf, int cap) {
    static const char *const name[PF_N] = {"upload:wait-for-slot", "upload:alloc", "upload:copies", "launch", "collect:wait", "download:copy", "download:order", "set_reference", "piece:submit", "piece:wait", "group:first-kernel-to-last-on-the-device", "group:launch-call-to-results-on-the-host"};
    int o = 0;
    if(!buf || cap < 1) return MDK_ERR_ARG;
    buf[0] = 0;
    for(int i = 0; i < PF_N && o < cap - 1; i++) o += snprintf(buf + o, (size_t)(cap - o), "%s%s %.3fs/%llu", i ? ", " : "", name[i], g_prof_ns[i].load() * 1e-9, (unsigned long long)g_prof_calls[i].load());
    return 0;
}

// ---- carved device memory (DBuf, mdk_hip_internal.hpp) ----
struct Arena { std::mutex mu; std::vector<char *> blocks; size_t cur = 0, used = 0; long live = 0; };       // cur: block being carved; used: bytes of it taken
static Arena g_arena[16];
static std::atomic<int> g_open_handles{0};       // carved memory starts over only when nothing is carved AND no handle is open (a handle's buffers come and go; its status arrays stay)
static const bool g_arena_on = getenv("MDK_NO_ARENA") == nullptr;
size_t mdk_arena_max() { static const size_t v = getenv("MDK_ARENA_MAX_MB") && atol(getenv("MDK_ARENA_MAX_MB")) > 0 ? (size_t)atol(getenv("MDK_ARENA_MAX_MB")) << 20 : ARENA_MAX; return v; }
void *arena_take(size_t bytes) {
    int dev = 0;
    if(!g_arena_on || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Arena &A = g_arena[dev];
    bytes = (bytes + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lk(A.mu);
    for(;;) {
        if(A.cur < A.blocks.size() && A.used + bytes <= ARENA_BLOCK) { char *p = A.blocks[A.cur] + A.used; A.used += bytes; A.live++; return p; }
        if(A.cur + 1 < A.blocks.size()) { A.cur++; A.used = 0; continue; }
        char *b = nullptr;
        MarkScope mk("arena hipMalloc 1 GiB");
        if(hipMalloc((void **)&b, ARENA_BLOCK) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        A.blocks.push_back(b); A.cur = A.blocks.size() - 1; A.used = 0;
    }
}
void arena_give(void *p) {
    for(Arena &A : g_arena) {
        std::lock_guard<std::mutex> lk(A.mu);
        for(char *b : A.blocks) if((char *)p >= b && (char *)p < b + ARENA_BLOCK) {
            if(--A.live == 0 && g_open_handles.load() == 0) { A.cur = 0; A.used = 0; while(A.blocks.size() > 2) { (void)hipFree(A.blocks.back()); A.blocks.pop_back(); } }      // nothing carved is in use and nobody could carve next to a reset: start over (a process that opens many handles in turn)
            return;
        }
    }
}
struct CallMark { std::atomic<const char *> what{nullptr}; std::atomic<double> t0{0}; };
static CallMark g_marks[256]; static std::atomic<int> g_mark_next{0};
static thread_local int t_mark_base = -1, t_mark_depth = 0;
int mdk_mark_begin(const char *what) {
    if(!mdk_prof_on()) return -1;
    if(t_mark_base < 0) t_mark_base = (g_mark_next.fetch_add(1) & 63) * 4;
    const int s = t_mark_base + (t_mark_depth < 3 ? t_mark_depth : 3); t_mark_depth++;
    g_marks[s].t0.store(mdk_now()); g_marks[s].what.store(what);
    return s;
}
void mdk_mark_end(int slot) { if(slot < 0) return; g_marks[slot].what.store(nullptr); if(t_mark_depth > 0) t_mark_depth--; }
void mdk_marks_dump(const char *why) {
    const double t = mdk_now(); char buf[1024]; int o = 0;
    for(int i = 0; i < 256 && o < 900; i++) { const char *w = g_marks[i].what.load(); if(w) o += snprintf(buf + o, sizeof(buf) - (size_t)o, " [%s for %.1f ms]", w, (t - g_marks[i].t0.load()) * 1e3); }
    fprintf(stderr, "[mdk hip] %s; other threads inside:%s\n", why, o ? buf : " nobody");
}
// ... and the same for pinned host memory (HBuf); pinned memory belongs to no device
struct HArena { std::mutex mu; std::vector<char *> blocks; size_t cur = 0, used = 0; long live = 0; };
static HArena g_harena;
void *harena_take(size_t bytes) {
    if(!g_arena_on) return nullptr;
    HArena &A = g_harena;
    bytes = (bytes + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lk(A.mu);
    for(;;) {
        if(A.cur < A.blocks.size() && A.used + bytes <= HARENA_BLOCK) { char *p = A.blocks[A.cur] + A.used; A.used += bytes; A.live++; return p; }
        if(A.cur + 1 < A.blocks.size()) { A.cur++; A.used = 0; continue; }
        char *b = nullptr;
        MarkScope mk("harena hipHostMalloc 32 MiB");
        if(hipHostMalloc((void **)&b, HARENA_BLOCK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        A.blocks.push_back(b); A.cur = A.blocks.size() - 1; A.used = 0;
    }
}
static void harena_reserve(size_t n_blocks) {
    HArena &A = g_harena;
    if(!g_arena_on) return;
    for(;;) {
        { std::lock_guard<std::mutex> lk(A.mu); if(A.blocks.size() >= n_blocks) return; }
        char *b = nullptr;
        MarkScope mk("harena_reserve hipHostMalloc 32 MiB");
        if(hipHostMalloc((void **)&b, HARENA_BLOCK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
        std::lock_guard<std::mutex> lk(A.mu); A.blocks.push_back(b);
    }
}
void harena_give(void *p) {
    HArena &A = g_harena;
    std::lock_guard<std::mutex> lk(A.mu);
    for(char *b : A.blocks) if((char *)p >= b && (char *)p < b + HARENA_BLOCK) {
        if(--A.live == 0 && g_open_handles.load() == 0) { A.cur = 0; A.used = 0; while(A.blocks.size() > 2) { (void)hipHostFree(A.blocks.back()); A.blocks.pop_back(); } }
        return;
    }
}
static void arena_reserve(size_t n_blocks) {      // blocks made ahead of their use (md_dev_warm's side thread): a hipMalloc inside arena_take holds every other taker up
    int dev = 0;
    if(!g_arena_on || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
    Arena &A = g_arena[dev];
    for(;;) {
        { std::lock_guard<std::mutex> lk(A.mu); if(A.blocks.size() >= n_blocks) return; }
        char *b = nullptr;
        MarkScope mk("arena_reserve hipMalloc 1 GiB");
        if(hipMalloc((void **)&b, ARENA_BLOCK) != hipSuccess) { (void)hipGetLastError(); return; }
        std::lock_guard<std::mutex> lk(A.mu); A.blocks.push_back(b);
    }
}
static std::atomic<uint64_t> g_reserve_hint{0};
static size_t reserve_blocks_wanted() { const uint64_t b = g_reserve_hint.load(); size_t n = (size_t)((b + ARENA_BLOCK - 1) / ARENA_BLOCK); if(n < 4) n = 4; if(n > 40) n = 40; return n; }      // (24 slots of a 30x 1 Mb chunk each carve ~2.5 GiB)
static std::atomic<int> g_reserve_done{0};      // the warm-up's helper thread has made its blocks: a hint that comes later is acted on by its caller's own helper
template <typename F> static void side_start(F &&f);
extern "C" void md_dev_reserve_hint(uint64_t device_bytes) {
    g_reserve_hint.store(device_bytes);
    if(g_reserve_done.load()) { int dev = 0; if(hipGetDevice(&dev) == hipSuccess) side_start([dev]() { if(hipSetDevice(dev) == hipSuccess) arena_reserve(reserve_blocks_wanted()); (void)hipGetLastError(); }); }
}
static void arena_prime(int device) {          // the first block, while the caller is still starting up (md_dev_warm)
    (void)device;
    void *p = arena_take(256); if(p) arena_give(p);
    p = harena_take(256); if(p) harena_give(p);
}

// ------------------------------------------------------------------------------------------------
// kernel parameters
// ------------------------------------------------------------------------------------------------

struct KParams {
    const md_seg *seg; const uint8_t *blob;
    const uint8_t *ctxcode; int64_t reflen;
    int64_t beg, end; int tile, ntiles, nper;
    const TileEnt *tiles;
    md_site *site; md_site_var *var; md_tile_seg *tseg; uint32_t *total, *total_next; int64_t cap_sites;
    int keepmask, minPhred;
    int bounds[16], abounds[16];
    int *err;
    int packed;                   // 0: payload = blob + 4*off4, qualities after the sequence padded to 4 bytes (host-built batches);
                                  // 1: payload = blob + off4 (byte offset into the uploaded BAM records), qualities directly after the sequence
    int mbias; uint32_t *hist; int hist_lq;     // mbias: window-relative contexts at the chunk edges; histogram rows [q][16]; rows kept in LDS
};

// kept query-index window [lo,hi) of a read after --OT-style and --nOT-style trimming (common.c:137-208)
__device__ __forceinline__ void trim_window(const KParams &P, int strand, int read2, int lq, int &lo, int &hi) {
    if(strand < 1) { lo = 0; hi = lq; return; }
    int b = 4 * (strand - 1) + (read2 ? 2 : 0);
    int lb = P.bounds[b], rb = P.bounds[b + 1];
    int alb = P.abounds[b], arb = P.abounds[b + 1];
    lb = lb < lq ? lb : lq; alb = alb < lq ? alb : lq; arb = arb < lq ? arb : lq;
    lo = lb > alb ? lb : alb;
    hi = (rb && rb < lq) ? rb : lq;
    if(lq - arb < hi) hi = lq - arb;
}

struct RD {       // addressing of one read's payload + its trimming window
    int lq, lo, hi;
    const uint8_t *seq, *qual;
};
__device__ __forceinline__ RD make_rd(const KParams &P, uint32_t off4, uint32_t lq, int strand, int read2) {
    RD d; d.lq = (int)lq;
    if(P.packed) { d.seq = P.blob + off4; d.qual = d.seq + ((d.lq + 1) >> 1); }
    else { d.seq = P.blob + 4ull * off4; d.qual = d.seq + ((((d.lq + 1) >> 1) + 3) & ~3); }
    trim_window(P, strand, read2, d.lq, d.lo, d.hi);
    return d;
}

#include "mdk_overlap_rule.h"      // boost, resolve_overlap (the literal rule), resolve_own (the same as selects)

// context code of a reference position: 0 = not C/G, else 1 + 2*type + isG (type 0 CpG, 1 CHG, 2 CHH).
// (x & 0x5f) folds 'c'->'C', 'g'->'G' and maps no other FASTA letter onto C or G.
__global__ __launch_bounds__(WG) void k_classify(const char *ref, uint8_t *code, int64_t n) {
    for(int64_t p = (int64_t)blockIdx.x * WG + threadIdx.x; p < n; p += (int64_t)gridDim.x * WG) {
        char c = ref[p] & 0x5f; int type, isG, out = 0;
        if(c == 'C') {
            isG = 0;
            if(p + 1 < n && (ref[p + 1] & 0x5f) == 'G') type = 0;
            else if(p + 2 < n && (ref[p + 2] & 0x5f) == 'G') type = 1;
            else type = 2;
            out = 1 + 2 * type + isG;
        } else if(c == 'G') {
            isG = 1;
            if(p > 0 && (ref[p - 1] & 0x5f) == 'C') type = 0;
            else if(p > 1 && (ref[p - 2] & 0x5f) == 'C') type = 1;
            else type = 2;
            out = 1 + 2 * type + isG;
        }
        code[p] = (uint8_t)out;
    }
}

// -l/--keepStrand (bed.c:46-64, extract.c:402-405,425): restrict a contig's sites to sorted disjoint runs.  A position
// outside every run stops being a site (code 0); inside a run the run's strand code (0 any, 1 '+', 2 '-') goes into
// bits 4-5 of the code, where the pileup reads it per position.  One thread per position, binary search over the runs.
__global__ __launch_bounds__(WG) void k_mask_regions(uint8_t *code, int64_t n, const md_region *runs, int64_t nruns) {
    for(int64_t p = (int64_t)blockIdx.x * WG + threadIdx.x; p < n; p += (int64_t)gridDim.x * WG) {
        const int c = code[p] & 15;
        if(!c) { code[p] = 0; continue; }
        int64_t a = 0, b = nruns;                     // first run with end > p
        while(a < b) { const int64_t m = (a + b) >> 1; if((int64_t)runs[m].end <= p) a = m + 1; else b = m; }
        code[p] = (a < nruns && (int64_t)runs[a].start <= p) ? (uint8_t)(c | ((runs[a].strand & 3) << 4)) : (uint8_t)0;
    }
}

#define KB 2      // positions per batch: their base/qual bytes (own + partner) are all in flight together

// One segment, one lane.
template <bool VARIANT>
__device__ __forceinline__ void lane_seg(const KParams &P, const md_seg &g, int T0, int T1,
                                         const uint16_t *listC, int nC, const uint16_t *listG, int nG,
                                         uint32_t *cm, uint32_t *cu, uint32_t *co, uint32_t *cv) {
    const int send = g.rpos + (int)g.len;
    if(g.rpos >= T1 || send <= T0) return;                        // inside the tile's run but not on the tile
    const int strand = g.sf & MDK_SF_STRAND;
    const bool odd = strand & 1, second = (g.sf & MDK_SF_SECOND) != 0, partner = (g.sf & MDK_SF_PARTNER) != 0;
    const RD o = make_rd(P, g.off4, g.l_qseq, strand, g.sf & MDK_SF_READ2);
    RD m = o;
    if(partner) m = make_rd(P, g.m_off4, g.m_l_qseq, g.msf & MDK_SF_STRAND, g.msf & MDK_SF_READ2);
    const int lo_off = g.rpos > T0 ? g.rpos - T0 : 0, hi_off = (send < T1 ? send : T1) - T0;   // tile offsets covered
    // --keepStrand: region strand codes (bits 13-14 of a list entry) this read is invisible at (bed.c:56-64):
    // '+' regions (1) want OT/CTOT, '-' regions (2) want OB/CTOB, a read of unknown strand matches neither
    const int badrs = strand == 0 ? 6 : (odd ? 4 : 2);
#pragma unroll
    for(int pass = 0; pass < (VARIANT ? 2 : 1); pass++) {
        const bool callpass = pass == 0;
        const bool useC = (odd == callpass);      // calls: OT/CTOT on C, OB/CTOB on G; opposite-strand evidence: the other list
        const uint16_t *list = useC ? listC : listG; const int n = useC ? nC : nG;
        int a = 0, b = n;
        while(a < b) { int mid = (a + b) >> 1; if((int)(list[mid] & 0x1fff) < lo_off) a = mid + 1; else b = mid; }
        int i = a;
        while(i < n) {
            // 1. which positions (no base has been touched yet)
            int li[KB];
#pragma unroll
            for(int k = 0; k < KB; k++) {
                li[k] = -1;
                if(i < n) { const int e = list[i], l = e & 0x1fff; if(l < hi_off) { li[k] = ((badrs >> (e >> 13)) & 1) ? -2 : l; i++; } else i = n; }
            }
            if(li[0] == -1) break;
            // 2. request every byte the batch needs (trimmed bases need none: they read as N with quality 0)
            uint32_t sb[KB], qb[KB], msb[KB], mqb[KB];
#pragma unroll
            for(int k = 0; k < KB; k++) {
                sb[k] = 0xff; qb[k] = 0; msb[k] = 0xff; mqb[k] = 0;
                if(li[k] >= 0) {
                    const int d = T0 + li[k] - g.rpos, q = (int)g.q0 + d, mq = (int)g.m_q0 + d;
                    if(q >= o.lo && q < o.hi) { sb[k] = o.seq[q >> 1]; qb[k] = o.qual[q]; }
                    if(partner && mq >= m.lo && mq < m.hi) { msb[k] = m.seq[mq >> 1]; mqb[k] = m.qual[mq]; }
                }
            }
            // 3. use them
#pragma unroll
            for(int k = 0; k < KB; k++) {
                if(li[k] < 0) continue;
                const int d = T0 + li[k] - g.rpos, q = (int)g.q0 + d, mq = (int)g.m_q0 + d;
                int bq = (q & 1) ? (sb[k] & 15) : (sb[k] >> 4), ql = (int)qb[k];
                if(partner) { int mb = (mq & 1) ? (msb[k] & 15) : (msb[k] >> 4); ql = resolve_overlap(second, bq, ql, mb, (int)mqb[k]); }
                if(callpass) {
                    if(strand == 0) atomicExch(P.err, 1);                 // reference: assert(strand != 0) (common.c:122-125)
                    if(ql >= P.minPhred) {
                        if(odd) { if(bq == 2) atomicAdd(&cm[li[k]], 1u); else if(bq == 8) atomicAdd(&cu[li[k]], 1u); }
                        else { if(bq == 4) atomicAdd(&cm[li[k]], 1u); else if(bq == 1) atomicAdd(&cu[li[k]], 1u); }
                    }
                } else if(VARIANT) {
                    if(ql >= P.minPhred) {
                        atomicAdd(&co[li[k]], 1u);
                        if(odd ? (bq != 4 && bq != 15) : (bq != 2 && bq != 15)) atomicAdd(&cv[li[k]], 1u);
                    }
                }
            }
        }
    }
}

// ---- dense contexts (CHG/CHH counted, ~20 sites per segment): a quarter of a wavefront per segment --------------------------
// With a lane per segment every lane walks its own read, so a wavefront's byte loads touch 64 different reads (one L1 access
// per loaded byte) and the lanes run as long as the segment with the most sites.  Here the 64 lanes first prepare their 64
// segments (trimming windows, payload offsets, the [a,b) range of the tile's site list the segment covers), then the
// wavefront goes through them four at a time: 16 neighbouring lanes take 16 neighbouring sites of ONE read (parameters by
// ds_bpermute from the lane that prepared it), so their loads fall into the same one or two cache lines.
#ifndef PILEUP_WAVES
#define PILEUP_WAVES 8
#endif
#ifndef QW_WAVES
#define QW_WAVES 6   // waves per SIMD the dense-context kernel is compiled for (6: 72-78 VGPRs, no scratch, three workgroups per CU: 217.6 us per 8-chunk launch; 8: 64 VGPRs and 44-64 bytes of scratch with the two forms of the site loop: 219.9)
#endif
#ifndef QL
#define QL 8          // lanes per segment
#endif
#ifndef QU
#define QU 2          // sites per lane whose bytes are in flight together
#endif
struct SegQ {
    uint32_t oseq, oqual, mseq, mqual;        // payload byte offsets in the blob (uploads are < 4 GiB): sequence, qualities; own and partner
    int cq, lo, mcq, mlo; uint32_t wlen, mwlen;   // query index = tile offset + cq; kept window [lo, lo + wlen)
    uint32_t w[2];                            // per pass: strand | second << 3 | partner << 4 | first list entry << 5 | number of sites << 18
};

// first list entry whose tile offset (low 13 bits) is >= x; `top` = highest power of two <= n (0 for an empty list)
__device__ __forceinline__ int list_lower_bound(const uint16_t *list, int n, int top, int x) {
    int a = 0;
    for(int step = top; step; step >>= 1) { const int ia = a + step; if(ia <= n && (int)(list[ia - 1] & 0x1fff) < x) a = ia; }
    return a;
}

template <bool VARIANT>
__device__ __forceinline__ SegQ seg_setup(const KParams &P, const md_seg &g, int T0, int T1, const uint16_t *listC, int nC, int nG, int topC, int topG) {
    SegQ s; s.oseq = s.oqual = s.mseq = s.mqual = 0; s.cq = s.lo = s.mcq = s.mlo = 0; s.wlen = s.mwlen = 0; s.w[0] = s.w[1] = 0;
    const int send = g.rpos + (int)g.len;
    if(g.rpos >= T1 || send <= T0) return s;                      // no segment for this lane, or inside the tile's run but not on the tile
    const int strand = g.sf & MDK_SF_STRAND;
    const bool odd = strand & 1, second = (g.sf & MDK_SF_SECOND) != 0, partner = (g.sf & MDK_SF_PARTNER) != 0;
    int lo, hi;
    trim_window(P, strand, g.sf & MDK_SF_READ2, (int)g.l_qseq, lo, hi);
    s.oseq = P.packed ? g.off4 : g.off4 << 2;
    s.oqual = s.oseq + (P.packed ? (g.l_qseq + 1) >> 1 : (((g.l_qseq + 1) >> 1) + 3) & ~3u);
    s.lo = lo; s.wlen = hi > lo ? (unsigned)(hi - lo) : 0u;
    s.cq = T0 - g.rpos + (int)g.q0;
    if(partner) {
        int mlo, mhi;
        trim_window(P, g.msf & MDK_SF_STRAND, g.msf & MDK_SF_READ2, (int)g.m_l_qseq, mlo, mhi);
        s.mseq = P.packed ? g.m_off4 : g.m_off4 << 2;
        s.mqual = s.mseq + (P.packed ? (g.m_l_qseq + 1) >> 1 : (((g.m_l_qseq + 1) >> 1) + 3) & ~3u);
        s.mlo = mlo; s.mwlen = mhi > mlo ? (unsigned)(mhi - mlo) : 0u;
        s.mcq = T0 - g.rpos + (int)g.m_q0;
    }
    const int lo_off = g.rpos > T0 ? g.rpos - T0 : 0, hi_off = (send < T1 ? send : T1) - T0;   // tile offsets covered
#pragma unroll
    for(int pass = 0; pass < (VARIANT ? 2 : 1); pass++) {
        const bool useC = (odd == (pass == 0));   // calls: OT/CTOT on C, OB/CTOB on G; opposite-strand evidence: the other list
        const uint16_t *list = useC ? listC : listC + P.tile; const int n = useC ? nC : nG, top = useC ? topC : topG;
        const int a = list_lower_bound(list, n, top, lo_off), b = list_lower_bound(list, n, top, hi_off);
        s.w[pass] = (uint32_t)strand | (second ? 8u : 0u) | (partner ? 16u : 0u) | (uint32_t)((useC ? 0 : P.tile) + a) << 5 | (uint32_t)(b - a) << 18;
    }
    return s;
}

// The 64 segments a wavefront has prepared, dealt out again so that the eight a step works on (the lanes l with l & 7 == step) are of one
// kind: first the segments without a partner, then those with one, last those with no site on this tile (their steps are skipped whole).
// Rank s goes to lane 8 (s & 7) + (s >> 3); ds_permute pushes every field there.
__device__ __forceinline__ SegQ seg_sort(const SegQ &q, const int lane) {
    const bool some = ((q.w[0] | q.w[1]) >> 18) != 0, partner = (q.w[0] >> 4) & 1;
    const unsigned long long below = (1ull << lane) - 1ull, ma = __ballot(some && !partner), mb = __ballot(some && partner), mc = ~(ma | mb);
    const int s = some ? (partner ? __popcll(ma) + __popcll(mb & below) : __popcll(ma & below)) : __popcll(ma) + __popcll(mb) + __popcll(mc & below);
    const int to = (8 * (s & 7) + (s >> 3)) << 2;
    SegQ r;
    r.oseq = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.oseq); r.oqual = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.oqual);
    r.mseq = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.mseq); r.mqual = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.mqual);
    r.cq = __builtin_amdgcn_ds_permute(to, q.cq); r.lo = __builtin_amdgcn_ds_permute(to, q.lo); r.mcq = __builtin_amdgcn_ds_permute(to, q.mcq); r.mlo = __builtin_amdgcn_ds_permute(to, q.mlo);
    r.wlen = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.wlen); r.mwlen = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.mwlen);
    r.w[0] = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.w[0]); r.w[1] = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)q.w[1]);
    return r;
}
template <bool VARIANT>
__device__ __forceinline__ void quarter_sites(const KParams &P, const SegQ &s, int lane, const uint16_t *listC,
                                              uint32_t *cm, uint32_t *cu, uint32_t *co, uint32_t *cv) {
    const int sub = lane & (QL - 1), qbase = lane & (64 - QL);
    const uint8_t *const blob = P.blob; const int minPhred = P.minPhred, tile = P.tile;
#pragma unroll
    for(int pass = 0; pass < (VARIANT ? 2 : 1); pass++) {
        const bool callpass = pass == 0;
        if(__ballot((s.w[pass] >> 18) != 0) == 0) continue;       // no lane of this wavefront has a site in this pass
        for(int it = 0; it < QL; it++) {
            const int owner = qbase | it;                          // the lane that prepared the segment this quarter works on now
            const uint32_t w = (uint32_t)__shfl((int)s.w[pass], owner);
            const int n = (int)(w >> 18);
            if(__ballot(n != 0) == 0) continue;
            const uint32_t oseq = (uint32_t)__shfl((int)s.oseq, owner), oqual = (uint32_t)__shfl((int)s.oqual, owner), wlen = (uint32_t)__shfl((int)s.wlen, owner);
            const int cq = __shfl(s.cq, owner), lo = __shfl(s.lo, owner);
#ifdef QW_EXP_NO_PARTNER                    // TIMING EXPERIMENT ONLY (wrong results): no segment has a partner
            const bool partner = false;
#else
            const bool partner = (w >> 4) & 1;
#endif
            uint32_t mseq = 0, mqual = 0, mwlen = 0; int mcq = 0, mlo = 0;
            const bool anyp = __ballot(partner && n != 0) != 0;
            if(anyp) {
                mseq = (uint32_t)__shfl((int)s.mseq, owner); mqual = (uint32_t)__shfl((int)s.mqual, owner); mwlen = (uint32_t)__shfl((int)s.mwlen, owner);
                mcq = __shfl(s.mcq, owner); mlo = __shfl(s.mlo, owner);
            }
            const int strand = w & 7; const bool odd = strand & 1, second = (w >> 3) & 1;
            // --keepStrand: region strand codes (bits 13-14 of a list entry) this read is invisible at (bed.c:56-64):
            // '+' regions (1) want OT/CTOT, '-' regions (2) want OB/CTOB, a read of unknown strand matches neither
            const int badrs = strand == 0 ? 6 : (odd ? 4 : 2);
            const uint16_t *list = listC + ((w >> 5) & 0x1fff);
            const int mcode = odd ? 2 : 4, ucode = odd ? 8 : 1;      // the strand's C read as C (methylated) / as T (unmethylated): BAM codes C=2 T=8, G=4 A=1
            // a lane takes two ADJACENT entries of the list per step (one LDS load), the 8 lanes of a segment 16 consecutive sites
            bool anyok = false;
            // the loop over the segment's sites, once with and once without the partner's half (the segments of a round are sorted so that
            // the eight of a step nearly always agree, seg_sort: a step of partner-less segments asks for half the bytes and runs no overlap rule)
            auto sites = [&](auto has_partner) {
            constexpr bool HP = decltype(has_partner)::value;
            for(int j = 2 * sub; j < n; j += 2 * QL) {
                // 1. which sites (no base has been touched yet); 2. every byte they need is requested; 3. they are used
                uint32_t e01; __builtin_memcpy(&e01, list + j, 4);             // (the second entry is whatever follows the list when j + 1 == n)
                const uint32_t e0 = e01 & 0xffffu, e1 = e01 >> 16;
                const int l0 = (int)(e0 & 0x1fffu), l1 = (int)(e1 & 0x1fffu);
                const bool ok0 = !((badrs >> (e0 >> 13)) & 1), ok1 = j + 1 < n && !((badrs >> (e1 >> 13)) & 1);
                const int q0 = l0 + cq, q1 = l1 + cq, mq0 = l0 + mcq, mq1 = l1 + mcq;
                uint32_t sb0 = 0xff, qb0 = 0, msb0 = 0xff, mqb0 = 0, sb1 = 0xff, qb1 = 0, msb1 = 0xff, mqb1 = 0;   // a trimmed base needs no load: it reads as N with quality 0
                if(ok0 && (unsigned)(q0 - lo) < wlen) { sb0 = blob[oseq + (uint32_t)(q0 >> 1)]; qb0 = blob[oqual + (uint32_t)q0]; }
                if(HP && ok0 && partner && (unsigned)(mq0 - mlo) < mwlen) { msb0 = blob[mseq + (uint32_t)(mq0 >> 1)]; mqb0 = blob[mqual + (uint32_t)mq0]; }
                if(ok1 && (unsigned)(q1 - lo) < wlen) { sb1 = blob[oseq + (uint32_t)(q1 >> 1)]; qb1 = blob[oqual + (uint32_t)q1]; }
                if(HP && ok1 && partner && (unsigned)(mq1 - mlo) < mwlen) { msb1 = blob[mseq + (uint32_t)(mq1 >> 1)]; mqb1 = blob[mqual + (uint32_t)mq1]; }
                auto use = [&](const bool ok, const int l, const int q, const int mq, const uint32_t sb, const uint32_t qb, const uint32_t msb, const uint32_t mqb) {
                    if(!ok) return;
                    const int bq = (int)((sb >> ((~q & 1) << 2)) & 15u); int ql = (int)qb;
                    if(HP && partner) { const int mb = (int)((msb >> ((~mq & 1) << 2)) & 15u); ql = resolve_own(second, bq, ql, mb, (int)mqb); }
                    if(callpass) {
                        if(ql >= minPhred && (bq == mcode || bq == ucode)) atomicAdd(&cm[(bq == ucode ? tile : 0) + l], 1u);     // cu = cm + tile
                    } else if(VARIANT) {
                        if(ql >= minPhred) {
                            atomicAdd(&co[l], 1u);
                            if(odd ? (bq != 4 && bq != 15) : (bq != 2 && bq != 15)) atomicAdd(&cv[l], 1u);
                        }
                    }
                };
                use(ok0, l0, q0, mq0, sb0, qb0, msb0, mqb0);
                use(ok1, l1, q1, mq1, sb1, qb1, msb1, mqb1);
                anyok = anyok || ok0 || ok1;
            }
            };
            if(anyp) sites(std::true_type{}); else sites(std::false_type{});
            if(callpass && strand == 0 && anyok) atomicExch(P.err, 1);        // a read of unknown strand reached a call; reference: assert(strand != 0) (common.c:122-125)
        }
    }
}

// barrier that orders LDS traffic only (does not drain this wave's outstanding global loads)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive scan over the 64 lanes of a wavefront.  The lane number goes through an empty asm first: the tile routine scans twice, long
// before and long after its main loop, and the six shuffle addresses of the first scan would otherwise be kept for the second -- in
// scratch, under the 64-VGPR cap: 24 bytes written and read back per thread, 100 MB per launch of 8 chunks.
__device__ __forceinline__ int wave_scan_incl(int v, int lane) {
    asm volatile("" : "+v"(lane));
#pragma unroll
    for(int d = 1; d < 64; d <<= 1) { const int n = __builtin_amdgcn_ds_bpermute((lane >= d ? lane - d : lane) << 2, v); if(lane >= d) v += n; }
    return v;
}


// context codes of the PER consecutive positions a thread owns (0 = not a site or context not wanted).
// mbias classifies inside the chunk's own reference window [beg, end] (MBias.c:147,172-180), so a G in the first two
// positions cannot see the C before the chunk and a C in the last position cannot see a G two past it.
__device__ __forceinline__ void load_codes(const KParams &P, int64_t T0, int tlen, int PER, int tid, int (&code)[PERMAX]) {
#pragma unroll
    for(int j = 0; j < PERMAX; j++) {
        const int i = tid * PER + j; const int64_t p = T0 + i;
        code[j] = (j < PER && i < tlen && p < P.reflen) ? P.ctxcode[p] : 0;
    }
#pragma unroll
    for(int j = 0; j < PERMAX; j++) {
        if(j < PER && code[j]) {
            if(P.mbias) {
                const int64_t p = T0 + tid * PER + j; const int c = (code[j] & 15) - 1, type = c >> 1, isG = c & 1;
                int nt = type;
                if(isG) { if(p == P.beg) nt = 2; else if(p == P.beg + 1 && type == 1) nt = 2; }
                else if(p == P.end - 1 && type == 1) nt = 2;
                code[j] = (code[j] & ~15) | (1 + 2 * nt + isG);
            }
            if(!((P.keepmask >> (((code[j] & 15) - 1) >> 1)) & 1)) code[j] = 0;
        }
    }
}

// sorted C and G position lists of the tile in LDS (wave scan + one cross-wave exchange): entry = tile offset | region strand code << 13
__device__ __forceinline__ void build_lists(const KParams &P, int PER, int tid, int lane, int wave, const int (&code)[PERMAX],
                                            uint16_t *listC, uint16_t *listG, int *wsum, int &nC, int &nG) {
    int cntC = 0, cntG = 0;
#pragma unroll
    for(int j = 0; j < PERMAX; j++)
        if(j < PER && code[j]) { if((code[j] - 1) & 1) cntG++; else cntC++; }      // bit 0 of (code-1) = isG (bits 4-5 do not reach it)
    const int packed = cntC | (cntG << 16), incl = wave_scan_incl(packed, lane);
    if(lane == 63) wsum[wave] = incl;
    lds_barrier();
    int pre = incl - packed, tot = 0;
    for(int w = 0; w < WAVES; w++) { const int c = wsum[w]; if(w < wave) pre += c; tot += c; }
    nC = tot & 0xffff; nG = tot >> 16;
    int oc = pre & 0xffff, og = pre >> 16;
#pragma unroll
    for(int j = 0; j < PERMAX; j++) {
        if(j < PER && code[j]) {
            const uint16_t ent = (uint16_t)((tid * PER + j) | ((code[j] >> 4) << 13));
            if((code[j] - 1) & 1) listG[og++] = ent; else listC[oc++] = ent;
        }
    }
    lds_barrier();
}

// one workgroup, one tile `t` of the interval P describes (b: the workgroup's index in the launch, for the phase profile only)
template <bool VARIANT, bool QW>
__device__ __forceinline__ void pileup_tile(const KParams &P, const int t, const int b) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int TILE = P.tile, PER = TILE / WG;
    uint32_t *cm = lds, *cu = lds + TILE, *co = lds + 2 * TILE, *cv = lds + 3 * TILE;
    uint16_t *listC = (uint16_t *)(lds + (VARIANT ? 4 : 2) * TILE), *listG = listC + TILE;
    __shared__ int wsum[WAVES];
    __shared__ uint32_t sbase;

    const int64_t T0 = P.beg + (int64_t)t * TILE;
    const int64_t T1 = (T0 + TILE < P.end) ? T0 + TILE : P.end;
    const int tlen = (int)(T1 - T0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if(t == 0 && tid == 0) *P.total_next = 0;    // the counter the NEXT launch of this slot will use

    // everything this thread needs from HBM before it can start is requested up front, in one round:
    // the context codes of the PER consecutive positions it owns, and its first segment record
    const TileEnt te = P.tiles[t];
    const int first = te.last > te.first ? te.first : 0, last = te.last > te.first ? te.last : 0;      // a tile no segment touches: (INT_MAX, 0) from the device preparation
    int code[PERMAX];
    load_codes(P, T0, tlen, PER, tid, code);
    md_seg g0; g0.rpos = 0x7fffffff; g0.len = 0;
    if(first + tid < last) g0 = P.seg[first + tid];

    // phase 1: sorted C / G position lists (wave scan + one cross-wave exchange), counters zeroed
#pragma unroll
    for(int j = 0; j < PERMAX; j++) {
        if(j < PER) {
            const int i = tid * PER + j;
            cm[i] = 0; cu[i] = 0;
            if(VARIANT) { co[i] = 0; cv[i] = 0; }
        }
    }
    int nC, nG;
    build_lists(P, PER, tid, lane, wave, code, listC, listG, wsum, nC, nG);

    // the context codes are needed again in phase 3: they cross phase 2 as ONE register (a byte each), not four
    uint32_t codes = 0;
#pragma unroll
    for(int j = 0; j < PERMAX; j++) codes |= (uint32_t)(code[j] & 0xff) << (8 * j);
    asm volatile("" : "+v"(codes));

    // phase 2: WG segments per round.  One segment per lane, or (QW) prepared by one lane each and worked on by 16
    if constexpr(QW) {
        const int topC = nC ? 1 << (31 - __clz(nC)) : 0, topG = nG ? 1 << (31 - __clz(nG)) : 0;
        for(int r0 = first; r0 < last; r0 += WG) {                 // uniform: every lane takes part in the exchange of every round
            md_seg g = g0;
            if(r0 != first) { g.rpos = 0x7fffffff; g.len = 0; if(r0 + tid < last) g = P.seg[r0 + tid]; }
#ifndef QW_NO_SORT
            const SegQ sq = seg_sort(seg_setup<VARIANT>(P, g, (int)T0, (int)T1, listC, nC, nG, topC, topG), lane);
#else
            const SegQ sq = seg_setup<VARIANT>(P, g, (int)T0, (int)T1, listC, nC, nG, topC, topG);
#endif
            quarter_sites<VARIANT>(P, sq, lane, listC, cm, cu, co, cv);
        }
    } else {
        if(first + tid < last) lane_seg<VARIANT>(P, g0, (int)T0, (int)T1, listC, nC, listG, nG, cm, cu, co, cv);
        for(int r = first + WG + tid; r < last; r += WG) {
            const md_seg g = P.seg[r];
            lane_seg<VARIANT>(P, g, (int)T0, (int)T1, listC, nC, listG, nG, cm, cu, co, cv);
        }
    }
    // reserve this tile's output segment: at most one site per kept context position (unused slots stay empty,
    // md_tile_seg.cnt says how many are filled).  Issued by the first thread once its own segments are done, so the
    // atomic's round trip overlaps the wait for the slower wavefronts, and the tiles' atomics are spread in time.
    uint32_t reserved = 0;
    if(tid == 0) reserved = (nC + nG) ? atomicAdd(P.total, (uint32_t)(nC + nG)) : 0u;
    __syncthreads();

    // phase 3: compaction.  Every thread packs the positions it owns; one atomic reserves the tile's segment;
    // 16-byte site records are written in ascending position order.
#pragma unroll
    for(int j = 0; j < PERMAX; j++) code[j] = (int)((codes >> (8 * j)) & 0xffu);
    uint32_t vm[PERMAX], vu[PERMAX], vo[PERMAX], vv[PERMAX]; int cnt = 0;
#pragma unroll
    for(int j = 0; j < PERMAX; j++) {
        vm[j] = vu[j] = vo[j] = vv[j] = 0;
        const int i = tid * PER + j;
        if(j < PER && i < tlen && code[j]) { vm[j] = cm[i]; vu[j] = cu[i]; if(VARIANT) { vo[j] = co[i]; vv[j] = cv[i]; } }
        if((vm[j] + vu[j]) > 0 || vo[j] > 0) cnt++;
    }
    const int incl = wave_scan_incl(cnt, lane);
    if(lane == 63) wsum[wave] = incl;
    if(tid == 0) sbase = reserved;
    __syncthreads();
    if(tid == 0) {
        int tot = 0; for(int w = 0; w < WAVES; w++) tot += wsum[w];
        md_tile_seg sg; sg.off = reserved; sg.cnt = (uint32_t)tot; P.tseg[t] = sg;
    }
    {
        uint32_t o = sbase + (uint32_t)(incl - cnt);
        for(int w = 0; w < wave; w++) o += (uint32_t)wsum[w];
#pragma unroll
        for(int j = 0; j < PERMAX; j++) {
            if((vm[j] + vu[j]) > 0 || vo[j] > 0) {
                if((int64_t)o < P.cap_sites) {
                    md_site rec; rec.pos = (uint32_t)(T0 + tid * PER + j); rec.nmeth = vm[j]; rec.nunmeth = vu[j]; rec.meta = (uint32_t)((code[j] & 15) - 1);
                    P.site[o] = rec;
                    if(VARIANT) { md_site_var rv; rv.noff = vo[j]; rv.nvar = vv[j]; P.var[o] = rv; }
                }
                o++;
            }
        }
    }
}

template <bool VARIANT, bool QW>
__global__ __launch_bounds__(WG, QW ? QW_WAVES : PILEUP_WAVES) void k_pileup(const KParams P) {     // 64 VGPRs: four 512-thread workgroups per CU
    const int b = blockIdx.x;
    const int t = (b & 7) * P.nper + (b >> 3);   // XCD-aware: workgroup b runs on XCD b%8; give each XCD a contiguous run of tiles
    if(t >= P.ntiles) return;
    pileup_tile<VARIANT, QW>(P, t, b);
}

// Several intervals (chunks of the reference's schedule, each with its own reads, outputs and site counter) in ONE launch:
// a 1 Mb chunk is 489 tiles -- fewer than two workgroups per CU, one short generation whose dispatch ramp and barrier tail
// are a third of its time -- so resident chunks are launched MAXM at a time.  The per-interval parameters travel in the
// kernel arguments; a workgroup finds its interval from the tile prefix.
struct KMulti { int n, nper; int tstart[MAXM + 1]; KParams P[MAXM]; };
template <bool VARIANT, bool QW>
__global__ __launch_bounds__(WG, QW ? QW_WAVES : PILEUP_WAVES) void k_pileup_multi(const KMulti M) {
    const int b = blockIdx.x;
    const int tg = (b & 7) * M.nper + (b >> 3);
    if(tg >= M.tstart[M.n]) return;
    int j = 0;
    while(j + 1 < M.n && tg >= M.tstart[j + 1]) j++;
    pileup_tile<VARIANT, QW>(M.P[j], tg - M.tstart[j], b);
}

// The results of a group launch, put where the host reads them by the device itself: every tile's run of site records is copied to
// its place in position order (the prefix sum over the tiles' counts: what md_sites_order does on the host) straight into the slot's
// pinned host buffer, and the slot's status block into the pinned mirror.  The collector then waits for the stream and has everything;
// the copies it used to queue -- status, then sites, variant counters and tile table of each of eight slots, SDMA and blit kernels by
// turns on one stream, each hop behind whatever k_inflate had on the device -- took it 5.4 ms per group, three quarters of the streaming
// phase of a 512 Mb run (profiles/r06i_*).  SlotStatus.pad tells the host what it got: the number of ordered sites, or PACK_NONE when
// they did not fit the host buffer (or the tile table is inconsistent): that slot is collected by copies as before.
#define PACK_WG 256
#define PACK_BLOCKS 8              // workgroups per slot, each copies every eighth tile
#define PACK_NONE 0xffffffffu
struct KPackSlot { const md_site *site; const md_site_var *var; const md_tile_seg *tseg; md_site *out; md_site_var *vout; const SlotStatus *d_st; SlotStatus *h_st; int ntiles; uint32_t cap, site_cap; };
struct KPack { int n; KPackSlot S[MAXM]; };
__global__ __launch_bounds__(PACK_WG) void k_sites_pack(const KPack K) {
    const int j = blockIdx.x / PACK_BLOCKS, b = blockIdx.x % PACK_BLOCKS;
    const KPackSlot &S = K.S[j];
    __shared__ uint32_t e_off[PACK_WG], e_cnt[PACK_WG], e_dst[PACK_WG], wsum[PACK_WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t base = 0; bool bad = false;
    for(int r0 = 0; r0 < S.ntiles; r0 += PACK_WG) {
        md_tile_seg sg; sg.off = 0; sg.cnt = 0;
        if(r0 + tid < S.ntiles) sg = S.tseg[r0 + tid];
        if((uint64_t)sg.off + sg.cnt > S.site_cap) { bad = true; sg.cnt = 0; }
        const uint32_t incl = (uint32_t)wave_scan_incl((int)sg.cnt, lane);
        if(lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t o = base + incl - sg.cnt, tot = 0;
        for(int w = 0; w < PACK_WG / 64; w++) { if(w < wave) o += wsum[w]; tot += wsum[w]; }
        e_off[tid] = sg.off; e_cnt[tid] = sg.cnt; e_dst[tid] = o;
        __syncthreads();
        const int ne = S.ntiles - r0 < PACK_WG ? S.ntiles - r0 : PACK_WG;
        if((uint64_t)base + tot <= S.cap)
            for(int e = b + PACK_BLOCKS * wave; e < ne; e += PACK_BLOCKS * (PACK_WG / 64)) {
                const uint32_t c = e_cnt[e], src = e_off[e], dst = e_dst[e];
                for(uint32_t i = lane; i < c; i += 64) { S.out[dst + i] = S.site[src + i]; if(S.var) S.vout[dst + i] = S.var[src + i]; }
            }
        base += tot;
        __syncthreads();
    }
    if(b == 0) {
        const int any_bad = __syncthreads_or(bad ? 1 : 0);
        const uint32_t *src = (const uint32_t *)S.d_st; uint32_t *dst = (uint32_t *)S.h_st;
        static_assert(sizeof(SlotStatus) % 4 == 0 && offsetof(SlotStatus, pad) % 4 == 0, "SlotStatus is copied by words");
        for(int i = tid; i < (int)(sizeof(SlotStatus) / 4); i += PACK_WG)
            dst[i] = i == (int)(offsetof(SlotStatus, pad) / 4) ? ((any_bad || base > S.cap) ? PACK_NONE : base) : src[i];
    }
}

// ------------------------------------------------------------------------------------------------
// mbias (MBias.c:57-230): the same admission, segments, context lists and trimming, but no mate-overlap resolution and
// no per-position counters: every call lands in a histogram over (strand, read number, position in read).
// Rows [q][16] with column (strand-1)*4 + (read 2 ? 2 : 0) + (unmethylated ? 1 : 0); the first MB_LQ rows are
// accumulated in LDS per workgroup and flushed once, longer reads go to the global histogram directly.
// ------------------------------------------------------------------------------------------------
#define MB_LQ 512

__device__ __forceinline__ void mbias_seg(const KParams &P, const md_seg &g, int T0, int T1,
                                          const uint16_t *listC, int nC, const uint16_t *listG, int nG, uint32_t *lh) {
    const int send = g.rpos + (int)g.len;
    if(g.rpos >= T1 || send <= T0) return;
    const int strand = g.sf & MDK_SF_STRAND;
    const bool odd = strand & 1;
    const RD o = make_rd(P, g.off4, g.l_qseq, strand, g.sf & MDK_SF_READ2);
    const int lo_off = g.rpos > T0 ? g.rpos - T0 : 0, hi_off = (send < T1 ? send : T1) - T0;
    const int badrs = strand == 0 ? 6 : (odd ? 4 : 2);
    const int col = (strand - 1) * 4 + ((g.sf & MDK_SF_READ2) ? 2 : 0);
    const uint16_t *list = odd ? listC : listG; const int n = odd ? nC : nG;
    int a = 0, b = n;
    while(a < b) { int mid = (a + b) >> 1; if((int)(list[mid] & 0x1fff) < lo_off) a = mid + 1; else b = mid; }
    for(int i = a; i < n; i++) {
        const int e = list[i], l = e & 0x1fff;
        if(l >= hi_off) break;
        if((badrs >> (e >> 13)) & 1) continue;
        if(strand == 0) { atomicExch(P.err, 1); return; }          // updateMetrics aborts on such a read (common.c:122-125)
        const int q = (int)g.q0 + (T0 + l - g.rpos);
        if(q < o.lo || q >= o.hi) continue;                        // trimmed: N with quality 0, below any -p
        const uint32_t sb = o.seq[q >> 1]; const int ql = o.qual[q];
        if(ql < P.minPhred) continue;
        const int bq = (q & 1) ? (sb & 15) : (sb >> 4);
        int un;
        if(odd) { if(bq == 2) un = 0; else if(bq == 8) un = 1; else continue; }
        else { if(bq == 4) un = 0; else if(bq == 1) un = 1; else continue; }
        const int idx = q * 16 + col + un;
        if(q < P.hist_lq) atomicAdd(&lh[idx], 1u); else atomicAdd(&P.hist[idx], 1u);
    }
}

__global__ __launch_bounds__(WG, 8) void k_mbias(const KParams P) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int TILE = P.tile, PER = TILE / WG;
    uint16_t *listC = (uint16_t *)lds, *listG = listC + TILE;
    uint32_t *lh = lds + TILE;
    __shared__ int wsum[WAVES];
    const int b = blockIdx.x;
    const int t = (b & 7) * P.nper + (b >> 3);
    if(t >= P.ntiles) return;
    const int64_t T0 = P.beg + (int64_t)t * TILE;
    const int64_t T1 = (T0 + TILE < P.end) ? T0 + TILE : P.end;
    const int tlen = (int)(T1 - T0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileEnt te = P.tiles[t];
    if(te.last <= te.first) { te.first = 0; te.last = 0; }
    int code[PERMAX];
    load_codes(P, T0, tlen, PER, tid, code);
    const int nh = 16 * P.hist_lq;
    for(int i = tid; i < nh; i += WG) lh[i] = 0;
    int nC, nG;
    build_lists(P, PER, tid, lane, wave, code, listC, listG, wsum, nC, nG);
    for(int r = te.first + tid; r < te.last; r += WG) {
        const md_seg g = P.seg[r];
        mbias_seg(P, g, (int)T0, (int)T1, listC, nC, listG, nG, lh);
    }
    __syncthreads();
    for(int i = tid; i < nh; i += WG) { const uint32_t v = lh[i]; if(v) atomicAdd(&P.hist[i], v); }
}

// ------------------------------------------------------------------------------------------------
// perRead (perRead.c:38-94): one lane per read walks its CIGAR and counts CpG calls of the read itself.  The walk is the
// reference's, including what it does after a base below -p: it steps one base on and evaluates that base without
// looking at its quality or at the CIGAR again, which can run one element past the sequence -- resolved as the BAM
// record layout resolves it (padding nibble of the last sequence byte, or the high nibble of the first quality byte).
// CpG context comes from the resident context codes, clipped to the reference window the command fetches for the chunk
// ([max(beg-2,0), end+10000], perRead.c:176): past its last base nothing is a CpG, and a C in its last base is not one.
// ------------------------------------------------------------------------------------------------
struct PRParams {
    const md_pr_read *read; const uint32_t *cigar; const uint8_t *blob; const uint8_t *ctxcode;
    int64_t reflen, wend; int n, minPhred; md_pr_count *out;
};
__global__ __launch_bounds__(256) void k_perread(const PRParams P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= P.n) return;
    const md_pr_read r = P.read[i];
    const uint8_t *seq = P.blob + 4ull * r.off4, *qual = seq + ((((r.l_qseq + 1) >> 1) + 3) & ~3u);
    const uint32_t *cig = P.cigar + r.cig_off;
    P.out[i] = perread_walk(seq, qual, r.l_qseq, (int)r.n_cigar, r.pos, r.strand & 1, P.ctxcode, P.reflen, P.wend, P.minPhred, [cig](int k) { return cig[k]; });
}

// test hook: effective (post-trim, post-overlap-resolution) base and quality of every base of every segment,
// one lane per base: out[ooff[s] + j] for base j of segment s
__global__ __launch_bounds__(WG) void k_debug_effective(const KParams P, int n_segs, uint8_t *ob, uint8_t *oq, const uint64_t *ooff) {
    int lane = threadIdx.x & 63;
    for(int s0 = blockIdx.x * WAVES + (threadIdx.x >> 6); s0 < n_segs; s0 += gridDim.x * WAVES) {
        const int si = __builtin_amdgcn_readfirstlane(s0);
        const md_seg g = P.seg[si];
        const int strand = g.sf & MDK_SF_STRAND; const bool second = (g.sf & MDK_SF_SECOND) != 0, partner = (g.sf & MDK_SF_PARTNER) != 0;
        const RD o = make_rd(P, g.off4, g.l_qseq, strand, g.sf & MDK_SF_READ2);
        RD m = o; if(partner) m = make_rd(P, g.m_off4, g.m_l_qseq, g.msf & MDK_SF_STRAND, g.msf & MDK_SF_READ2);
        for(int j = lane; j < (int)g.len; j += 64) {
            int q = (int)g.q0 + j, mq = (int)g.m_q0 + j, bq = 15, ql = 0;
            if(q >= o.lo && q < o.hi) { uint32_t sb = o.seq[q >> 1]; bq = (q & 1) ? (sb & 15) : (sb >> 4); ql = o.qual[q]; }
            if(partner) {
                int mb = 15, mqv = 0;
                if(mq >= m.lo && mq < m.hi) { uint32_t sb = m.seq[mq >> 1]; mb = (mq & 1) ? (sb & 15) : (sb >> 4); mqv = m.qual[mq]; }
                ql = resolve_overlap(second, bq, ql, mb, mqv);
            }
            ob[ooff[si] + j] = (uint8_t)bq; oq[ooff[si] + j] = (uint8_t)ql;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the library
// ------------------------------------------------------------------------------------------------
extern "C" const char *md_dev_last_error(void) { return g_err; }

extern "C" int md_dev_count(void) {
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    if(e != hipSuccess) { fail(MDK_ERR_NODEVICE, "hipGetDeviceCount", e); return MDK_ERR_NODEVICE; }
    return n;
}

// the four instantiations of the pileup: with / without the opposite-strand counters, lane or quarter-wavefront per segment
static const void *pileup_fn(bool variant, bool qw) {
    return variant ? (qw ? (const void *)k_pileup<true, true> : (const void *)k_pileup<true, false>) : (qw ? (const void *)k_pileup<false, true> : (const void *)k_pileup<false, false>);
}
static const void *pileup_multi_fn(bool variant, bool qw) {
    return variant ? (qw ? (const void *)k_pileup_multi<true, true> : (const void *)k_pileup_multi<true, false>) : (qw ? (const void *)k_pileup_multi<false, true> : (const void *)k_pileup_multi<false, false>);
}
static void launch_pileup(const md_dev *h, int grid, size_t lds, hipStream_t st, const KParams &P) {
    if(h->variant) { if(h->qw) hipLaunchKernelGGL((k_pileup<true, true>), dim3(grid), dim3(WG), lds, st, P); else hipLaunchKernelGGL((k_pileup<true, false>), dim3(grid), dim3(WG), lds, st, P); }
    else { if(h->qw) hipLaunchKernelGGL((k_pileup<false, true>), dim3(grid), dim3(WG), lds, st, P); else hipLaunchKernelGGL((k_pileup<false, false>), dim3(grid), dim3(WG), lds, st, P); }
}
static void launch_pileup_multi(const md_dev *h, int grid, size_t lds, hipStream_t st, const KMulti &M) {
    if(h->variant) { if(h->qw) hipLaunchKernelGGL((k_pileup_multi<true, true>), dim3(grid), dim3(WG), lds, st, M); else hipLaunchKernelGGL((k_pileup_multi<true, false>), dim3(grid), dim3(WG), lds, st, M); }
    else { if(h->qw) hipLaunchKernelGGL((k_pileup_multi<false, true>), dim3(grid), dim3(WG), lds, st, M); else hipLaunchKernelGGL((k_pileup_multi<false, false>), dim3(grid), dim3(WG), lds, st, M); }
}

// Bring the runtime all the way up for a device -- context, code object -- without needing a configuration yet, so that a
// caller can overlap it with its own start-up; md_dev_open afterwards finds it done.
// streams made ahead of md_dev_open by md_dev_warm (creating one costs the runtime ~5 ms, and needs nothing the options decide)
static std::mutex g_stash_mu; static std::vector<hipStream_t> g_stash; static int g_stash_dev = -1;
static std::atomic<bool> g_warm_reg_stop{false};       // the device is open (any command): whoever uploads from a block registers it from here on
// (Round 4 tried stream priorities -- the consumer's streams high, the device inflate's low, so that kernels of microseconds would not queue
// behind thousands of members: 512 Mb 1.04 -> 0.99 s and 128 Mb 0.376 -> 0.339 s inside WITHOUT them, gpurun_out r04n; all streams are alike.)
#define WARM_STREAMS 4
static std::condition_variable g_stash_cv; static int g_warm_state = 0;       // 1: md_dev_warm is on its way to making the streams, 2: it has
static hipStream_t stream_take(int device) {
    {   // (a caller that overtakes the warm-up -- md_dev_open on its own thread -- waits for the streams being made rather than making its own
        // next to them: stream creation is serial inside the runtime, 5-9 ms each)
        std::unique_lock<std::mutex> lk(g_stash_mu);
        while(g_stash.empty() && g_warm_state == 1) g_stash_cv.wait(lk);
        if(g_stash_dev == device && !g_stash.empty()) { hipStream_t s = g_stash.back(); g_stash.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}
hipStream_t mdk_stream_take(int device) { return stream_take(device); }
// The device inflate's streams have LOW priority, everybody else's the default.  Not for the order of dispatch in the first place: the runtime keeps a pool
// of hardware queues PER PRIORITY, so the pieces' streams no longer share hardware queues with the streams the groups of chunks are launched on.
// With one pool (4 queues for 4 + 3 + 1 streams) a group's five small kernels sat in a queue behind a piece's copy, k_inflate, k_crc32 and walks:
// 15 ms from the launch call to the results for 0.5-2 ms of kernels, three groups in flight -> a group every 5 ms whatever else got faster
// (profiles/r06pb_*: MDK_HOST_PROFILE's group lines, the kernel trace: no group kernel runs while pieces are queued).  Round 4's priority experiment
// (high for the consumer, low for the inflate) predates the group launches and the pieces' shared streams.
// OFF again since the pieces have their lanes (mdk_inflate.hip: every k_inflate on one stream of its own, which shares its hardware queue with copies only): with
// the lanes in, low priority costs the 512 Mb run 4 % (1.044 -> 1.005 s by interleaved runs, profiles/r06_e2e_ab.txt r06pr).  MDK_PIECE_PRIO=1 turns it on.
static std::atomic<int> g_warm_streams_made{0};      // of the WARM_STREAMS the handle's own work runs on
static std::vector<hipStream_t> g_stash_piece;      // (g_stash_mu) made by md_dev_warm's side thread
static bool piece_prio_wanted() { static const bool on = getenv("MDK_PIECE_PRIO") && atoi(getenv("MDK_PIECE_PRIO")) != 0; return on; }
static hipStream_t piece_stream_new() {
    hipStream_t s = nullptr;
    if(piece_prio_wanted()) {
        int least = 0, greatest = 0;
        if(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest && hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) == hipSuccess) return s;
        (void)hipGetLastError(); s = nullptr;
    }
    if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}
hipStream_t mdk_piece_stream_take(int device) {
    { std::lock_guard<std::mutex> lk(g_stash_mu); if(g_stash_dev == device && !g_stash_piece.empty()) { hipStream_t s = g_stash_piece.back(); g_stash_piece.pop_back(); return s; } }
    return piece_stream_new();
}
struct WarmScope { WarmScope() { std::lock_guard<std::mutex> lk(g_stash_mu); g_warm_state = 1; } ~WarmScope() { { std::lock_guard<std::mutex> lk(g_stash_mu); g_warm_state = 2; } g_stash_cv.notify_all(); } };
// The warm-up's side threads.  They are JOINABLE and md_dev_quiesce joins them: before a handle is closed, before the command leaves with
// _exit (mdk_extract.c leave_fast) -- no thread of this library is inside the runtime when the process goes (round 4's threads were detached,
// and a short command could reach _exit while they were still in hipMalloc / hipHostRegister).  None of them touches a kernel symbol: the
// device library's one code object is loaded by md_dev_warm's own thread, once, before anything is launched from it.
static std::mutex g_side_mu; static std::vector<std::thread> &g_side = *new std::vector<std::thread>();      // (never destroyed: a joinable std::thread must not meet its destructor at exit)
static std::atomic<bool> g_side_quit{false}, g_side_forever{false};
// forever: the process is leaving (the command's _exit, atexit) -- nothing is started again.  Otherwise (a handle is being closed in a process that lives
// on: Python, a test suite, bench.py opening handles in turn) the threads are stopped and joined, and the next md_dev_warm / md_dev_open starts its own.
static void side_join(bool forever) {
    std::vector<std::thread> mine;
    { std::lock_guard<std::mutex> lk(g_side_mu); g_side_quit.store(true); g_warm_reg_stop.store(true); mine.swap(g_side); }
    for(std::thread &t : mine) if(t.joinable()) t.join();
    if(!forever) { std::lock_guard<std::mutex> lk(g_side_mu); if(!g_side_forever.load()) g_side_quit.store(false); }
    else g_side_forever.store(true);
}
extern "C" void md_dev_quiesce(void) { side_join(true); }
template <typename F> static void side_start(F &&f) {
    static std::once_flag hook;
    std::call_once(hook, [] { (void)atexit(md_dev_quiesce); });     // a library caller that never closes a handle: joined before the runtime's own exit handlers run
    std::lock_guard<std::mutex> lk(g_side_mu);
    if(g_side_quit.load()) return;                       // the process is on its way out (or a handle has been closed): nothing new is started
    g_side.emplace_back(std::forward<F>(f));
}
static std::once_flag g_code_once; static int g_code_rc = 0;
// the one code object of the library (-fgpu-rdc), loaded by ONE thread; every launch path of the library comes through md_dev_open, which calls this
static int code_object_load() {
    std::call_once(g_code_once, [] {
        hipFuncAttributes fa;
        if(hipFuncGetAttributes(&fa, pileup_fn(false, false)) != hipSuccess || hipFuncGetAttributes(&fa, (const void *)k_classify) != hipSuccess) { g_code_rc = 1; (void)hipGetLastError(); return; }
        if(prep_kernels_init()) g_code_rc = 1;          // (same code object: the attribute of the scan kernel, nothing is loaded again)
        inflate_kernels_warm();
    });
    return g_code_rc;
}
extern "C" int md_dev_warm(int device) {
    WarmScope warm_scope;
    if(!g_side_forever.load()) g_warm_reg_stop.store(false);       // (a warm-up after a handle was closed: its side threads run again)
    const double t0 = mdk_now();
    int n = md_dev_count();
    const double t1 = mdk_now();
    if(n <= 0 || device < 0 || device >= n) return MDK_ERR_NODEVICE;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(nullptr));
    const double t2 = mdk_now();
    { std::lock_guard<std::mutex> lk(g_stash_mu); if(g_stash_dev != device) { g_stash.clear(); g_stash_piece.clear(); g_stash_dev = device; } }
    // What the first chunk and the first piece would otherwise pay for on the pipeline's critical path (gpurun_out r04p, 3 ms time series: the first
    // upload took 75 ms and the first group 60 ms, a later group 8): the copy engines' queues in both directions (made at the first copy), carved
    // device blocks, the pieces' streams.  On threads of their own; whoever needs one of them first waits inside the runtime for that one.
    const bool side_streams = !getenv("MDK_NO_WARM_SIDE") && !g_side_quit.load();
    g_warm_streams_made.store(0);
    if(side_streams) {
        // ... and the staging blocks the host's inflate fills while the runtime comes up are made known to it from the moment it IS up, not from the
        // moment the device handle is open 60-90 ms later (the first chunk's upload waited 50-85 ms for the registration of ~30 blocks, r04q).
        // Until the device is open (md_dev_open stops it), not longer.
        if(!getenv("MDK_NO_PIN") && !getenv("MDK_NO_PREREG"))
            side_start([device]() {
                if(hipSetDevice(device) != hipSuccess) return;
                const double t0 = mdk_now();
                while(!g_warm_reg_stop.load() && !g_side_quit.load() && mdk_now() - t0 < 1.0) { if(md_host_register_all(nullptr, 1) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
            });
        side_start([device]() {          // the device inflate's streams (mdk_inflate.hip piece_stream_of takes them from the stash)
            if(hipSetDevice(device) != hipSuccess) return;
            // the handle's own streams first (the groups', the contigs'): they are wanted the moment the code object is in, and making them next to its load
            // instead of after it brings the device's first use 30 ms forward; then the pieces'
            for(int i = 0; i < WARM_STREAMS && !g_side_quit.load(); i++) {
                hipStream_t s = nullptr;
                if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
                { std::lock_guard<std::mutex> lk(g_stash_mu); if(g_stash_dev == device) g_stash.push_back(s); else { (void)hipStreamDestroy(s); break; } }
                g_warm_streams_made.fetch_add(1); g_stash_cv.notify_all();
            }
            g_warm_streams_made.store(WARM_STREAMS); g_stash_cv.notify_all();
            for(int i = 0; i < 6 && !g_side_quit.load(); i++) { hipStream_t s = piece_stream_new(); if(!s) return; std::lock_guard<std::mutex> lk(g_stash_mu); if(g_stash_dev == device) g_stash_piece.push_back(s); else { (void)hipStreamDestroy(s); return; } }
        });
        side_start([device]() {
            if(hipSetDevice(device) != hipSuccess) return;
            hipStream_t s = stream_take(device); if(!s) return;
            void *hp = harena_take(1u << 20), *dp = arena_take(1u << 20);
            if(hp && dp) { memset(hp, 0, 1u << 20); (void)hipMemcpyAsync(dp, hp, 1u << 20, hipMemcpyHostToDevice, s); (void)hipMemcpyAsync(hp, dp, 1u << 20, hipMemcpyDeviceToHost, s); }
            (void)hipStreamSynchronize(s);
            if(hp) harena_give(hp); if(dp) arena_give(dp);
            if(!g_side_quit.load()) { arena_reserve(reserve_blocks_wanted()); g_reserve_done.store(1); }
            if(!g_side_quit.load() && !getenv("MDK_NO_HARENA_RESERVE")) harena_reserve(4);            // (the slots' pinned result and table buffers: 2-4 MB each, two dozen slots)
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(g_stash_mu); g_stash.push_back(s);
        });
    }
    if(code_object_load()) return fail(MDK_ERR_HIP, "loading the device library's code object", hipGetLastError());
    const double t3 = mdk_now();
    if(side_streams) { std::unique_lock<std::mutex> lk(g_stash_mu); g_stash_cv.wait_for(lk, std::chrono::milliseconds(500), [] { return g_warm_streams_made.load() >= WARM_STREAMS; }); }      // (made by the helper thread meanwhile)
    else for(int i = 0; i < WARM_STREAMS; i++) {
        hipStream_t s = nullptr;
        if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        { std::lock_guard<std::mutex> lk(g_stash_mu);
          if(g_stash_dev != device) { g_stash.clear(); g_stash_dev = device; }
          g_stash.push_back(s); }
        g_stash_cv.notify_all();
    }
    const double t4 = mdk_now();
    arena_prime(device);
    if(mdk_prof_on()) fprintf(stderr, "[mdk hip] warm-up: runtime init + device count %.3fs, context (hipSetDevice + hipFree(0)) %.3fs, the library's code object %.3fs, %d streams %.3fs, first blocks of carved device / pinned memory %.3fs\n", t1 - t0, t2 - t1, t3 - t2, WARM_STREAMS, t4 - t3, mdk_now() - t4);
    return 0;
}

static int fixed_lds(int tile, bool variant) { return tile * ((variant ? 16 : 8) + 4); }

extern "C" int md_dev_open(int device, const md_dev_cfg *cfg, md_dev **out) {
    if(!cfg || !out) return fail(MDK_ERR_ARG, "md_dev_open", hipSuccess);
    *out = nullptr;
    const double t_open0 = mdk_now();
    // the boost identity the kernels rely on, checked against the reference's C expression
    for(int q = 0; q < 256; q++) {
        uint8_t v = (uint8_t)q; v = (uint8_t)(int)(v + 0.2 * v);
        if(v != (uint8_t)(((q * 6) / 5) & 255)) { snprintf(g_err, sizeof(g_err), "boost identity fails at q=%d", q); return MDK_ERR_ARG; }
    }
    int n = md_dev_count();
    if(n <= 0) { if(n == 0) snprintf(g_err, sizeof(g_err), "no HIP device visible"); return MDK_ERR_NODEVICE; }
    if(device < 0 || device >= n) return fail(MDK_ERR_ARG, "md_dev_open: device index", hipSuccess);
    HIPCHK(hipSetDevice(device));
    g_warm_reg_stop.store(true);                       // from here on, whoever uploads from a staging block registers it (every command, not only extract)
    if(code_object_load()) return fail(MDK_ERR_HIP, "loading the device library's code object", hipGetLastError());
    md_dev *h = new md_dev();
    h->device = device; h->cfg = *cfg;
    // default tile: 4 positions per thread (2048) for CpG-only runs, 3 (1536) when CHG/CHH are counted: with ~20 sites per
    // segment the quarter-wavefront kernel wants a tile's segments in one round of the workgroup (8-chunk launches: 33.8 us
    // per 1 Mb chunk at 1536, 36.0 at 2048, 40.6 at 1024; one chunk per launch: 57.7 / 66.0 / 48.3; profiles/r02_kbench_experiments.txt)
    h->tile = cfg->tile > 0 ? cfg->tile : ((cfg->keepCHG || cfg->keepCHH) ? 1536 : DEFAULT_TILE);
    h->tile = (h->tile + WG - 1) / WG * WG;
    if(h->tile > MAX_TILE) h->tile = MAX_TILE;
    h->n_slots = cfg->n_slots > 0 ? cfg->n_slots : 2;
    h->variant = cfg->minOppositeDepth > 0;
    h->qw = (cfg->keepCHG || cfg->keepCHH) && !getenv("MDK_NO_QW");       // dense contexts: a quarter of a wavefront per segment (MDK_NO_QW: the lane-per-segment kernel, for comparison)
    while(fixed_lds(h->tile, h->variant) > LDS_LIMIT - 1024 && h->tile > WG) h->tile -= WG;       // stay inside 160 KiB of LDS
    if(fixed_lds(h->tile, h->variant) > 65536) {    // more than the default dynamic-LDS window: opt in
        HIPCHK(hipFuncSetAttribute(pileup_fn(h->variant, h->qw), hipFuncAttributeMaxDynamicSharedMemorySize, fixed_lds(h->tile, h->variant)));
        HIPCHK(hipFuncSetAttribute(pileup_multi_fn(h->variant, h->qw), hipFuncAttributeMaxDynamicSharedMemorySize, fixed_lds(h->tile, h->variant)));
    }
    h->slots.resize(h->n_slots);
    if(h->d_status.need((size_t)h->n_slots) || h->h_status.need((size_t)h->n_slots)) return MDK_ERR_NOMEM;
    HIPCHK(hipMemset(h->d_status.p, 0, sizeof(SlotStatus) * (size_t)h->n_slots));
    HIPCHK(hipDeviceSynchronize());                    // (the memset has run before any of the slots' non-blocking streams is used)
    memset(h->h_status.p, 0, sizeof(SlotStatus) * (size_t)h->n_slots);
    for(int i = 0; i < h->n_slots; i++) { Slot &s = h->slots[i]; s.index = i; s.d_total.p = h->d_status.p[i].total; s.d_err.p = &h->d_status.p[i].err; s.d_pcnt.p = &h->d_status.p[i].pc; s.h_st.p = &h->h_status.p[i]; }
    const double t_open1 = mdk_now();
    const bool shared = cfg->n_streams > 0;
    const int n_streams = shared ? std::min(cfg->n_streams, (h->n_slots + MAXM - 1) / MAXM) : h->n_slots;
    for(int i = 0; i < n_streams; i++) { hipStream_t st = stream_take(device); if(!st) return fail(MDK_ERR_HIP, "hipStreamCreateWithFlags", hipGetLastError()); h->streams.push_back(st); }
    for(auto &s : h->slots) {
        s.stream = h->streams[(size_t)(shared ? s.index / MAXM : s.index) % (size_t)n_streams];
        HIPCHK(hipEventCreate(&s.e0)); HIPCHK(hipEventCreate(&s.e1)); HIPCHK(hipEventCreate(&s.k0)); HIPCHK(hipEventCreate(&s.k1));
        s.run = s.stream;
    }
    if(mdk_prof_on()) fprintf(stderr, "[mdk hip] md_dev_open: %.3fs (of which streams and events of %d slots %.3fs)\n", mdk_now() - t_open0, h->n_slots, mdk_now() - t_open1);
    g_open_handles.fetch_add(1);
    *out = h;
    return 0;
}

static void ref_release(md_dev *h, size_t tid);
extern "C" void md_dev_close(md_dev *h) {
    if(!h) return;
    side_join(false);
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for(auto &s : h->slots) {
        s.d_seg_in.release(); s.d_blob.release(); s.d_tiles.release(); s.h_tiles.release();
        s.d_raw.release(); s.d_recoff.release(); s.d_prd.release(); s.d_zero.release();
        s.d_aidx.release(); s.h_aidx.release(); s.d_hnext.release(); s.h_rectab.release();
        s.d_pr.release(); s.d_cig.release(); s.d_prc.release(); s.h_prc.release();
        s.d_site.release(); s.d_var.release(); s.d_seg.release();
        s.h_site.release(); s.h_sorted.release(); s.h_var.release(); s.h_vsorted.release(); s.h_seg.release();
        if(s.e0) (void)hipEventDestroy(s.e0); if(s.e1) (void)hipEventDestroy(s.e1); if(s.k0) (void)hipEventDestroy(s.k0); if(s.k1) (void)hipEventDestroy(s.k1);
    }
    for(hipStream_t st : h->streams) if(st) (void)hipStreamDestroy(st);
    for(hipStream_t st : h->piece_streams) if(st) (void)hipStreamDestroy(st);
    if(h->piece_in) (void)hipStreamDestroy(h->piece_in);
    if(h->piece_inf) (void)hipStreamDestroy(h->piece_inf);
    if(h->ref_stream) (void)hipStreamDestroy(h->ref_stream);
    g_open_handles.fetch_sub(1);                       // (before the last carved buffers go: the give that brings the count to zero may start the blocks over)
    h->d_status.release(); h->h_status.release();
    if(h->d_crc) (void)hipFree(h->d_crc);
    if(h->d_hist) (void)hipFree(h->d_hist);
    for(uint32_t *p : h->mapbits) if(p) (void)hipFree(p);
    for(md_region *p : h->d_runs) if(p) (void)hipFree(p);
    for(size_t t = 0; t < h->ref.size(); t++) ref_release(h, t);
    delete h;
}

extern "C" int md_dev_tile(const md_dev *h) { return h ? h->tile : MDK_ERR_ARG; }

// the per-contig tables at their final size: md_dev_set_reference / _regions / _mappability of one contig then never move the entries
// of another, which a thread working on a slot of that contig may be reading
extern "C" int md_dev_reserve_contigs(md_dev *h, int32_t n) {
    if(!h || n < 0) return fail(MDK_ERR_ARG, "md_dev_reserve_contigs", hipSuccess);
    const size_t k = (size_t)n;
    if(h->ref.size() < k) { h->ref.resize(k, nullptr); h->refcode.resize(k, nullptr); h->reflen.resize(k, 0); }
    if(h->ref_carved.size() < k) h->ref_carved.resize(k, 0);
    if(h->d_runs.size() < k) { h->d_runs.resize(k, nullptr); h->n_runs.resize(k, 0); h->has_runs.resize(k, 0); }
    if(h->mapbits.size() < k) { h->mapbits.resize(k, nullptr); h->maplen.resize(k, 0); }
    return 0;
}

static void ref_release(md_dev *h, size_t tid) {
    const bool carved = tid < h->ref_carved.size() && h->ref_carved[tid];
    if(h->ref[tid]) { if(carved) arena_give(h->ref[tid]); else (void)hipFree(h->ref[tid]); h->ref[tid] = nullptr; }
    if(h->refcode[tid]) { if(carved) arena_give(h->refcode[tid]); else (void)hipFree(h->refcode[tid]); h->refcode[tid] = nullptr; }
    if(tid < h->ref_carved.size()) h->ref_carved[tid] = 0;
}
extern "C" int md_dev_set_reference(md_dev *h, int32_t tid, const char *seq, int64_t len) {
    if(!h || tid < 0 || !seq || len < 0) return fail(MDK_ERR_ARG, "md_dev_set_reference", hipSuccess);
    ProfScope pf(PF_SETREF); MarkScope mk_ref("md_dev_set_reference");
    HIPCHK(hipSetDevice(h->device));
    if((size_t)tid >= h->ref.size()) { h->ref.resize(tid + 1, nullptr); h->refcode.resize(tid + 1, nullptr); h->reflen.resize(tid + 1, 0); }
    if(h->ref_carved.size() < h->ref.size()) h->ref_carved.resize(h->ref.size(), 0);
    ref_release(h, tid);
    char *d = nullptr; uint8_t *c = nullptr; bool carved = false;
    if((size_t)len + 16 < ARENA_MAX) { d = (char *)arena_take((size_t)len + 16); c = d ? (uint8_t *)arena_take((size_t)len + 16) : nullptr; if(d && !c) { arena_give(d); d = nullptr; } carved = d != nullptr; }      // (made ahead by the warm-up: no allocation next to the pieces' copies)
    if(!carved) {
        hipError_t e = hipMalloc((void **)&d, (size_t)len + 16);
        if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc(reference)", e);
        e = hipMalloc((void **)&c, (size_t)len + 16);
        if(e != hipSuccess) { (void)hipFree(d); return fail(MDK_ERR_NOMEM, "hipMalloc(reference codes)", e); }
    }
    h->ref_carved[tid] = carved ? 1 : 0;
    // on a stream of its own: a contig can be uploaded (by another thread) while chunks of the contigs before it are worked on, and neither waits for the other
    if(!h->ref_stream && !(h->ref_stream = stream_take(h->device))) return fail(MDK_ERR_HIP, "hipStreamCreateWithFlags", hipGetLastError());
    HIPCHK(hipMemcpyAsync(d, seq, (size_t)len, hipMemcpyHostToDevice, h->ref_stream));
    if(len > 0) {
        int64_t blocks = (len + WG - 1) / WG; if(blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_classify, dim3((unsigned)blocks), dim3(WG), 0, h->ref_stream, d, c, len);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(h->ref_stream));
    h->ref[tid] = d; h->refcode[tid] = c; h->reflen[tid] = len;
    return 0;
}

extern "C" int md_dev_set_regions(md_dev *h, int32_t tid, const md_region *runs, int64_t n) {
    if(!h || tid < 0 || n < 0 || (n && !runs)) return fail(MDK_ERR_ARG, "md_dev_set_regions", hipSuccess);
    if((size_t)tid >= h->ref.size() || !h->refcode[tid]) { snprintf(g_err, sizeof(g_err), "reference for tid %d not uploaded", tid); return MDK_ERR_NOREF; }
    for(int64_t i = 0; i < n; i++)
        if(runs[i].start < 0 || runs[i].end <= runs[i].start || (i && runs[i].start < runs[i - 1].end) || runs[i].strand < 0 || runs[i].strand > 2)
            return fail(MDK_ERR_ARG, "md_dev_set_regions: runs must be sorted, disjoint, non-empty, strand in 0..2", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    const int64_t len = h->reflen[tid];
    md_region *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, sizeof(md_region) * (size_t)(n ? n : 1));
    if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc(regions)", e);
    if(n) { e = hipMemcpy(d, runs, sizeof(md_region) * (size_t)n, hipMemcpyHostToDevice); if(e != hipSuccess) { (void)hipFree(d); return fail(MDK_ERR_HIP, "hipMemcpy(regions)", e); } }
    int64_t blocks = (len + WG - 1) / WG; if(blocks > 65536) blocks = 65536;
    if(!h->ref_stream && !(h->ref_stream = stream_take(h->device))) return fail(MDK_ERR_HIP, "hipStreamCreateWithFlags", hipGetLastError());
    if(len > 0) hipLaunchKernelGGL(k_mask_regions, dim3((unsigned)blocks), dim3(WG), 0, h->ref_stream, h->refcode[tid], len, d, n);
    e = hipGetLastError(); if(e == hipSuccess) e = hipStreamSynchronize(h->ref_stream);
    if(e != hipSuccess) { (void)hipFree(d); return fail(MDK_ERR_HIP, "k_mask_regions", e); }
    // the runs stay resident: the device chunk preparation tests every read's span against them (common.c:432-439)
    if((size_t)tid >= h->d_runs.size()) { h->d_runs.resize(tid + 1, nullptr); h->n_runs.resize(tid + 1, 0); h->has_runs.resize(tid + 1, 0); }
    if(h->d_runs[tid]) (void)hipFree(h->d_runs[tid]);
    h->d_runs[tid] = d; h->n_runs[tid] = n; h->has_runs[tid] = 1;
    return 0;
}

Slot *get_slot(md_dev *h, int slot) { if(!h || slot < 0 || slot >= h->n_slots) { fail(MDK_ERR_ARG, "bad slot", hipSuccess); return nullptr; } return &h->slots[slot]; }

// segment run of every tile
static void build_tiles(const md_read_batch *b, int TILE, TileEnt *te, int ntiles) {
    for(int t = 0; t < ntiles; t++) { te[t].first = 0x7fffffff; te[t].last = 0; }
    for(int i = 0; i < b->n_segs; i++) {
        int64_t lo = b->seg[i].rpos, hi = lo + b->seg[i].len;
        if(hi <= lo || hi <= b->beg || lo >= b->end) continue;
        if(lo < b->beg) lo = b->beg; if(hi > b->end) hi = b->end;
        int t0 = (int)((lo - b->beg) / TILE), t1 = (int)((hi - 1 - b->beg) / TILE);
        for(int t = t0; t <= t1; t++) { if(te[t].first > i) te[t].first = i; te[t].last = i + 1; }
    }
    for(int t = 0; t < ntiles; t++) if(te[t].last == 0) te[t].first = 0;
}

extern "C" int md_dev_upload(md_dev *h, int slot, const md_read_batch *b) {
    Slot *s = get_slot(h, slot);
    if(!s || !b || b->n_segs < 0 || b->end < b->beg) return fail(MDK_ERR_ARG, "md_dev_upload", hipSuccess);
    if(b->n_segs && (!b->seg || !b->blob)) return fail(MDK_ERR_ARG, "md_dev_upload: null array", hipSuccess);
    if(b->tid < 0 || (size_t)b->tid >= h->ref.size() || !h->ref[b->tid]) { snprintf(g_err, sizeof(g_err), "reference for tid %d not uploaded", b->tid); return MDK_ERR_NOREF; }
    HIPCHK(hipSetDevice(h->device));
    if(s->busy) {                                     // the slot's previous contents are being replaced
        HIPCHK(hipStreamSynchronize(s->stream));
        if(s->run && s->run != s->stream) HIPCHK(hipStreamSynchronize(s->run));
    }
    s->busy = true;
    s->fresh = true;
    if((uint64_t)b->blob_bytes >= (1ull << 32) - 64) return fail(MDK_ERR_ARG, "md_dev_upload: more than 4 GiB of read payload in one chunk", hipSuccess);   // the dense-context kernel addresses payload bytes with 32 bits
    const int64_t span = b->end - b->beg;
    s->n_segs = b->n_segs; s->n_reads = b->n_reads; s->tid = b->tid; s->beg = b->beg; s->end = b->end; s->uploaded = false; s->launched = false; s->raw_layout = false;
    const int TILE = h->tile;
    const int ntiles = (int)((span + TILE - 1) / TILE);
    if(s->h_tiles.need((size_t)(ntiles > 0 ? ntiles : 1))) return MDK_ERR_NOMEM;
    build_tiles(b, TILE, s->h_tiles.p, ntiles);
    s->tile = TILE; s->ntiles = ntiles; s->lds_bytes = fixed_lds(TILE, h->variant);
    s->read_bytes = b->algo_bytes;
    size_t ns = (size_t)b->n_segs, nt = (size_t)(ntiles > 0 ? ntiles : 1);
    if(s->d_seg_in.need(ns + 1) || s->d_blob.need((size_t)b->blob_bytes + 64)) return MDK_ERR_NOMEM;
    if(s->d_tiles.need(nt) || s->d_seg.need(nt)) return MDK_ERR_NOMEM;
    if(!s->b_site) {
        if(s->d_site.need((size_t)span + 16)) return MDK_ERR_NOMEM;
        if(h->variant && s->d_var.need((size_t)span + 16)) return MDK_ERR_NOMEM;
    }
    if(ns) {
        host_block_ensure_registered(b->seg); host_block_ensure_registered(b->blob);
        HIPCHK(hipMemcpyAsync(s->d_seg_in.p, b->seg, ns * sizeof(md_seg), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_blob.p, b->blob, (size_t)b->blob_bytes, hipMemcpyHostToDevice, s->stream));
    }
    if(ntiles) HIPCHK(hipMemcpyAsync(s->d_tiles.p, s->h_tiles.p, nt * sizeof(TileEnt), hipMemcpyHostToDevice, s->stream));
    s->uploaded = true;
    return 0;
}

extern "C" int md_dev_bind_output(md_dev *h, int slot, void *d_site, void *d_var, void *d_seg, int64_t cap_sites, int64_t cap_tiles) {
    Slot *s = get_slot(h, slot);
    if(!s) return MDK_ERR_ARG;
    if(!d_site && !d_seg) { s->b_site = nullptr; s->b_var = nullptr; s->b_seg = nullptr; s->b_cap_sites = s->b_cap_tiles = 0; return 0; }
    if(!d_site || !d_seg || cap_sites < 0 || cap_tiles < 0 || (h->variant && !d_var)) return fail(MDK_ERR_ARG, "md_dev_bind_output", hipSuccess);
    s->b_site = (md_site *)d_site; s->b_var = (md_site_var *)d_var; s->b_seg = (md_tile_seg *)d_seg; s->b_cap_sites = cap_sites; s->b_cap_tiles = cap_tiles;
    return 0;
}

static int fill_kparams(md_dev *h, Slot *s, KParams &P) {
    memset(&P, 0, sizeof(P));
    P.seg = s->d_seg_in.p; P.blob = s->raw_layout ? const_cast<uint8_t *>(s->raw_at) : s->d_blob.p; P.packed = s->raw_layout ? 1 : 0;
    P.ctxcode = h->refcode[s->tid]; P.reflen = h->reflen[s->tid];
    P.beg = s->beg; P.end = s->end; P.tile = s->tile; P.ntiles = s->ntiles; P.nper = (s->ntiles + 7) / 8;
    P.tiles = s->d_tiles.p;
    if(s->b_site) {
        if(s->b_cap_tiles < s->ntiles) return fail(MDK_ERR_ARG, "bound tile-segment buffer too small", hipSuccess);
        P.site = s->b_site; P.var = s->b_var; P.tseg = s->b_seg; P.cap_sites = s->b_cap_sites;
    } else { P.site = s->d_site.p; P.var = s->d_var.p; P.tseg = s->d_seg.p; P.cap_sites = (int64_t)s->d_site.cap; }
    P.total = s->d_total.p + (s->ring % RING); P.total_next = s->d_total.p + ((s->ring + 1) % RING);
    P.keepmask = (h->cfg.keepCpG ? 1 : 0) | (h->cfg.keepCHG ? 2 : 0) | (h->cfg.keepCHH ? 4 : 0);
    P.minPhred = h->cfg.minPhred;
    for(int i = 0; i < 16; i++) { P.bounds[i] = h->cfg.bounds[i]; P.abounds[i] = h->cfg.absoluteBounds[i]; }
    P.err = s->d_err.p;
    return 0;
}

// `on` = stream to launch on (the slot's own unless a benchmark lines several slots up on one stream)
int launch_kernels(md_dev *h, Slot *s, bool time_pileup, hipStream_t on) {
    hipStream_t st = on ? on : s->stream;
    if(s->fresh && st != s->stream) { HIPCHK(hipEventRecord(s->e0, s->stream)); HIPCHK(hipStreamWaitEvent(st, s->e0, 0)); }      // the slot's upload comes first
    if(s->raw_layout && s->prep_pending) { Slot *one[1] = {s}; int rc = enqueue_prep_group(h, one, 1, st); if(rc) return rc; }       // its preparation, then the pileup
    s->fresh = false; s->run = st; s->packed = false;
    s->ring++;                                          // a fresh (already zero) site counter for this launch
    if(s->ntiles > 0) {
        KParams P; int rc = fill_kparams(h, s, P); if(rc) return rc;
        int grid = P.nper * 8;
        if(time_pileup) HIPCHK(hipEventRecord(s->k0, st));
        launch_pileup(h, grid, (size_t)s->lds_bytes, st, P);
        if(time_pileup) HIPCHK(hipEventRecord(s->k1, st));
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemsetAsync(s->d_total.p, 0, sizeof(uint32_t) * RING, st));
    }
    return 0;
}

extern "C" int md_dev_launch(md_dev *h, int slot) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded) return fail(MDK_ERR_ARG, "md_dev_launch: slot not uploaded", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_kernels(h, s, false);
    if(rc) return rc;
    s->launched = true;
    return 0;
}

// One kernel launch over up to MAXM uploaded slots (see k_pileup_multi).  The launch goes to the first slot's stream, ordered
// after whatever the other slots' streams still have queued (their uploads / preparation); every slot's stream then waits
// for it, so download / wait per slot work as after md_dev_launch.
static bool pack_wanted() { static const bool on = !getenv("MDK_NO_PACK"); return on; }
int launch_group_on(md_dev *h, const int *slots, int n, hipStream_t on, bool cross_sync) {
    if(!h || !slots || n < 1 || n > MAXM) return fail(MDK_ERR_ARG, "md_dev_launch_group: 1..8 slots", hipSuccess);
    static_assert(sizeof(KMulti) <= 4096, "kernel arguments are limited to 4 KiB");
    KMulti M; memset(&M, 0, sizeof(M));
    Slot *s0 = get_slot(h, slots[0]); if(!s0) return MDK_ERR_ARG;
    hipStream_t st = on ? on : s0->stream;
    int total = 0;
    for(int i = 0; i < n; i++) {
        Slot *s = get_slot(h, slots[i]);
        if(!s || !s->uploaded) return fail(MDK_ERR_ARG, "md_dev_launch_group: slot not uploaded", hipSuccess);
        for(int k = 0; k < i; k++) if(slots[k] == slots[i]) return fail(MDK_ERR_ARG, "md_dev_launch_group: a slot is listed twice", hipSuccess);
        if(s->tile != s0->tile) return fail(MDK_ERR_ARG, "md_dev_launch_group: slots of different tile size", hipSuccess);
        s->ring++;
        int rc = fill_kparams(h, s, M.P[i]); if(rc) return rc;
        M.tstart[i] = total; total += s->ntiles > 0 ? s->ntiles : 0;
        if(s->ntiles <= 0) HIPCHK(hipMemsetAsync(s->d_total.p, 0, sizeof(uint32_t) * RING, st));
        if(cross_sync && s->fresh && s->stream != st) { HIPCHK(hipEventRecord(s->e0, s->stream)); HIPCHK(hipStreamWaitEvent(st, s->e0, 0)); }      // its upload / preparation comes first
        s->fresh = false; s->run = st;
    }
    M.n = n; M.tstart[n] = total; M.nper = (total + 7) / 8;
    if(mdk_prof_on() && !on) { HIPCHK(hipEventRecord(s0->k0, st)); s0->t_launch = mdk_now(); }      // MDK_HOST_PROFILE: the group's kernels on the device's clock, the launch on the host's
    {   // chunks whose records were uploaded but not prepared yet: their preparation kernels, all chunks per launch
        Slot *pend[MAXM]; int np = 0;
        for(int i = 0; i < n; i++) { Slot *s = get_slot(h, slots[i]); if(s->raw_layout && s->prep_pending) pend[np++] = s; }
        if(np) { int rc = enqueue_prep_group(h, pend, np, st); if(rc) return rc; }
    }
    if(total > 0) {
        launch_pileup_multi(h, M.nper * 8, (size_t)s0->lds_bytes, st, M);
        HIPCHK(hipGetLastError());
    }
    for(int i = 0; i < n; i++) get_slot(h, slots[i])->packed = false;
    if(pack_wanted() && !on) {      // (not in the benchmarks' resident loops: their step is preparation + pileup) the ordered sites and the status blocks go to pinned host memory by a kernel of the same stream (k_sites_pack)
        KPack K; memset(&K, 0, sizeof(K)); K.n = n; bool ok = true;
        for(int i = 0; i < n && ok; i++) {
            Slot *s = get_slot(h, slots[i]);
            if(s->b_site) { ok = false; break; }               // caller-bound device output (the exchange between GPUs reads it there)
            const size_t span = (size_t)(s->end > s->beg ? s->end - s->beg : 0);
            size_t want = std::max<size_t>(span / 8, 65536); want = std::min(want, span + 16);      // (a slot that met more sites than this before has grown its buffer to fit them: md_dev_download_group)
            if(s->h_sorted.need(want) || (h->variant && s->h_vsorted.need(s->h_sorted.cap))) { (void)hipGetLastError(); ok = false; break; }
            KPackSlot &S = K.S[i];
            S.site = s->d_site.p; S.var = h->variant ? s->d_var.p : nullptr; S.tseg = s->d_seg.p; S.out = s->h_sorted.p; S.vout = h->variant ? s->h_vsorted.p : nullptr;
            S.d_st = h->d_status.p + s->index; S.h_st = h->h_status.p + s->index; S.ntiles = s->ntiles > 0 ? s->ntiles : 0;
            S.cap = (uint32_t)std::min<size_t>(h->variant ? std::min(s->h_sorted.cap, s->h_vsorted.cap) : s->h_sorted.cap, 0xfffffff0u); S.site_cap = (uint32_t)std::min<size_t>(s->d_site.cap, 0xfffffff0u);
        }
        if(ok) {
            hipLaunchKernelGGL(k_sites_pack, dim3(n * PACK_BLOCKS), dim3(PACK_WG), 0, st, K);
            HIPCHK(hipGetLastError());
            for(int i = 0; i < n; i++) get_slot(h, slots[i])->packed = true;
        }
    }
    if(mdk_prof_on() && !on) HIPCHK(hipEventRecord(s0->k1, st));
    for(int i = 0; i < n; i++) get_slot(h, slots[i])->launched = true;      // collected through each slot's `run` stream (finish_count)
    return 0;
}
// One kernel launch over up to MAXM uploaded slots (see k_pileup_multi).  The launch goes to the first slot's stream, ordered
// after whatever the other slots' streams still have queued (their uploads / preparation); every slot's stream then waits
// for it, so download / wait per slot work as after md_dev_launch.
extern "C" int md_dev_launch_group(md_dev *h, const int *slots, int n) {
    if(!h) return fail(MDK_ERR_ARG, "md_dev_launch_group", hipSuccess);
    ProfScope pf(PF_LAUNCH);
    HIPCHK(hipSetDevice(h->device));
    const double t0 = mdk_prof_on() ? mdk_now() : 0;
    int rc; { MarkScope mk("md_dev_launch_group"); rc = launch_group_on(h, slots, n, nullptr, true); }
    if(mdk_prof_on() && mdk_now() - t0 > 0.004) { char w[96]; snprintf(w, sizeof(w), "slow md_dev_launch_group: %.1f ms", (mdk_now() - t0) * 1e3); mdk_marks_dump(w); }
    return rc;
}
extern "C" int md_dev_group_max(void) { return MAXM; }

extern "C" int md_dev_submit(md_dev *h, int slot, const md_read_batch *b) {
    int rc = md_dev_upload(h, slot, b);
    if(rc) return rc;
    return md_dev_launch(h, slot);
}

// ------------------------------------------------------------------------------------------------
// mbias entry points
// ------------------------------------------------------------------------------------------------
static int hist_reserve(md_dev *h, int rows) {
    if(rows <= h->hist_cap) return 0;
    int cap = h->hist_cap ? h->hist_cap : 1024;
    while(cap < rows) cap *= 2;
    HIPCHK(hipDeviceSynchronize());                    // launches in flight still add into the old buffer
    uint32_t *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, (size_t)cap * 16 * sizeof(uint32_t));
    if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc(mbias histogram)", e);
    HIPCHK(hipMemset(d, 0, (size_t)cap * 16 * sizeof(uint32_t)));
    if(h->d_hist) { HIPCHK(hipMemcpy(d, h->d_hist, (size_t)h->hist_cap * 16 * sizeof(uint32_t), hipMemcpyDeviceToDevice)); (void)hipFree(h->d_hist); }
    HIPCHK(hipDeviceSynchronize());                    // hipMemset of device memory returns before it has run, and the slots' streams do not wait for the null stream: a k_mbias launched now could add its counts first and have them wiped
    h->d_hist = d; h->hist_cap = cap;
    return 0;
}

extern "C" int md_dev_mbias_submit(md_dev *h, int slot, const md_read_batch *b) {
    int rc = md_dev_upload(h, slot, b);
    if(rc) return rc;
    Slot *s = get_slot(h, slot);
    int maxlq = 0;
    for(int i = 0; i < b->n_segs; i++) if((int)b->seg[i].l_qseq > maxlq) maxlq = (int)b->seg[i].l_qseq;
    if((rc = hist_reserve(h, maxlq > 1 ? maxlq : 1)) != 0) return rc;
    if(maxlq > h->hist_len) h->hist_len = maxlq;
    if(s->ntiles <= 0 || b->n_segs == 0) return 0;
    KParams P; if((rc = fill_kparams(h, s, P)) != 0) return rc;
    P.mbias = 1; P.hist = h->d_hist;
    P.hist_lq = maxlq < MB_LQ ? (maxlq + 7) & ~7 : MB_LQ; if(P.hist_lq > h->hist_cap) P.hist_lq = h->hist_cap;
    const size_t lds = (size_t)s->tile * 4 + (size_t)P.hist_lq * 16 * sizeof(uint32_t);
    hipLaunchKernelGGL(k_mbias, dim3(P.nper * 8), dim3(WG), lds, s->stream, P);
    HIPCHK(hipGetLastError());
    return 0;
}

// The same from the chunk's raw records: the device prepares them (no pairing: md_prep_cfg.no_pairing) and tells how long the longest
// admitted read is, which sizes the histogram rows kept in LDS.  The histogram kernel of a chunk therefore waits for its preparation's
// report -- but the host does not wait for it here: a submit queues the chunk's upload, preparation and report, and then sends the OTHER
// slots' chunks, whose reports have had a whole upload's time to arrive, on to the histogram kernel (round 4 waited for every chunk's
// preparation before the next chunk's records could start crossing the link).
static int mbias_finish(md_dev *h, Slot *s) {
    if(!s->mb_pending) return 0;
    s->mb_pending = false;
    HIPCHK(hipStreamSynchronize(s->stream));
    int rc = prep_outcome(h, s);
    if(rc == MDK_ERR_PREP_REDO) {
        HIPCHK(hipMemcpyAsync(s->h_st.p, h->d_status.p + s->index, sizeof(SlotStatus), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        rc = prep_outcome(h, s);
    }
    if(rc) return rc;
    const int maxlq = (int)s->h_st.p->pc.max_lq;
    if((rc = hist_reserve(h, maxlq > 1 ? maxlq : 1)) != 0) return rc;
    if(maxlq > h->hist_len) h->hist_len = maxlq;
    if(s->ntiles <= 0 || s->n_segs <= 0) return 0;
    KParams P; if((rc = fill_kparams(h, s, P)) != 0) return rc;
    P.mbias = 1; P.hist = h->d_hist;
    P.hist_lq = maxlq < MB_LQ ? (maxlq + 7) & ~7 : MB_LQ; if(P.hist_lq > h->hist_cap) P.hist_lq = h->hist_cap;
    const size_t lds = (size_t)s->tile * 4 + (size_t)P.hist_lq * 16 * sizeof(uint32_t);
    hipLaunchKernelGGL(k_mbias, dim3(P.nper * 8), dim3(WG), lds, s->stream, P);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int md_dev_mbias_submit_raw(md_dev *h, int slot, const md_raw_batch *b) {
    if(!h || !h->prep_set || !h->prep.no_pairing) return fail(MDK_ERR_ARG, "md_dev_mbias_submit_raw: md_dev_set_prep with no_pairing first", hipSuccess);
    Slot *s = get_slot(h, slot);
    if(!s) return MDK_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    int rc = mbias_finish(h, s);                       // (a slot submitted twice in a row: its earlier chunk goes on first)
    if(rc) return rc;
    if((rc = md_dev_upload_raw(h, slot, b)) != 0) return rc;
    { Slot *one[1] = {s}; rc = enqueue_prep_group(h, one, 1, s->stream); if(rc) return rc; }
    HIPCHK(hipMemcpyAsync(s->h_st.p, h->d_status.p + s->index, sizeof(SlotStatus), hipMemcpyDeviceToHost, s->stream));
    s->mb_pending = true;
    for(Slot &o : h->slots) if(&o != s && o.mb_pending && (rc = mbias_finish(h, &o)) != 0) return rc;
    return 0;
}

extern "C" int md_dev_slot_sync(md_dev *h, int slot) {
    Slot *s = get_slot(h, slot);
    if(!s) return MDK_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    { const int rc = mbias_finish(h, s); if(rc) return rc; }
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int md_dev_mbias_read(md_dev *h, md_mbias *out) {
    if(!h || !out) return fail(MDK_ERR_ARG, "md_dev_mbias_read", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    for(Slot &o : h->slots) { const int rc = mbias_finish(h, &o); if(rc) return rc; }
    HIPCHK(hipDeviceSynchronize());
    for(auto &s : h->slots) {
        int err = 0;
        HIPCHK(hipMemcpy(&err, s.d_err.p, sizeof(int), hipMemcpyDeviceToHost));
        if(err) { snprintf(g_err, sizeof(g_err), "Can't determine the strand of a read!"); (void)hipMemset(s.d_err.p, 0, sizeof(int)); return MDK_ERR_STRAND0; }
    }
    h->h_hist.assign((size_t)(h->hist_len > 0 ? h->hist_len : 1) * 16, 0u);
    if(h->hist_len > 0) HIPCHK(hipMemcpy(h->h_hist.data(), h->d_hist, (size_t)h->hist_len * 16 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out->len = h->hist_len; out->count = h->h_hist.data();
    return 0;
}

extern "C" int md_dev_mbias_reset(md_dev *h) {
    if(!h) return MDK_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    for(Slot &o : h->slots) { const int rc = mbias_finish(h, &o); if(rc) return rc; }
    HIPCHK(hipDeviceSynchronize());
    if(h->d_hist) { HIPCHK(hipMemset(h->d_hist, 0, (size_t)h->hist_cap * 16 * sizeof(uint32_t))); HIPCHK(hipDeviceSynchronize()); }
    h->hist_len = 0;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// perRead entry points
// ------------------------------------------------------------------------------------------------
extern "C" int md_dev_perread_submit(md_dev *h, int slot, const md_pr_batch *b) {
    Slot *s = get_slot(h, slot);
    if(!s || !b || b->n_reads < 0 || b->end < b->beg) return fail(MDK_ERR_ARG, "md_dev_perread_submit", hipSuccess);
    if(b->n_reads && (!b->read || !b->blob || (b->n_cigar && !b->cigar))) return fail(MDK_ERR_ARG, "md_dev_perread_submit: null array", hipSuccess);
    if(b->tid < 0 || (size_t)b->tid >= h->ref.size() || !h->ref[b->tid]) { snprintf(g_err, sizeof(g_err), "reference for tid %d not uploaded", b->tid); return MDK_ERR_NOREF; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->pr_n = -1;
    const size_t n = (size_t)b->n_reads;
    if(s->d_pr.need(n + 1) || s->d_cig.need((size_t)b->n_cigar + 1) || s->d_blob.need((size_t)b->blob_bytes + 64) || s->d_prc.need(n + 1) || s->h_prc.need(n + 1)) return MDK_ERR_NOMEM;
    if(n) {
        host_block_ensure_registered(b->blob);
        HIPCHK(hipMemcpyAsync(s->d_pr.p, b->read, n * sizeof(md_pr_read), hipMemcpyHostToDevice, s->stream));
        if(b->n_cigar) HIPCHK(hipMemcpyAsync(s->d_cig.p, b->cigar, (size_t)b->n_cigar * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_blob.p, b->blob, (size_t)b->blob_bytes, hipMemcpyHostToDevice, s->stream));
        PRParams P;
        P.read = s->d_pr.p; P.cigar = s->d_cig.p; P.blob = s->d_blob.p; P.ctxcode = h->refcode[b->tid]; P.reflen = h->reflen[b->tid];
        P.wend = b->end + 10000; if(P.wend > P.reflen - 1) P.wend = P.reflen - 1;
        P.n = b->n_reads; P.minPhred = h->cfg.minPhred; P.out = s->d_prc.p;
        hipLaunchKernelGGL(k_perread, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, P);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(s->h_prc.p, s->d_prc.p, n * sizeof(md_pr_count), hipMemcpyDeviceToHost, s->stream));
    }
    s->pr_n = b->n_reads;
    return 0;
}

extern "C" int md_dev_perread_download(md_dev *h, int slot, const md_pr_count **out, int64_t *n) {
    Slot *s = get_slot(h, slot);
    if(!s || !out || !n || s->pr_n < 0) return fail(MDK_ERR_ARG, "md_dev_perread_download: nothing submitted on this slot", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    *out = s->h_prc.p; *n = s->pr_n;
    return 0;
}

// what a launch left for the host: the status block is already in h_st
int64_t finish_eval(md_dev *h, Slot *s) {
    const SlotStatus &st = *s->h_st.p;
    if(s->raw_layout) {
        int rc = prep_outcome(h, s);
        if(rc == MDK_ERR_PREP_REDO) {            // the segment array was too small: preparation is queued again, the pileup follows it
            rc = launch_kernels(h, s, false); if(rc) return rc;
            return finish_count(h, s);
        }
        if(rc) return rc;
    }
    if(st.err) { snprintf(g_err, sizeof(g_err), "Can't determine the strand of a read!"); (void)hipMemset(s->d_err.p, 0, sizeof(int)); return MDK_ERR_STRAND0; }
    int64_t n = (int64_t)st.total[s->ring % RING];
    int64_t cap = s->b_site ? s->b_cap_sites : (int64_t)s->d_site.cap;
    if(n > cap) { snprintf(g_err, sizeof(g_err), "site buffer too small: %lld sites, capacity %lld", (long long)n, (long long)cap); return MDK_ERR_ARG; }
    return n;
}
// wait for the launch, read the status block (site count, error word, preparation counters) with one copy
int64_t finish_count(md_dev *h, Slot *s) {
    if(!s->launched) { fail(MDK_ERR_ARG, "slot not launched", hipSuccess); return MDK_ERR_ARG; }
    hipStream_t st = s->run ? s->run : s->stream;
    {
        ProfScope pf(PF_FIN_WAIT);
        if(hipMemcpyAsync(s->h_st.p, h->d_status.p + s->index, sizeof(SlotStatus), hipMemcpyDeviceToHost, st) != hipSuccess) return fail(MDK_ERR_HIP, "D2H status", hipGetLastError());
        hipError_t e = hipStreamSynchronize(st);
        if(e != hipSuccess) return fail(MDK_ERR_HIP, "hipStreamSynchronize", e);
    }
    return finish_eval(h, s);
}
// the same for the slots of one group launch: one copy covering all of them
int finish_group(md_dev *h, const int *slots, int n, int64_t *counts) {
    int lo = 0x7fffffff, hi = -1; hipStream_t st = nullptr;
    for(int i = 0; i < n; i++) {
        Slot *s = get_slot(h, slots[i]); if(!s || !s->launched) return fail(MDK_ERR_ARG, "slot not launched", hipSuccess);
        if(i == 0) st = s->run; else if(s->run != st) st = nullptr;
        lo = std::min(lo, s->index); hi = std::max(hi, s->index);
    }
    if(!st) { for(int i = 0; i < n; i++) { int64_t c = finish_count(h, get_slot(h, slots[i])); if(c < 0) return (int)c; counts[i] = c; } return 0; }
    if(hipMemcpyAsync(h->h_status.p + lo, h->d_status.p + lo, sizeof(SlotStatus) * (size_t)(hi - lo + 1), hipMemcpyDeviceToHost, st) != hipSuccess) return fail(MDK_ERR_HIP, "D2H status", hipGetLastError());
    hipError_t e = hipStreamSynchronize(st);
    if(e != hipSuccess) return fail(MDK_ERR_HIP, "hipStreamSynchronize", e);
    for(int i = 0; i < n; i++) { int64_t c = finish_eval(h, get_slot(h, slots[i])); if(c < 0) return (int)c; counts[i] = c; }
    return 0;
}

extern "C" int md_dev_wait(md_dev *h, int slot, md_sites_dev *out) {
    Slot *s = get_slot(h, slot);
    if(!s || !out) return fail(MDK_ERR_ARG, "md_dev_wait", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    memset(out, 0, sizeof(*out));
    int64_t n = finish_count(h, s);
    if(n < 0) return (int)n;
    out->n_slots = n; out->n_tiles = s->ntiles;
    out->d_site = s->b_site ? s->b_site : s->d_site.p; out->d_var = h->variant ? (s->b_site ? s->b_var : s->d_var.p) : nullptr;
    out->d_seg = s->b_site ? s->b_seg : s->d_seg.p;
    return 0;
}

extern "C" int64_t md_sites_order(const md_site *site, const md_site_var *var, const md_tile_seg *seg, int32_t n_tiles, int64_t n_slots, md_site *out_site, md_site_var *out_var) {
    if(n_slots < 0 || n_tiles < 0 || (n_slots && (!site || !seg || !out_site))) return MDK_ERR_ARG;
    int64_t o = 0;
    for(int t = 0; t < n_tiles; t++) {
        uint32_t c = seg[t].cnt;
        if(!c) continue;
        if((int64_t)seg[t].off + c > n_slots || o + c > n_slots) return MDK_ERR_ARG;
        memcpy(out_site + o, site + seg[t].off, (size_t)c * sizeof(md_site));
        if(var && out_var) memcpy(out_var + o, var + seg[t].off, (size_t)c * sizeof(md_site_var));
        o += c;
    }
    return o;
}

extern "C" int md_dev_download(md_dev *h, int slot, md_sites *out) {
    Slot *s = get_slot(h, slot);
    if(!s || !out) return fail(MDK_ERR_ARG, "md_dev_download", hipSuccess);
    memset(out, 0, sizeof(*out));
    md_sites_dev dv;
    int rc = md_dev_wait(h, slot, &dv);
    if(rc) return rc;
    size_t nn = (size_t)dv.n_slots, nt = (size_t)(s->ntiles > 0 ? s->ntiles : 1);
    int64_t nsites = 0;
    if(s->h_site.need(nn + 1) || s->h_sorted.need(nn + 1) || s->h_seg.need(nt)) return MDK_ERR_NOMEM;
    if(h->variant && (s->h_var.need(nn + 1) || s->h_vsorted.need(nn + 1))) return MDK_ERR_NOMEM;
    if(nn) {
        {
            ProfScope pf(PF_DL_COPY);
            HIPCHK(hipMemcpyAsync(s->h_site.p, dv.d_site, nn * sizeof(md_site), hipMemcpyDeviceToHost, s->stream));
            if(h->variant) HIPCHK(hipMemcpyAsync(s->h_var.p, dv.d_var, nn * sizeof(md_site_var), hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipMemcpyAsync(s->h_seg.p, dv.d_seg, (size_t)s->ntiles * sizeof(md_tile_seg), hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        ProfScope pf2(PF_DL_ORDER);
        nsites = md_sites_order(s->h_site.p, h->variant ? s->h_var.p : nullptr, s->h_seg.p, s->ntiles, dv.n_slots, s->h_sorted.p, h->variant ? s->h_vsorted.p : nullptr);
        if(nsites < 0) return fail(MDK_ERR_ARG, "md_dev_download: inconsistent tile segments", hipSuccess);
    }
    s->busy = false;                                   // its stream has been waited for and nothing was queued since
    out->n_sites = nsites; out->site = s->h_sorted.p; out->var = h->variant ? s->h_vsorted.p : nullptr;
    return 0;
}

// the results of one group launch: one wait (the status blocks of all its slots with one copy), then the site arrays of every slot
// queued before a single synchronisation -- a download per slot costs a wait and a round trip each, ~1 ms per chunk
extern "C" int md_dev_download_group(md_dev *h, const int *slots, int n, md_sites *out, int *rcs) {
    if(!h || !slots || !out || !rcs || n < 1 || n > MAXM) return fail(MDK_ERR_ARG, "md_dev_download_group", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    int lo = 0x7fffffff, hi = -1; hipStream_t st = nullptr; Slot *ss[MAXM]; int64_t cnt[MAXM];
    for(int i = 0; i < n; i++) {
        Slot *s = get_slot(h, slots[i]); if(!s || !s->launched) return fail(MDK_ERR_ARG, "md_dev_download_group: slot not launched", hipSuccess);
        ss[i] = s; memset(&out[i], 0, sizeof(out[i])); rcs[i] = 0;
        if(i == 0) st = s->run; else if(s->run != st) st = nullptr;
        lo = std::min(lo, s->index); hi = std::max(hi, s->index);
    }
    if(!st) {      // not one launch: slot by slot
        for(int i = 0; i < n; i++) rcs[i] = md_dev_download(h, slots[i], &out[i]);
        return 0;
    }
    static std::atomic<int> first_call{1}; const bool first = mdk_prof_on() && first_call.exchange(0); const double tf0 = first ? mdk_now() : 0;
    bool all_packed = true; for(int i = 0; i < n; i++) all_packed = all_packed && ss[i]->packed;
    {
        ProfScope pf(PF_FIN_WAIT);
        // (a packed group's status blocks were written into the pinned mirror by k_sites_pack: the wait is all there is)
        if(!all_packed && hipMemcpyAsync(h->h_status.p + lo, h->d_status.p + lo, sizeof(SlotStatus) * (size_t)(hi - lo + 1), hipMemcpyDeviceToHost, st) != hipSuccess) return fail(MDK_ERR_HIP, "D2H status", hipGetLastError());
        hipError_t e = hipStreamSynchronize(st);
        if(e != hipSuccess) return fail(MDK_ERR_HIP, "hipStreamSynchronize", e);
    }
    if(first) fprintf(stderr, "[mdk hip] the first group's results waited for %.3fs\n", mdk_now() - tf0);
    if(mdk_prof_on() && ss[0]->t_launch > 0) { float ms = 0; if(hipEventElapsedTime(&ms, ss[0]->k0, ss[0]->k1) == hipSuccess) mdk_prof_add(PF_GRP_DEV, ms * 1e-3); else (void)hipGetLastError(); mdk_prof_add(PF_GRP_TURN, mdk_now() - ss[0]->t_launch); ss[0]->t_launch = 0; }
    uint32_t packed_n[MAXM];
    for(int i = 0; i < n; i++) {
        packed_n[i] = (all_packed && ss[i]->packed) ? ss[i]->h_st.p->pad : PACK_NONE;
        cnt[i] = finish_eval(h, ss[i]); if(cnt[i] < 0) rcs[i] = (int)cnt[i];      // (a chunk whose segment array had to grow is prepared and piled up again in there: what was packed is void then)
        if(!ss[i]->packed) packed_n[i] = PACK_NONE;
    }
    {
        ProfScope pf(PF_DL_COPY);
        bool any = false;
        for(int i = 0; i < n; i++) {
            Slot *s = ss[i]; if(rcs[i] || cnt[i] == 0 || packed_n[i] != PACK_NONE) continue;
            const size_t nn = (size_t)cnt[i], nt = (size_t)(s->ntiles > 0 ? s->ntiles : 1);
            if(s->h_site.need(nn + 1) || s->h_sorted.need(nn + 1) || s->h_seg.need(nt) || (h->variant && (s->h_var.need(nn + 1) || s->h_vsorted.need(nn + 1)))) { rcs[i] = MDK_ERR_NOMEM; continue; }
            const md_site *d_site = s->b_site ? s->b_site : s->d_site.p; const md_site_var *d_var = s->b_site ? s->b_var : s->d_var.p; const md_tile_seg *d_seg = s->b_site ? s->b_seg : s->d_seg.p;
            hipStream_t cs = s->run ? s->run : st;
            HIPCHK(hipMemcpyAsync(s->h_site.p, d_site, nn * sizeof(md_site), hipMemcpyDeviceToHost, cs));
            if(h->variant) HIPCHK(hipMemcpyAsync(s->h_var.p, d_var, nn * sizeof(md_site_var), hipMemcpyDeviceToHost, cs));
            HIPCHK(hipMemcpyAsync(s->h_seg.p, d_seg, (size_t)s->ntiles * sizeof(md_tile_seg), hipMemcpyDeviceToHost, cs));
            if(cs != st) HIPCHK(hipStreamSynchronize(cs));
            any = true;
        }
        if(any) HIPCHK(hipStreamSynchronize(st));
    }
    for(int i = 0; i < n; i++) ss[i]->busy = false;   // the stream has been waited for and nothing was queued since (a slot that reported an error included)
    ProfScope pf2(PF_DL_ORDER);
    for(int i = 0; i < n; i++) {
        Slot *s = ss[i]; if(rcs[i]) continue;
        int64_t nsites = 0;
        if(packed_n[i] != PACK_NONE) nsites = (int64_t)packed_n[i];
        else if(cnt[i]) {
            nsites = md_sites_order(s->h_site.p, h->variant ? s->h_var.p : nullptr, s->h_seg.p, s->ntiles, cnt[i], s->h_sorted.p, h->variant ? s->h_vsorted.p : nullptr);
            if(nsites < 0) { rcs[i] = fail(MDK_ERR_ARG, "md_dev_download_group: inconsistent tile segments", hipSuccess); continue; }
        }
        out[i].n_sites = nsites; out[i].site = s->h_sorted.p; out[i].var = h->variant ? s->h_vsorted.p : nullptr;
    }
    return 0;
}

extern "C" int md_dev_sync(md_dev *h) {
    if(!h) return fail(MDK_ERR_ARG, "md_dev_sync", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    for(hipStream_t st : h->streams) HIPCHK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int md_dev_bench(md_dev *h, int slot, int warmup, int iters, md_bench_result *out) {
    Slot *s = get_slot(h, slot);
    if(!s || !out || !s->uploaded || iters < 1) return fail(MDK_ERR_ARG, "md_dev_bench", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    memset(out, 0, sizeof(*out));
    int rc = launch_kernels(h, s, false); if(rc) return rc; s->launched = true;
    int64_t n;
    { md_sites tmp; rc = md_dev_download(h, slot, &tmp); if(rc) return rc; n = tmp.n_sites; }
    for(int i = 0; i < warmup; i++) { rc = launch_kernels(h, s, false); if(rc) return rc; }
    HIPCHK(hipStreamSynchronize(s->stream));
    // (a) one launch at a time, bracketed by events (includes the dispatch latency of a lone launch)
    double tot = 0, pk = 0;
    for(int i = 0; i < iters; i++) {
        float a = 0;
        HIPCHK(hipEventRecord(s->e0, s->stream));
        rc = launch_kernels(h, s, false); if(rc) return rc;
        HIPCHK(hipEventRecord(s->e1, s->stream));
        HIPCHK(hipEventSynchronize(s->e1));
        HIPCHK(hipEventElapsedTime(&a, s->e0, s->e1));
        tot += a;
    }
    // (b) `iters` launches back to back between two events: the sustained per-launch time of the kernel, which is
    // what rocprofv3 --kernel-trace reports as its average duration (plus the ~1.5 us kernel-to-kernel boundary)
    {
        float b = 0;
        HIPCHK(hipEventRecord(s->k0, s->stream));
        for(int i = 0; i < iters; i++) { rc = launch_kernels(h, s, false); if(rc) return rc; }
        HIPCHK(hipEventRecord(s->k1, s->stream));
        HIPCHK(hipEventSynchronize(s->k1));
        HIPCHK(hipEventElapsedTime(&b, s->k0, s->k1));
        pk = (double)b;
    }
    out->ms_total = (float)(tot / iters); out->ms_pileup = (float)(pk / iters);
    out->n_sites = (uint64_t)n;
    // sum over reads [16 + 4 n_cigar + ceil(l/2) + l] + interval length + 8 per site (+8 with nOff/nVariant)
    out->algo_bytes = s->read_bytes + (uint64_t)(s->end - s->beg) + (uint64_t)n * (h->variant ? 16 : 8);
    out->tile = s->tile; out->n_tiles = s->ntiles; out->lds_bytes = s->lds_bytes;
    return 0;
}

// Several uploaded slots (distinct intervals, all resident) launched round robin on ONE stream between two HIP events:
// the per-launch time of the pileup kernel when its inputs stream from HBM (the slots together exceed the 256 MiB
// Infinity Cache) instead of being re-read from cache as in md_dev_bench.
extern "C" int md_dev_bench_rotate(md_dev *h, const int *slots, int n, int per_launch, int warmup, int iters, md_bench_result *out) {
    if(!h || !slots || n < 1 || iters < 1 || !out || per_launch < 1 || per_launch > MAXM || n % per_launch) return fail(MDK_ERR_ARG, "md_dev_bench_rotate", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    memset(out, 0, sizeof(*out));
    uint64_t bytes = 0, sites = 0;
    for(int i = 0; i < n; i++) {
        Slot *s = get_slot(h, slots[i]);
        if(!s || !s->uploaded) return fail(MDK_ERR_ARG, "md_dev_bench_rotate: slot not uploaded", hipSuccess);
        int rc = launch_kernels(h, s, false); if(rc) return rc; s->launched = true;
        md_sites tmp; rc = md_dev_download(h, slots[i], &tmp); if(rc) return rc;
        sites += (uint64_t)tmp.n_sites;
        bytes += s->read_bytes + (uint64_t)(s->end - s->beg) + (uint64_t)tmp.n_sites * (h->variant ? 16 : 8);
    }
    Slot *s0 = get_slot(h, slots[0]);
    const int groups = n / per_launch;
    auto go = [&](int g) -> int { return per_launch == 1 ? launch_kernels(h, get_slot(h, slots[g % n]), false, s0->stream) : launch_group_on(h, slots + (g % groups) * per_launch, per_launch, s0->stream, false); };
    for(int i = 0; i < warmup; i++) { int rc = go(i); if(rc) return rc; }
    HIPCHK(hipStreamSynchronize(s0->stream));
    float ms = 0;
    HIPCHK(hipEventRecord(s0->k0, s0->stream));
    for(int i = 0; i < iters; i++) { int rc = go(i); if(rc) return rc; }
    HIPCHK(hipEventRecord(s0->k1, s0->stream));
    HIPCHK(hipEventSynchronize(s0->k1));
    HIPCHK(hipEventElapsedTime(&ms, s0->k0, s0->k1));
    out->ms_pileup = ms / (float)iters; out->ms_total = out->ms_pileup;
    out->algo_bytes = bytes / (uint64_t)groups; out->n_sites = sites / (uint64_t)groups;       // per launch, averaged over the rotation
    out->tile = s0->tile; out->n_tiles = s0->ntiles * per_launch; out->lds_bytes = s0->lds_bytes;
    return 0;
}

extern "C" int md_dev_debug_effective(md_dev *h, int slot, uint8_t *out_base, uint8_t *out_qual, const uint64_t *out_off) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded || !out_base || !out_qual || !out_off) return fail(MDK_ERR_ARG, "md_dev_debug_effective", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    if(s->n_segs == 0) return 0;
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<md_seg> sg(s->n_segs);
    HIPCHK(hipMemcpy(sg.data(), s->d_seg_in.p, sizeof(md_seg) * s->n_segs, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for(int i = 0; i < s->n_segs; i++) total = std::max<uint64_t>(total, out_off[i] + sg[i].len);
    uint8_t *db = nullptr, *dq = nullptr; uint64_t *doff = nullptr;
    HIPCHK(hipMalloc((void **)&db, total + 1)); HIPCHK(hipMalloc((void **)&dq, total + 1)); HIPCHK(hipMalloc((void **)&doff, sizeof(uint64_t) * s->n_segs));
    HIPCHK(hipMemset(db, 0xff, total + 1)); HIPCHK(hipMemset(dq, 0xff, total + 1));
    HIPCHK(hipMemcpy(doff, out_off, sizeof(uint64_t) * s->n_segs, hipMemcpyHostToDevice));
    KParams P; int rc = fill_kparams(h, s, P); if(rc) return rc;
    int grid = (s->n_segs + WAVES - 1) / WAVES; if(grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_debug_effective, dim3(grid), dim3(WG), 0, s->stream, P, s->n_segs, db, dq, doff);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(out_base, db, total, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out_qual, dq, total, hipMemcpyDeviceToHost));
    (void)hipFree(db); (void)hipFree(dq); (void)hipFree(doff);
    return 0;
}

// Staging memory for the host.  Blocks of several MB are 2 MiB-aligned and offered to the kernel as transparent huge pages (a 45 MB
// slab is 23 page faults for the inflate threads instead of 11,500); the library REGISTERS such a block with the runtime the
// first time an upload reads from it (hipHostRegister of huge-page memory: 4.7 ms per 512 MB on the MI355X box,
// profiles/r03a_pin_probe.json), after which hipMemcpyAsync from it is a DMA at the link's 56 GB/s that costs the submitting
// thread a microsecond -- pageable, the same copy is a 13 GB/s CPU copy on the submitting thread.  Allocation itself never
// touches the HIP runtime, so the inflate threads can fill slabs while the device is still coming up.  A 64-byte header in front
// of the block remembers which kind it is.  md_host_set_pinned(1) (default off in the commands) allocates with hipHostMalloc.
struct HostBlock { char *base; size_t len; int state; };          // state: 0 not registered, 1 being registered (by whoever set it), 2 registered, 3 cannot be
static std::mutex g_blocks_mu; static std::vector<HostBlock> g_blocks;       // sorted by base
static std::condition_variable g_blocks_cv;                          // a block left state 1
static double g_reg_seconds = 0; static uint64_t g_reg_calls = 0, g_reg_bytes = 0;
static bool block_less(const HostBlock &x, const HostBlock &y) { return x.base < y.base; }
static void *plain_alloc(size_t n) {
    void *p = nullptr;
    static const int thp = getenv("MDK_NO_THP") ? 0 : 1;
    size_t len = n;
    if(thp && n >= (4u << 20)) {
        len = (n + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        if(posix_memalign(&p, 2u << 20, len) != 0) return nullptr;
        (void)madvise(p, len, MADV_HUGEPAGE);
        std::lock_guard<std::mutex> lk(g_blocks_mu);
        HostBlock b{(char *)p, len, 0};
        g_blocks.insert(std::upper_bound(g_blocks.begin(), g_blocks.end(), b, block_less), b);
    } else if(posix_memalign(&p, 4096, n) != 0) return nullptr;
    memcpy(p, "MDKMAL", 7);
    return (char *)p + 64;
}
// If `ptr` lies in a huge-page staging block not yet known to the runtime, register the block.  Called before an upload reads from it
// (md_dev_upload_raw, md_piece_submit) and, once the device is open, by the inflate team that has just filled it (md_host_register): the
// lock is given up for the duration of hipHostRegister, and whoever meets a block in the middle of that waits for it.
MDK_HIDDEN void host_block_ensure_registered(const void *ptr) {
    static const int off = getenv("MDK_NO_PIN") ? 1 : 0;
    if(off) return;
    std::unique_lock<std::mutex> lk(g_blocks_mu);
    for(;;) {
        HostBlock key{(char *)ptr, 0, 0};
        auto it = std::upper_bound(g_blocks.begin(), g_blocks.end(), key, block_less);
        if(it == g_blocks.begin()) return;
        --it;
        if((char *)ptr >= it->base + it->len || it->state >= 2) return;
        if(it->state == 1) { g_blocks_cv.wait(lk); continue; }
        char *const base = it->base; const size_t len = it->len; it->state = 1;
        lk.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        bool ok; { MarkScope mk("hipHostRegister"); ok = hipHostRegister(base, len, hipHostRegisterDefault) == hipSuccess; }
        if(!ok) (void)hipGetLastError();                             // a block that cannot be registered is uploaded pageable
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        lk.lock();
        HostBlock k2{base, 0, 0};
        it = std::lower_bound(g_blocks.begin(), g_blocks.end(), k2, block_less);     // (the vector may have moved; the block cannot have gone: md_host_free waits for state 1 to pass)
        if(it != g_blocks.end() && it->base == base) it->state = ok ? 2 : 3;
        g_reg_seconds += dt; g_reg_calls++; g_reg_bytes += len;
        g_blocks_cv.notify_all();
        return;
    }
}
extern "C" void md_host_register(md_dev *h, const void *ptr) { if(h) (void)hipSetDevice(h->device); host_block_ensure_registered(ptr); }
// MDK_HOST_PROFILE: what registering the staging blocks cost
extern "C" void md_host_profile(double *seconds, uint64_t *calls, uint64_t *bytes) { std::lock_guard<std::mutex> lk(g_blocks_mu); if(seconds) *seconds = g_reg_seconds; if(calls) *calls = g_reg_calls; if(bytes) *bytes = g_reg_bytes; }
// every staging block not yet known to the runtime is registered now, by `threads` threads (the caller: a helper thread of the command, once
// the device is up -- the slabs filled while the runtime was still starting would otherwise be registered one by one by the thread that uploads)
extern "C" int md_host_register_all(md_dev *h, int threads) {
    if(h) (void)hipSetDevice(h->device);
    std::vector<char *> todo;
    { std::lock_guard<std::mutex> lk(g_blocks_mu); for(const HostBlock &b : g_blocks) if(b.state == 0) todo.push_back(b.base); }
    if(threads < 1) threads = 1;
    if(threads > 16) threads = 16;
    std::atomic<size_t> next{0};
    auto work = [&]() { if(h) (void)hipSetDevice(h->device); for(;;) { const size_t i = next.fetch_add(1); if(i >= todo.size()) break; host_block_ensure_registered(todo[i]); } };
    std::vector<std::thread> th;
    for(int i = 1; i < threads && (size_t)i < todo.size(); i++) th.emplace_back(work);
    work();
    for(auto &t : th) t.join();
    return (int)todo.size();
}
static std::atomic<int> g_want_pinned{1};
extern "C" void md_host_set_pinned(int on) { g_want_pinned.store(on != 0); }
extern "C" void *md_host_alloc(uint64_t bytes) {
    void *p = nullptr; size_t n = (size_t)bytes + 64;
    static std::once_flag once; static int pinned_ok = 0;          // several chunk workers may be the first caller at the same time
    if(!g_want_pinned.load()) return plain_alloc(n);               // (does not touch the HIP runtime)
    std::call_once(once, [] { int c = 0; pinned_ok = (!getenv("MDK_NO_PIN") && hipGetDeviceCount(&c) == hipSuccess && c > 0) ? 1 : 0; });
    if(pinned_ok) { MarkScope mk("md_host_alloc hipHostMalloc"); if(hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess) { memcpy(p, "MDKPIN", 7); return (char *)p + 64; } }
    return plain_alloc(n);
}
extern "C" void md_host_free(void *q) {
    if(!q) return;
    char *p = (char *)q - 64;
    if(!memcmp(p, "MDKPIN", 7)) { MarkScope mk("md_host_free hipHostFree"); (void)hipHostFree(p); return; }
    {
        std::unique_lock<std::mutex> lk(g_blocks_mu);
        HostBlock key{p, 0, 0};
        for(;;) {
            auto it = std::lower_bound(g_blocks.begin(), g_blocks.end(), key, block_less);
            if(it == g_blocks.end() || it->base != p) break;
            if(it->state == 1) { g_blocks_cv.wait(lk); continue; }          // somebody is registering it (or taking its registration away) right now
            if(it->state == 2) {                                            // the lock is given up while the runtime unpins the block: uploaders looking their blocks up do not queue behind it
                it->state = 1; lk.unlock();
                { MarkScope mk("hipHostUnregister"); (void)hipHostUnregister(p); }
                lk.lock();
                it = std::lower_bound(g_blocks.begin(), g_blocks.end(), key, block_less);
                if(it != g_blocks.end() && it->base == p) g_blocks.erase(it);
                g_blocks_cv.notify_all();
                break;
            }
            g_blocks.erase(it);
            break;
        }
    }
    free(p);
}

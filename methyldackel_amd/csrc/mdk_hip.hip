// mdk_hip.hip -- MI355X (gfx950 / CDNA4) device library for the `MethylDackel extract` hot path.
//
// The reference sweeps a pileup buffer column by column (htslib bam_mplp64_auto driven from
// extract.c:399-493) and, per column, loops over the reads covering it.  Every per-position output is a
// plain sum over (read, aligned base) pairs, so on the GPU the same result is computed READ-parallel:
//
//   k_pileup  one workgroup per tile of reference positions.  The tile's context codes (common.c:49-82,
//             precedence extract.c:407-418) are derived from the resident reference into LDS, the tile's
//             counters live in LDS, and the 4 wavefronts of the workgroup stream the admitted reads that
//             overlap the tile: one 64-lane wavefront per read, lanes laid along the read's M/=/X run so
//             that consecutive lanes touch consecutive seq nibbles / qual bytes (coalesced) and consecutive
//             LDS counters (conflict-free).  Per base: trimming (common.c:137-208) is a predicate on the
//             query index, mate-overlap resolution (overlaps.c:54-119) is evaluated on the fly against the
//             mate's base at the same reference position (nothing is written back, so a launch is
//             idempotent), then getStrand/updateMetrics/isVariant arithmetic (common.c:118-134,
//             extract.c:225-239,420-441) and an LDS atomic.  Tiles are compacted from LDS with wave ballots
//             into a per-tile staging segment: no global atomics, deterministic output.
//   k_scan    exclusive scan of the per-tile site counts.
//   k_gather  packs the staged segments into ascending-position SoA site arrays.
//
// Integer/byte work, HBM-bound: no MFMA anywhere (see DESIGN.md for the roofline accounting).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "mdk_hip.h"

#define WG 256
#define WAVES (WG / 64)
#define DEFAULT_TILE 1024

static thread_local char g_err[512] = "";
static int fail(int code, const char *what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return fail(MDK_ERR_HIP, #call, e_); } while(0)

// ------------------------------------------------------------------------------------------------
// kernel parameters
// ------------------------------------------------------------------------------------------------
struct KParams {
    const md_read_hdr *hdr; const int32_t *mate; const uint8_t *blob;
    const char *ref; int64_t reflen;
    int64_t beg, end; int tile, ntiles, nper;
    const int32_t *tfirst, *tlast;
    uint32_t *spos, *smeth, *sunmeth, *soff, *svar; uint8_t *smeta; uint32_t *tcnt;
    int keepCpG, keepCHG, keepCHH, minPhred;
    int bounds[16], abounds[16];
    int *err;
};

struct RD {               // one read, wave-uniform
    int pos, ncig, lq, strand, flags, lo, hi;
    const uint32_t *cig; const uint8_t *seq, *qual;
};

__device__ __forceinline__ int is_mtype(int op) { return op == 0 || op == 7 || op == 8; }

// kept query-index window [lo,hi) after --OT-style and --nOT-style trimming (common.c:137-208)
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

__device__ __forceinline__ RD load_rd(const KParams &P, int r) {
    RD d; md_read_hdr h = P.hdr[r];
    const uint8_t *pay = P.blob + 4ull * h.off4;
    d.pos = h.pos; d.ncig = h.n_cigar; d.lq = (int)h.l_qseq; d.strand = h.strand; d.flags = h.flags;
    d.cig = (const uint32_t *)pay;
    d.seq = pay + 4 * d.ncig;
    d.qual = d.seq + ((((d.lq + 1) >> 1) + 3) & ~3);
    trim_window(P, d.strand, d.flags & MDK_RF_READ2, d.lq, d.lo, d.hi);
    return d;
}

__device__ __forceinline__ void fetch_bq(const RD &d, int q, int &b, int &ql) {
    if(q < d.lo || q >= d.hi) { b = 15; ql = 0; return; }     // trimmed: base N, qual 0
    uint8_t sb = d.seq[q >> 1];
    b = (q & 1) ? (sb & 15) : (sb >> 4);
    ql = d.qual[q];
}

// the mate's (base, qual) at reference position p, if p falls in an M/=/X run of the mate
__device__ __forceinline__ bool mate_at(const RD &m, int p, int &mb, int &mq) {
    int x = m.pos, y = 0; bool found = false; int q = 0;
    for(int k = 0; k < m.ncig; k++) {
        uint32_t c = m.cig[k]; int op = c & 15, len = (int)(c >> 4);
        if(is_mtype(op)) { if(!found && p >= x && p < x + len) { q = y + (p - x); found = true; } x += len; y += len; }
        else if(op == 1 || op == 4) y += len;
        else if(op == 2 || op == 3) x += len;
    }
    if(found && q < m.lq) fetch_bq(m, q, mb, mq); else found = false;
    return found;
}

// (uint8_t)(q + 0.2*q) as evaluated by the reference on x86-64 (overlaps.c:103,106): floor(6q/5) mod 256.
// md_dev_open checks this identity against the C expression for all 256 values.
__device__ __forceinline__ int boost(int q) { return ((q * 6) / 5) & 255; }

// effective (base, qual) of query base q (reference position p) of read o after trimming and, when the read
// has an overlap-resolution partner, after cust_tweak_overlap_quality (overlaps.c:81-114)
__device__ __forceinline__ void effective_bq(const RD &o, bool hasMate, const RD &m, int p, int q, int &b, int &ql) {
    fetch_bq(o, q, b, ql);
    if(hasMate) {
        int mb, mq;
        if(mate_at(m, p, mb, mq)) {
            bool second = (o.flags & MDK_RF_SECOND) != 0;      // 'a' = earlier in file, 'b' = later
            int ba = second ? mb : b, qa = second ? mq : ql, bb = second ? b : mb, qb = second ? ql : mq;
            if(ba != bb) {
                if(qa > qb && ba != 15) { qa -= qb; qb = 0; }
                else if(qb > qa && bb != 15) { qb -= qa; qa = 0; }
                else { qa = 0; qb = 0; }
            } else {
                if(qa > qb) { qa = boost(qa); qb = 0; }
                else { qb = boost(qb); qa = 0; }
            }
            ql = second ? qb : qa;
        }
    }
}

// context code of reference position p: 0 = not a kept C/G, else 1 + 2*type + isG (type 0 CpG, 1 CHG, 2 CHH)
__device__ __forceinline__ int context_code(const KParams &P, int64_t p) {
    char c = P.ref[p] & 0x5f; int type, isG;
    if(c == 'C') {
        isG = 0;
        if(p + 1 < P.reflen && (P.ref[p + 1] & 0x5f) == 'G') type = 0;
        else if(p + 2 < P.reflen && (P.ref[p + 2] & 0x5f) == 'G') type = 1;
        else type = 2;
    } else if(c == 'G') {
        isG = 1;
        if(p > 0 && (P.ref[p - 1] & 0x5f) == 'C') type = 0;
        else if(p > 1 && (P.ref[p - 2] & 0x5f) == 'C') type = 1;
        else type = 2;
    } else return 0;
    if(type == 0 ? !P.keepCpG : type == 1 ? !P.keepCHG : !P.keepCHH) return 0;
    return 1 + 2 * type + isG;
}

// NB (P.ref[x] & 0x5f) maps 'c'->'C', 'g'->'G' and no other FASTA letter onto C/G.

template <bool VARIANT>
__global__ __launch_bounds__(WG) void k_pileup(const KParams P) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int TILE = P.tile;
    uint32_t *cm = lds, *cu = lds + TILE, *co = lds + 2 * TILE, *cv = lds + 3 * TILE;
    uint8_t *ctx = (uint8_t *)(lds + (VARIANT ? 4 : 2) * TILE);
    __shared__ int wsum[WAVES];

    // XCD-aware tile assignment: workgroup b runs on XCD b%8 (observed dispatch order); give every XCD a
    // contiguous run of tiles so that halo reads and mates are shared through one L2.
    const int b = blockIdx.x;
    const int t = (b & 7) * P.nper + (b >> 3);
    if(t >= P.ntiles) return;
    const int64_t T0 = P.beg + (int64_t)t * TILE;
    const int64_t T1 = (T0 + TILE < P.end) ? T0 + TILE : P.end;
    const int tlen = (int)(T1 - T0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for(int i = tid; i < TILE; i += WG) {
        int64_t p = T0 + i;
        ctx[i] = (i < tlen && p < P.reflen) ? (uint8_t)context_code(P, p) : 0;
        cm[i] = 0; cu[i] = 0;
        if(VARIANT) { co[i] = 0; cv[i] = 0; }
    }
    __syncthreads();

    const int first = P.tfirst[t], last = P.tlast[t];
    for(int r0 = first + wave; r0 < last; r0 += WAVES) {
        const int r = __builtin_amdgcn_readfirstlane(r0);
        RD o = load_rd(P, r);
        if(o.pos >= T1) continue;
        int mi = P.mate[r]; bool hasMate = mi >= 0; RD m = o;
        if(hasMate) { m = load_rd(P, mi); if(((o.strand - m.strand) & 1) != 0) hasMate = false; }
        const bool odd = o.strand & 1;
        int x = o.pos, y = 0;
        for(int k = 0; k < o.ncig; k++) {
            uint32_t c = o.cig[k]; int op = c & 15, len = (int)(c >> 4);
            if(is_mtype(op)) {
                int j0 = (T0 > x) ? (int)(T0 - x) : 0;
                int j1 = ((int64_t)x + len > T1) ? (int)(T1 - x) : len;
                if(o.lq - y < j1) j1 = o.lq - y;               // malformed CIGAR guard
                for(int jb = j0; jb < j1; jb += 64) {
                    int j = jb + lane;
                    if(j < j1) {
                        int p = x + j, q = y + j, li = (int)(p - T0);
                        int cc = ctx[li];
                        if(cc) {
                            bool isG = (cc - 1) & 1;
                            bool callpath = (odd != isG);       // OT/CTOT on a C, OB/CTOB on a G
                            if(callpath) {
                                if(o.strand == 0) atomicExch(P.err, 1);   // reference: assert(strand != 0)
                                int bq, ql; effective_bq(o, hasMate, m, p, q, bq, ql);
                                if(ql >= P.minPhred) {
                                    if(odd) { if(bq == 2) atomicAdd(&cm[li], 1u); else if(bq == 8) atomicAdd(&cu[li], 1u); }
                                    else { if(bq == 4) atomicAdd(&cm[li], 1u); else if(bq == 1) atomicAdd(&cu[li], 1u); }
                                }
                            } else if(VARIANT) {
                                int bq, ql; effective_bq(o, hasMate, m, p, q, bq, ql);
                                if(ql >= P.minPhred) {
                                    atomicAdd(&co[li], 1u);
                                    if(odd ? (bq != 4 && bq != 15) : (bq != 2 && bq != 15)) atomicAdd(&cv[li], 1u);
                                }
                            }
                        }
                    }
                }
                x += len; y += len;
            } else if(op == 1 || op == 4) y += len;
            else if(op == 2 || op == 3) x += len;
            if(x >= T1) break;
        }
    }
    __syncthreads();

    // compaction: positions with any evidence, ascending, into this tile's staging segment
    int base = 0; const size_t seg = (size_t)t * TILE;
    for(int s0 = 0; s0 < tlen; s0 += WG) {
        int i = s0 + tid; bool nz = false; uint32_t vm = 0, vu = 0, vo = 0, vv = 0;
        if(i < tlen) { vm = cm[i]; vu = cu[i]; if(VARIANT) { vo = co[i]; vv = cv[i]; } nz = (vm + vu) > 0 || vo > 0; }
        unsigned long long bal = __ballot(nz);
        int wcnt = __popcll(bal), wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if(lane == 0) wsum[wave] = wcnt;
        __syncthreads();
        int pre = base, tot = 0;
        for(int w = 0; w < WAVES; w++) { int c = wsum[w]; if(w < wave) pre += c; tot += c; }
        if(nz) {
            size_t o = seg + pre + wpre;
            P.spos[o] = (uint32_t)(T0 + i); P.smeth[o] = vm; P.sunmeth[o] = vu; P.smeta[o] = (uint8_t)(ctx[i] - 1);
            if(VARIANT) { P.soff[o] = vo; P.svar[o] = vv; }
        }
        base += tot;
        __syncthreads();
    }
    if(tid == 0) P.tcnt[t] = (uint32_t)base;
}

// exclusive scan of tcnt[0..n) -> toff, total; one workgroup of 1024 threads, chunked
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *tcnt, uint32_t *toff, uint32_t *total, int n) {
    __shared__ uint32_t part[1024];
    int tid = threadIdx.x, per = (n + 1023) / 1024, lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for(int i = lo; i < hi; i++) s += tcnt[i];
    part[tid] = s; __syncthreads();
    for(int d = 1; d < 1024; d <<= 1) { uint32_t v = tid >= d ? part[tid - d] : 0; __syncthreads(); part[tid] += v; __syncthreads(); }
    uint32_t run = part[tid] - s;
    for(int i = lo; i < hi; i++) { toff[i] = run; run += tcnt[i]; }
    if(tid == 1023) *total = part[1023];
}

struct GParams {
    const uint32_t *spos, *smeth, *sunmeth, *soff, *svar; const uint8_t *smeta;
    uint32_t *pos, *meth, *unmeth, *off, *var; uint8_t *meta;
    const uint32_t *tcnt, *toff; int tile, ntiles; int64_t cap;
};
__global__ __launch_bounds__(WG) void k_gather(const GParams G) {
    for(int t = blockIdx.x; t < G.ntiles; t += gridDim.x) {
        uint32_t n = G.tcnt[t], o = G.toff[t]; size_t seg = (size_t)t * G.tile;
        for(uint32_t i = threadIdx.x; i < n; i += WG) {
            if((int64_t)(o + i) >= G.cap) break;
            G.pos[o + i] = G.spos[seg + i]; G.meth[o + i] = G.smeth[seg + i]; G.unmeth[o + i] = G.sunmeth[seg + i];
            G.meta[o + i] = G.smeta[seg + i];
            if(G.off) { G.off[o + i] = G.soff[seg + i]; G.var[o + i] = G.svar[seg + i]; }
        }
    }
}

// test hook: effective base/qual of every query base (one wavefront per read)
__global__ __launch_bounds__(WG) void k_debug_effective(const KParams P, int n_reads, uint8_t *ob, uint8_t *oq, const uint64_t *ooff) {
    int lane = threadIdx.x & 63;
    for(int r0 = blockIdx.x * WAVES + (threadIdx.x >> 6); r0 < n_reads; r0 += gridDim.x * WAVES) {
        int r = __builtin_amdgcn_readfirstlane(r0);
        RD o = load_rd(P, r);
        int mi = P.mate[r]; bool hasMate = mi >= 0; RD m = o;
        if(hasMate) { m = load_rd(P, mi); if(((o.strand - m.strand) & 1) != 0) hasMate = false; }
        uint64_t base = ooff[r];
        int x = o.pos, y = 0;
        for(int k = 0; k < o.ncig; k++) {
            uint32_t c = o.cig[k]; int op = c & 15, len = (int)(c >> 4);
            if(is_mtype(op) || op == 1 || op == 4) {
                for(int j = lane; j < len && y + j < o.lq; j += 64) {
                    int bq, ql;
                    if(is_mtype(op)) effective_bq(o, hasMate, m, x + j, y + j, bq, ql); else fetch_bq(o, y + j, bq, ql);
                    ob[base + y + j] = (uint8_t)bq; oq[base + y + j] = (uint8_t)ql;
                }
                y += len; if(is_mtype(op)) x += len;
            } else if(op == 2 || op == 3) x += len;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the library
// ------------------------------------------------------------------------------------------------
template <typename T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    int need(size_t n) {
        if(n <= cap) return 0;
        if(p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc", e);
        cap = want; return 0;
    }
    void release() { if(p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T> struct HBuf {
    T *p = nullptr; size_t cap = 0;
    int need(size_t n) {
        if(n <= cap) return 0;
        if(p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipHostMalloc", e);
        cap = want; return 0;
    }
    void release() { if(p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Slot {
    hipStream_t stream = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, k0 = nullptr, k1 = nullptr;
    DBuf<md_read_hdr> d_hdr; DBuf<int32_t> d_mate; DBuf<uint8_t> d_blob;
    DBuf<int32_t> d_tfirst, d_tlast; HBuf<int32_t> h_tfirst, h_tlast;
    DBuf<uint32_t> d_spos, d_smeth, d_sunmeth, d_soff, d_svar; DBuf<uint8_t> d_smeta;
    DBuf<uint32_t> d_tcnt, d_toff, d_total;
    DBuf<uint32_t> d_pos, d_meth, d_unmeth, d_off, d_var; DBuf<uint8_t> d_meta;
    HBuf<uint32_t> h_pos, h_meth, h_unmeth, h_off, h_var, h_total; HBuf<uint8_t> h_meta; HBuf<int> h_err;
    DBuf<int> d_err;
    int n_reads = 0, ntiles = 0, tid = -1; int64_t beg = 0, end = 0; uint64_t read_bytes = 0;
    bool uploaded = false, launched = false;
};

struct md_dev {
    int device; md_dev_cfg cfg; int tile, n_slots; bool variant;
    std::vector<Slot> slots;
    std::vector<char *> ref; std::vector<int64_t> reflen;
};

extern "C" const char *md_dev_last_error(void) { return g_err; }

extern "C" int md_dev_count(void) {
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    if(e != hipSuccess) { fail(MDK_ERR_NODEVICE, "hipGetDeviceCount", e); return MDK_ERR_NODEVICE; }
    return n;
}

extern "C" int md_dev_open(int device, const md_dev_cfg *cfg, md_dev **out) {
    if(!cfg || !out) return fail(MDK_ERR_ARG, "md_dev_open", hipSuccess);
    *out = nullptr;
    // the boost identity the kernels rely on, checked against the reference's C expression
    for(int q = 0; q < 256; q++) {
        uint8_t v = (uint8_t)q; v = (uint8_t)(int)(v + 0.2 * v);
        if(v != (uint8_t)(((q * 6) / 5) & 255)) { snprintf(g_err, sizeof(g_err), "boost identity fails at q=%d", q); return MDK_ERR_ARG; }
    }
    int n = md_dev_count();
    if(n <= 0) { if(n == 0) snprintf(g_err, sizeof(g_err), "no HIP device visible"); return MDK_ERR_NODEVICE; }
    if(device < 0 || device >= n) return fail(MDK_ERR_ARG, "md_dev_open: device index", hipSuccess);
    HIPCHK(hipSetDevice(device));
    md_dev *h = new md_dev();
    h->device = device; h->cfg = *cfg;
    h->tile = cfg->tile > 0 ? cfg->tile : DEFAULT_TILE;
    h->tile = (h->tile + WG - 1) / WG * WG;
    if(h->tile > 8192) h->tile = 8192;
    h->n_slots = cfg->n_slots > 0 ? cfg->n_slots : 2;
    h->variant = cfg->minOppositeDepth > 0;
    h->slots.resize(h->n_slots);
    for(auto &s : h->slots) {
        HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&s.e0)); HIPCHK(hipEventCreate(&s.e1)); HIPCHK(hipEventCreate(&s.k0)); HIPCHK(hipEventCreate(&s.k1));
        if(s.d_err.need(1) || s.h_err.need(1) || s.d_total.need(1) || s.h_total.need(1)) return MDK_ERR_NOMEM;
        HIPCHK(hipMemset(s.d_err.p, 0, sizeof(int)));
    }
    *out = h;
    return 0;
}

extern "C" void md_dev_close(md_dev *h) {
    if(!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for(auto &s : h->slots) {
        s.d_hdr.release(); s.d_mate.release(); s.d_blob.release(); s.d_tfirst.release(); s.d_tlast.release(); s.h_tfirst.release(); s.h_tlast.release();
        s.d_spos.release(); s.d_smeth.release(); s.d_sunmeth.release(); s.d_soff.release(); s.d_svar.release(); s.d_smeta.release();
        s.d_tcnt.release(); s.d_toff.release(); s.d_total.release();
        s.d_pos.release(); s.d_meth.release(); s.d_unmeth.release(); s.d_off.release(); s.d_var.release(); s.d_meta.release();
        s.h_pos.release(); s.h_meth.release(); s.h_unmeth.release(); s.h_off.release(); s.h_var.release(); s.h_total.release(); s.h_meta.release();
        s.h_err.release(); s.d_err.release();
        if(s.e0) (void)hipEventDestroy(s.e0); if(s.e1) (void)hipEventDestroy(s.e1); if(s.k0) (void)hipEventDestroy(s.k0); if(s.k1) (void)hipEventDestroy(s.k1);
        if(s.stream) (void)hipStreamDestroy(s.stream);
    }
    for(char *p : h->ref) if(p) (void)hipFree(p);
    delete h;
}

extern "C" int md_dev_tile(const md_dev *h) { return h ? h->tile : MDK_ERR_ARG; }

extern "C" int md_dev_set_reference(md_dev *h, int32_t tid, const char *seq, int64_t len) {
    if(!h || tid < 0 || !seq || len < 0) return fail(MDK_ERR_ARG, "md_dev_set_reference", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    if((size_t)tid >= h->ref.size()) { h->ref.resize(tid + 1, nullptr); h->reflen.resize(tid + 1, 0); }
    if(h->ref[tid]) { (void)hipFree(h->ref[tid]); h->ref[tid] = nullptr; }
    char *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, (size_t)len + 16);
    if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc(reference)", e);
    HIPCHK(hipMemcpy(d, seq, (size_t)len, hipMemcpyHostToDevice));
    h->ref[tid] = d; h->reflen[tid] = len;
    return 0;
}

static Slot *get_slot(md_dev *h, int slot) { if(!h || slot < 0 || slot >= h->n_slots) { fail(MDK_ERR_ARG, "bad slot", hipSuccess); return nullptr; } return &h->slots[slot]; }

extern "C" int md_dev_upload(md_dev *h, int slot, const md_read_batch *b) {
    Slot *s = get_slot(h, slot);
    if(!s || !b || b->n_reads < 0 || b->end < b->beg) return fail(MDK_ERR_ARG, "md_dev_upload", hipSuccess);
    if(b->n_reads && (!b->hdr || !b->rend || !b->mate || !b->blob)) return fail(MDK_ERR_ARG, "md_dev_upload: null array", hipSuccess);
    if(b->tid < 0 || (size_t)b->tid >= h->ref.size() || !h->ref[b->tid]) { snprintf(g_err, sizeof(g_err), "reference for tid %d not uploaded", b->tid); return MDK_ERR_NOREF; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(s->stream));          // the slot's previous contents are being replaced
    const int TILE = h->tile; const int64_t span = b->end - b->beg;
    const int ntiles = (int)((span + TILE - 1) / TILE);
    s->n_reads = b->n_reads; s->ntiles = ntiles; s->tid = b->tid; s->beg = b->beg; s->end = b->end;
    s->uploaded = false; s->launched = false;
    size_t nr = (size_t)b->n_reads, nt = (size_t)(ntiles > 0 ? ntiles : 1);
    if(s->d_hdr.need(nr + 1) || s->d_mate.need(nr + 1) || s->d_blob.need((size_t)b->blob_bytes + 16)) return MDK_ERR_NOMEM;
    if(s->d_tfirst.need(nt) || s->d_tlast.need(nt) || s->h_tfirst.need(nt) || s->h_tlast.need(nt)) return MDK_ERR_NOMEM;
    if(s->d_tcnt.need(nt) || s->d_toff.need(nt)) return MDK_ERR_NOMEM;
    size_t stg = (size_t)nt * TILE;
    if(s->d_spos.need(stg) || s->d_smeth.need(stg) || s->d_sunmeth.need(stg) || s->d_smeta.need(stg)) return MDK_ERR_NOMEM;
    if(h->variant && (s->d_soff.need(stg) || s->d_svar.need(stg))) return MDK_ERR_NOMEM;
    // read index range per tile (reads are coordinate sorted, so each tile sees one contiguous run;
    // reads inside the run that end before the tile are skipped by the kernel)
    for(int t = 0; t < ntiles; t++) { s->h_tfirst.p[t] = 0x7fffffff; s->h_tlast.p[t] = 0; }
    uint64_t rbytes = 0;
    for(int i = 0; i < b->n_reads; i++) {
        const md_read_hdr &hd = b->hdr[i];
        rbytes += 16 + 4ull * hd.n_cigar + (hd.l_qseq + 1) / 2 + hd.l_qseq;
        int64_t lo = hd.pos, hi = b->rend[i];
        if(hi <= lo || hi <= b->beg || lo >= b->end) continue;
        if(lo < b->beg) lo = b->beg; if(hi > b->end) hi = b->end;
        int t0 = (int)((lo - b->beg) / TILE), t1 = (int)((hi - 1 - b->beg) / TILE);
        for(int t = t0; t <= t1; t++) { if(s->h_tfirst.p[t] > i) s->h_tfirst.p[t] = i; s->h_tlast.p[t] = i + 1; }
    }
    for(int t = 0; t < ntiles; t++) if(s->h_tlast.p[t] == 0) s->h_tfirst.p[t] = 0;
    s->read_bytes = rbytes;
    if(nr) {
        HIPCHK(hipMemcpyAsync(s->d_hdr.p, b->hdr, nr * sizeof(md_read_hdr), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_mate.p, b->mate, nr * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_blob.p, b->blob, (size_t)b->blob_bytes, hipMemcpyHostToDevice, s->stream));
    }
    if(ntiles) {
        HIPCHK(hipMemcpyAsync(s->d_tfirst.p, s->h_tfirst.p, nt * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_tlast.p, s->h_tlast.p, nt * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    }
    s->uploaded = true;
    return 0;
}

static void fill_kparams(md_dev *h, Slot *s, KParams &P) {
    memset(&P, 0, sizeof(P));
    P.hdr = s->d_hdr.p; P.mate = s->d_mate.p; P.blob = s->d_blob.p;
    P.ref = h->ref[s->tid]; P.reflen = h->reflen[s->tid];
    P.beg = s->beg; P.end = s->end; P.tile = h->tile; P.ntiles = s->ntiles; P.nper = (s->ntiles + 7) / 8;
    P.tfirst = s->d_tfirst.p; P.tlast = s->d_tlast.p;
    P.spos = s->d_spos.p; P.smeth = s->d_smeth.p; P.sunmeth = s->d_sunmeth.p; P.soff = s->d_soff.p; P.svar = s->d_svar.p; P.smeta = s->d_smeta.p;
    P.tcnt = s->d_tcnt.p;
    P.keepCpG = h->cfg.keepCpG; P.keepCHG = h->cfg.keepCHG; P.keepCHH = h->cfg.keepCHH; P.minPhred = h->cfg.minPhred;
    for(int i = 0; i < 16; i++) { P.bounds[i] = h->cfg.bounds[i]; P.abounds[i] = h->cfg.absoluteBounds[i]; }
    P.err = s->d_err.p;
}

static int launch_kernels(md_dev *h, Slot *s, bool time_pileup) {
    if(s->ntiles > 0) {
        KParams P; fill_kparams(h, s, P);
        size_t lds = (size_t)h->tile * ((h->variant ? 16 : 8) + 1);
        int grid = P.nper * 8;
        if(time_pileup) HIPCHK(hipEventRecord(s->k0, s->stream));
        if(h->variant) hipLaunchKernelGGL(k_pileup<true>, dim3(grid), dim3(WG), lds, s->stream, P);
        else hipLaunchKernelGGL(k_pileup<false>, dim3(grid), dim3(WG), lds, s->stream, P);
        if(time_pileup) HIPCHK(hipEventRecord(s->k1, s->stream));
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s->stream, s->d_tcnt.p, s->d_toff.p, s->d_total.p, s->ntiles);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemsetAsync(s->d_total.p, 0, sizeof(uint32_t), s->stream));
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

extern "C" int md_dev_submit(md_dev *h, int slot, const md_read_batch *b) {
    int rc = md_dev_upload(h, slot, b);
    if(rc) return rc;
    return md_dev_launch(h, slot);
}

// wait for the pileup+scan, read the total, check the error word
static int64_t finish_count(md_dev *h, Slot *s) {
    if(!s->launched) { fail(MDK_ERR_ARG, "slot not launched", hipSuccess); return MDK_ERR_ARG; }
    if(hipMemcpyAsync(s->h_total.p, s->d_total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return fail(MDK_ERR_HIP, "D2H total", hipGetLastError());
    if(hipMemcpyAsync(s->h_err.p, s->d_err.p, sizeof(int), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return fail(MDK_ERR_HIP, "D2H err", hipGetLastError());
    hipError_t e = hipStreamSynchronize(s->stream);
    if(e != hipSuccess) return fail(MDK_ERR_HIP, "hipStreamSynchronize", e);
    if(s->h_err.p[0]) { snprintf(g_err, sizeof(g_err), "Can't determine the strand of a read!"); (void)hipMemset(s->d_err.p, 0, sizeof(int)); return MDK_ERR_STRAND0; }
    return (int64_t)s->h_total.p[0];
}

static int gather_into(md_dev *h, Slot *s, uint32_t *pos, uint32_t *meth, uint32_t *unmeth, uint32_t *off, uint32_t *var, uint8_t *meta, int64_t cap) {
    if(s->ntiles <= 0) return 0;
    GParams G; memset(&G, 0, sizeof(G));
    G.spos = s->d_spos.p; G.smeth = s->d_smeth.p; G.sunmeth = s->d_sunmeth.p; G.soff = s->d_soff.p; G.svar = s->d_svar.p; G.smeta = s->d_smeta.p;
    G.pos = pos; G.meth = meth; G.unmeth = unmeth; G.off = h->variant ? off : nullptr; G.var = h->variant ? var : nullptr; G.meta = meta;
    G.tcnt = s->d_tcnt.p; G.toff = s->d_toff.p; G.tile = h->tile; G.ntiles = s->ntiles; G.cap = cap;
    int grid = s->ntiles < 4096 ? s->ntiles : 4096;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(WG), 0, s->stream, G);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int md_dev_download(md_dev *h, int slot, md_sites *out) {
    Slot *s = get_slot(h, slot);
    if(!s || !out) return fail(MDK_ERR_ARG, "md_dev_download", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    memset(out, 0, sizeof(*out));
    int64_t n = finish_count(h, s);
    if(n < 0) return (int)n;
    size_t nn = (size_t)n;
    if(s->d_pos.need(nn + 1) || s->d_meth.need(nn + 1) || s->d_unmeth.need(nn + 1) || s->d_meta.need(nn + 1)) return MDK_ERR_NOMEM;
    if(s->h_pos.need(nn + 1) || s->h_meth.need(nn + 1) || s->h_unmeth.need(nn + 1) || s->h_meta.need(nn + 1)) return MDK_ERR_NOMEM;
    if(h->variant && (s->d_off.need(nn + 1) || s->d_var.need(nn + 1) || s->h_off.need(nn + 1) || s->h_var.need(nn + 1))) return MDK_ERR_NOMEM;
    if(n) {
        int rc = gather_into(h, s, s->d_pos.p, s->d_meth.p, s->d_unmeth.p, s->d_off.p, s->d_var.p, s->d_meta.p, n);
        if(rc) return rc;
        HIPCHK(hipMemcpyAsync(s->h_pos.p, s->d_pos.p, nn * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(s->h_meth.p, s->d_meth.p, nn * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(s->h_unmeth.p, s->d_unmeth.p, nn * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(s->h_meta.p, s->d_meta.p, nn, hipMemcpyDeviceToHost, s->stream));
        if(h->variant) {
            HIPCHK(hipMemcpyAsync(s->h_off.p, s->d_off.p, nn * 4, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipMemcpyAsync(s->h_var.p, s->d_var.p, nn * 4, hipMemcpyDeviceToHost, s->stream));
        }
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    out->n_sites = n; out->pos = s->h_pos.p; out->nmeth = s->h_meth.p; out->nunmeth = s->h_unmeth.p; out->meta = s->h_meta.p;
    out->noff = h->variant ? s->h_off.p : nullptr; out->nvar = h->variant ? s->h_var.p : nullptr;
    return 0;
}

extern "C" int64_t md_dev_sites_to_device(md_dev *h, int slot, uint32_t *d_pos, uint32_t *d_nmeth, uint32_t *d_nunmeth, uint32_t *d_noff, uint32_t *d_nvar, uint8_t *d_meta, int64_t cap) {
    Slot *s = get_slot(h, slot);
    if(!s || !d_pos || !d_nmeth || !d_nunmeth || !d_meta) return fail(MDK_ERR_ARG, "md_dev_sites_to_device", hipSuccess);
    if(h->variant && (!d_noff || !d_nvar)) return fail(MDK_ERR_ARG, "md_dev_sites_to_device: noff/nvar required", hipSuccess);
    if(hipSetDevice(h->device) != hipSuccess) return MDK_ERR_HIP;
    int64_t n = finish_count(h, s);
    if(n < 0) return n;
    if(n > cap) return fail(MDK_ERR_ARG, "md_dev_sites_to_device: capacity too small", hipSuccess);
    if(n) {
        int rc = gather_into(h, s, d_pos, d_nmeth, d_nunmeth, d_noff, d_nvar, d_meta, cap);
        if(rc) return rc;
        hipError_t e = hipStreamSynchronize(s->stream);
        if(e != hipSuccess) return fail(MDK_ERR_HIP, "hipStreamSynchronize", e);
    }
    return n;
}

extern "C" int md_dev_sync(md_dev *h) {
    if(!h) return fail(MDK_ERR_ARG, "md_dev_sync", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    for(auto &s : h->slots) HIPCHK(hipStreamSynchronize(s.stream));
    return 0;
}

extern "C" int md_dev_bench(md_dev *h, int slot, int warmup, int iters, md_bench_result *out) {
    Slot *s = get_slot(h, slot);
    if(!s || !out || !s->uploaded || iters < 1) return fail(MDK_ERR_ARG, "md_dev_bench", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    memset(out, 0, sizeof(*out));
    // output buffers sized once, outside the timed region
    int rc = launch_kernels(h, s, false); if(rc) return rc; s->launched = true;
    int64_t n = finish_count(h, s); if(n < 0) return (int)n;
    size_t nn = (size_t)n + 1;
    if(s->d_pos.need(nn) || s->d_meth.need(nn) || s->d_unmeth.need(nn) || s->d_meta.need(nn)) return MDK_ERR_NOMEM;
    if(h->variant && (s->d_off.need(nn) || s->d_var.need(nn))) return MDK_ERR_NOMEM;
    for(int i = 0; i < warmup; i++) {
        rc = launch_kernels(h, s, false); if(rc) return rc;
        rc = gather_into(h, s, s->d_pos.p, s->d_meth.p, s->d_unmeth.p, s->d_off.p, s->d_var.p, s->d_meta.p, n); if(rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    double tot = 0, pk = 0;
    for(int i = 0; i < iters; i++) {
        float a = 0, b = 0;
        HIPCHK(hipEventRecord(s->e0, s->stream));
        rc = launch_kernels(h, s, true); if(rc) return rc;
        rc = gather_into(h, s, s->d_pos.p, s->d_meth.p, s->d_unmeth.p, s->d_off.p, s->d_var.p, s->d_meta.p, n); if(rc) return rc;
        HIPCHK(hipEventRecord(s->e1, s->stream));
        HIPCHK(hipEventSynchronize(s->e1));
        HIPCHK(hipEventElapsedTime(&a, s->e0, s->e1));
        if(s->ntiles > 0) HIPCHK(hipEventElapsedTime(&b, s->k0, s->k1));
        tot += a; pk += b;
    }
    out->ms_total = (float)(tot / iters); out->ms_pileup = (float)(pk / iters);
    out->n_sites = (uint64_t)n;
    // sum over reads [16 + 4 n_cigar + ceil(l/2) + l] + interval length + 8 per site (+8 with nOff/nVariant)
    out->algo_bytes = s->read_bytes + (uint64_t)(s->end - s->beg) + (uint64_t)n * (h->variant ? 16 : 8);
    return 0;
}

extern "C" int md_dev_debug_effective(md_dev *h, int slot, uint8_t *out_base, uint8_t *out_qual, const uint64_t *out_off) {
    Slot *s = get_slot(h, slot);
    if(!s || !s->uploaded || !out_base || !out_qual || !out_off) return fail(MDK_ERR_ARG, "md_dev_debug_effective", hipSuccess);
    HIPCHK(hipSetDevice(h->device));
    if(s->n_reads == 0) return 0;
    HIPCHK(hipStreamSynchronize(s->stream));
    // total bytes = out_off[n-1] + l_qseq of the last read; the caller guarantees the layout, so recompute from it
    std::vector<md_read_hdr> hdr(s->n_reads);
    HIPCHK(hipMemcpy(hdr.data(), s->d_hdr.p, sizeof(md_read_hdr) * s->n_reads, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for(int i = 0; i < s->n_reads; i++) total = std::max<uint64_t>(total, out_off[i] + hdr[i].l_qseq);
    uint8_t *db = nullptr, *dq = nullptr; uint64_t *doff = nullptr;
    HIPCHK(hipMalloc((void **)&db, total + 1)); HIPCHK(hipMalloc((void **)&dq, total + 1)); HIPCHK(hipMalloc((void **)&doff, sizeof(uint64_t) * s->n_reads));
    HIPCHK(hipMemset(db, 0xff, total + 1)); HIPCHK(hipMemset(dq, 0xff, total + 1));
    HIPCHK(hipMemcpy(doff, out_off, sizeof(uint64_t) * s->n_reads, hipMemcpyHostToDevice));
    KParams P; fill_kparams(h, s, P);
    int grid = (s->n_reads + WAVES - 1) / WAVES; if(grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_debug_effective, dim3(grid), dim3(WG), 0, s->stream, P, s->n_reads, db, dq, doff);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(out_base, db, total, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out_qual, dq, total, hipMemcpyDeviceToHost));
    (void)hipFree(db); (void)hipFree(dq); (void)hipFree(doff);
    return 0;
}

// Staging memory for the host: pinned when a device is present (so hipMemcpyAsync really is asynchronous),
// ordinary page-aligned memory otherwise (lets the host-side packing logic be exercised on a machine without a
// GPU; nothing is computed there).  A 64-byte header in front of the block remembers which kind it is.
extern "C" void *md_host_alloc(uint64_t bytes) {
    void *p = nullptr; size_t n = (size_t)bytes + 64;
    static int pinned_ok = -1;
    if(pinned_ok < 0) { int c = 0; pinned_ok = (hipGetDeviceCount(&c) == hipSuccess && c > 0) ? 1 : 0; }
    if(pinned_ok && hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess) { memcpy(p, "MDKPIN", 7); return (char *)p + 64; }
    if(posix_memalign(&p, 4096, n) != 0) return nullptr;
    memcpy(p, "MDKMAL", 7);
    return (char *)p + 64;
}
extern "C" void md_host_free(void *q) {
    if(!q) return;
    char *p = (char *)q - 64;
    if(!memcmp(p, "MDKPIN", 7)) (void)hipHostFree(p); else free(p);
}

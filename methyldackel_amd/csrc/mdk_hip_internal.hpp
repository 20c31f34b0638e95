// mdk_hip_internal.hpp -- shared between the translation units of libmdk_hip.so (not part of the C ABI).
#ifndef MDK_HIP_INTERNAL_HPP
#define MDK_HIP_INTERNAL_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <vector>
#include <mutex>
#include "mdk_hip.h"

#ifndef WG
#define WG 512
#endif
#define WAVES (WG / 64)
#define RING 64                    // site counters: launch i uses counter i%RING and clears the next one
#define MAXM 8                     // chunks one launch (preparation or pileup) covers at most

#define MDK_HIDDEN __attribute__((visibility("hidden")))
MDK_HIDDEN char *mdk_err_buf();            // the calling thread's message buffer (512 bytes), what md_dev_last_error returns
#define MDK_ERR_BYTES 512
MDK_HIDDEN int fail(int code, const char *what, hipError_t e);
#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return fail(MDK_ERR_HIP, #call, e_); } while(0)

// MDK_HOST_PROFILE: which of this library's calls into the runtime are under way, per thread -- when one call takes long, what the others were inside
// is printed next to it (the runtime serialises more than its interface says: a hipFree waits for the whole device, registrations hold its memory lock)
MDK_HIDDEN int mdk_mark_begin(const char *what);
MDK_HIDDEN void mdk_mark_end(int slot);
MDK_HIDDEN void mdk_marks_dump(const char *why);
struct MarkScope { int s; MarkScope(const char *w) : s(mdk_mark_begin(w)) {} ~MarkScope() { mdk_mark_end(s); } };
struct TileEnt { int first, last; };     // segments [first,last) of the batch overlap the tile (one contiguous run: the batch is coordinate sorted)

// Device buffers.  A hipMalloc costs the calling thread ~0.2 ms whatever its size, and a slot needs ten buffers the first time it is used
// (two dozen slots: 50 ms of the thread that uploads): buffers below ARENA_MAX are carved out of 1 GiB blocks instead (one hipMalloc per
// block, per device).  Carved memory is handed back only when the last carved buffer of the device is released -- then the blocks are
// reused from their start --, and a buffer that grows takes a new piece: 288 GB of HBM make that a fair price.
#define ARENA_BLOCK (1ull << 30)
#define ARENA_MAX (512ull << 20)      /* (a 96 MB piece's inflated bytes with their headroom are 420 MB) */
MDK_HIDDEN size_t mdk_arena_max();                 // ARENA_MAX, or MDK_ARENA_MAX_MB (experiments)
MDK_HIDDEN void *arena_take(size_t bytes);          // NULL: no room could be made (the caller falls back to hipMalloc)
MDK_HIDDEN void arena_give(void *p);
template <typename T> struct DBuf {
    T *p = nullptr; size_t cap = 0; bool carved = false;
    int need(size_t n) {
        if(n <= cap) return 0;
        release();
        size_t want = n + n / 4 + 64;
        if(want * sizeof(T) < mdk_arena_max()) { p = (T *)arena_take(want * sizeof(T)); if(p) { carved = true; cap = want; return 0; } }
        MarkScope mk("DBuf hipMalloc");
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if(e != hipSuccess) { p = nullptr; return fail(MDK_ERR_NOMEM, "hipMalloc", e); }
        cap = want; return 0;
    }
    void release() { if(p) { if(carved) arena_give(p); else { MarkScope mk("DBuf hipFree"); (void)hipFree(p); } } p = nullptr; cap = 0; carved = false; }
};
// pinned host buffers (results on their way back): a hipHostMalloc costs the calling thread ~1 ms, and every slot needs three the first time
// its results are collected -- small ones are carved out of 32 MiB pinned blocks the same way
#define HARENA_BLOCK (32ull << 20)
#define HARENA_MAX (8ull << 20)
MDK_HIDDEN void *harena_take(size_t bytes);
MDK_HIDDEN void harena_give(void *p);
template <typename T> struct HBuf {
    T *p = nullptr; size_t cap = 0; bool carved = false;
    int need(size_t n) {
        if(n <= cap) return 0;
        release();
        size_t want = n + n / 4 + 64;
        if(want * sizeof(T) < HARENA_MAX) { p = (T *)harena_take(want * sizeof(T)); if(p) { carved = true; cap = want; return 0; } }
        MarkScope mk("HBuf hipHostMalloc");
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if(e != hipSuccess) { p = nullptr; return fail(MDK_ERR_NOMEM, "hipHostMalloc", e); }
        cap = want; return 0;
    }
    void release() { if(p) { if(carved) harena_give(p); else { MarkScope mk("HBuf hipHostFree"); (void)hipHostFree(p); } } p = nullptr; cap = 0; carved = false; }
};

// device chunk preparation (mdk_prep.hip)
// one candidate record of the chunk, at its index in the chunk's record table (file order).  64 bytes, so that k_prep_segs meets everything it
// needs about a read and its mate in four 16-byte loads instead of going back to the record: the name's first 16 bytes (zero-filled past
// its end; nlen = its strlen), the first three CIGAR operations, and the start of the read admitted just before it (what htslib's pileup
// buffer evicts against, mdk_pair_rule.h).  adm = 0: the record was not admitted (filter_func said no) and nothing else in it means anything.
// (perRead's selection keeps the admitted reads compacted instead: rd[a] is the a-th kept read, adm = 1 in all of them.)
#define PREP_PREV_NONE INT32_MIN            // no read was admitted before this one
#define PREP_PREV_UNKNOWN (INT32_MIN + 1)   // the one before it belongs to an earlier workgroup of k_prep_scan: k_prep_segs looks it up
struct alignas(16) PrepRead {
    int32_t pos, rend; uint16_t ncig, flag; uint8_t strand, nlen, adm, lqn;             // quad 0 + quad 1: what pairing looks at in the OTHER reads of a name (lqn = l_read_name)
    uint32_t name[4];
    uint32_t lq; int32_t prev; uint32_t cig[2];                                         // quad 2: what the segments of a read (and of its mate) need
};
// 48 bytes (64 until round 6): where the record's name, CIGAR and sequence lie follows from the record's own offset -- rec_off[i], which
// k_prep_segs loads coalesced -- with l_read_name and n_cigar_op: name at +36, CIGAR behind the name, sequence behind the CIGAR.  The first
// three CIGAR operations travel with the read in cig[] as 21 bits each (length < 2^17 << 4 | operation) under a flag in bit 63 that says they
// are there; a longer operation, or a fourth one, is read where it lies.
static_assert(sizeof(PrepRead) == 48, "PrepRead layout");
struct PrepCounters { uint32_t n_adm, n_segs, malformed, strand0, fallback, max_lq; uint64_t algo_bytes; uint32_t far, pad; };      // max_lq: longest admitted read (mbias sizes its histogram by it)
#define MDK_ERR_PREP_REDO (-100)   // internal: the segment array was enlarged and the preparation re-enqueued

// everything the host reads back after a launch, one block per slot inside ONE device array (and its pinned mirror), so that
// a launch over several slots is collected with a single copy
struct SlotStatus { uint32_t total[RING]; int32_t err; uint32_t pad; PrepCounters pc; };
template <typename T> struct Ref { T *p = nullptr; };        // a view into the status arrays (not owned)

struct Slot {
    hipStream_t stream = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, k0 = nullptr, k1 = nullptr;
    DBuf<md_seg> d_seg_in; DBuf<uint8_t> d_blob;
    DBuf<TileEnt> d_tiles; HBuf<TileEnt> h_tiles;
    DBuf<md_site> d_site; DBuf<md_site_var> d_var; DBuf<md_tile_seg> d_seg; Ref<uint32_t> d_total; Ref<int> d_err;
    HBuf<md_site> h_site, h_sorted; HBuf<md_site_var> h_var, h_vsorted; HBuf<md_tile_seg> h_seg; Ref<SlotStatus> h_st; int index = 0;
    hipStream_t run = nullptr; bool fresh = false;       // run: the stream the latest pileup launch went to; fresh: work queued on `stream` that no launch has been ordered after yet
    DBuf<uint8_t> d_raw; DBuf<uint32_t> d_recoff; HBuf<uint32_t> h_rectab;      /* h_rectab: the host's record tables on their way to d_recoff, pinned */
    DBuf<PrepRead> d_prd; DBuf<uint32_t> d_aidx; HBuf<uint32_t> h_aidx; DBuf<int32_t> d_hnext; Ref<PrepCounters> d_pcnt;
    DBuf<uint8_t> d_zero;              // what a preparation launch starts from zeroed: name table (keys, heads), per-workgroup counts, tickets
    uint32_t hmask = 0; int pr_nrec = 0; uint64_t raw_bytes = 0; bool raw_layout = false; int64_t woff = 0, wlen = 0;
    // where the kernels find the chunk's records and their table: the slot's own copies (d_raw, d_recoff), or -- md_dev_upload_raw_inplace -- the caller's piece:
    // raw_at + rec_at[i] is record i; inplace_delta / inplace_bytes: the range inside the piece (md_dev_read_raw hands back the range alone)
    const uint8_t *raw_at = nullptr; const uint32_t *rec_at = nullptr; uint64_t raw_span = 0; bool inplace = false; uint32_t inplace_delta = 0;
    bool prep_pending = false;         // records uploaded, preparation kernels not yet queued (they go with the launch, several chunks at a time)
    bool mb_pending = false;           // mbias: the chunk's preparation is queued, its histogram kernel not yet (it waits for what the preparation reports: the longest read)
    DBuf<md_pr_read> d_pr; DBuf<uint32_t> d_cig; DBuf<md_pr_count> d_prc; HBuf<md_pr_count> h_prc; int pr_n = -1;      // perRead
    // caller-bound output (device memory owned by the caller)
    md_site *b_site = nullptr; md_site_var *b_var = nullptr; md_tile_seg *b_seg = nullptr; int64_t b_cap_sites = 0, b_cap_tiles = 0;
    int n_segs = 0, n_reads = 0, ntiles = 0, tid = -1, tile = 0, lds_bytes = 0; int64_t beg = 0, end = 0; uint64_t read_bytes = 0;
    bool uploaded = false, launched = false; unsigned ring = 0;
    double t_launch = 0;               // MDK_HOST_PROFILE: when the group launch this slot led was issued
    bool packed = false;               // the latest launch was a group launch that also put the slot's ordered sites and status into pinned host memory (k_sites_pack)
    bool busy = false;                 // work of this slot may still be running on its stream (uploaded or launched, results not collected yet): the next upload waits for the stream first
};

struct md_dev {
    int device; md_dev_cfg cfg; int tile, n_slots; bool variant; bool qw = false;
    std::vector<hipStream_t> streams;        // the streams the slots work on (cfg.n_streams of them, or one per slot)
    std::mutex crc_mu; void *d_crc = nullptr;   // constants of k_crc32 (mdk_inflate.hip), made by the first piece
    std::mutex piece_mu; std::vector<hipStream_t> piece_streams; int piece_rr = 0; hipStream_t piece_in = nullptr, piece_inf = nullptr;      /* the pieces' lanes (mdk_inflate.hip): the stream their compressed bytes cross the link on, the stream k_inflate runs on */      // the pieces' streams: a few, shared (mdk_inflate.hip piece_stream_of)
    hipStream_t ref_stream = nullptr;           // md_dev_set_reference works here, so that it neither waits for nor holds up the slots' streams    /* qw: dense contexts, a quarter of a wavefront per segment */
    std::vector<Slot> slots;
    std::vector<char *> ref; std::vector<uint8_t *> refcode; std::vector<int64_t> reflen; std::vector<char> ref_carved;      /* ref_carved[tid]: the contig's two arrays came out of the carved blocks */
    DBuf<SlotStatus> d_status; HBuf<SlotStatus> h_status;
    md_prep_cfg prep; bool prep_set = false; std::vector<uint32_t *> mapbits; std::vector<int64_t> maplen;
    std::vector<md_region *> d_runs; std::vector<int64_t> n_runs; std::vector<char> has_runs;       // -l runs kept for the read prefilter
    uint32_t *d_hist = nullptr; int hist_cap = 0, hist_len = 0; std::vector<uint32_t> h_hist;      // mbias: rows [q][16], q < hist_cap
    std::atomic<bool> scan_wide{getenv("MDK_SCAN_WIDE") && atoi(getenv("MDK_SCAN_WIDE")) > 0}; bool scan_wide_fixed = getenv("MDK_SCAN_WIDE") != nullptr;      // k_prep_scan's windows: wide for libraries with long read names (prep_outcome decides from the first chunks unless the environment has)
};


// perRead (perRead.c:38-94): the walk of one read over its CIGAR, shared by the kernel over host-built batches (k_perread) and the
// one over device-prepared records (k_perread_raw).  `cig(k)` returns CIGAR word k.  What it does after a base below -p -- it steps
// one base on and evaluates that base without looking at its quality or at the CIGAR again, which can run one element past the
// sequence -- is the reference's; that element is what the BAM record holds there (padding nibble of the last sequence byte, or
// the high nibble of the first quality byte).  CpG context comes from the resident context codes, clipped to the window the command
// fetches for the chunk ([max(beg-2,0), end+10000], perRead.c:176): past `wend` nothing is a CpG, and a C at `wend` is not one.
__device__ __forceinline__ int pr_cigar_type(uint32_t op) { return (0x3C1A7u >> ((op & 15) << 1)) & 3; }    // M I D N S H P = X (B: 0): bit 0 query, bit 1 reference
template <typename CigarAt>
__device__ __forceinline__ md_pr_count perread_walk(const uint8_t *seq, const uint8_t *qual, uint32_t l_qseq, int n_cigar, int32_t pos, bool odd,
                                                    const uint8_t *ctxcode, int64_t reflen, int64_t wend, int minPhred, CigarAt cig) {
    uint32_t rp = 0, mp = (uint32_t)pos, nm = 0, nu = 0; int k = 0, off = 0;
    while(rp < l_qseq && k < n_cigar) {
        if(off >= (int)(cig(k) >> 4)) { off = 0; k++; }
        if(k >= n_cigar) break;
        const uint32_t c = cig(k); const int type = pr_cigar_type(c);
        if(type & 2) {
            if(type & 1) {
                if((int)qual[rp] < minPhred) { mp++; rp++; off++; }
                int dir = 0;
                if((int64_t)mp <= wend && (int64_t)mp < reflen) {
                    const int code = ctxcode[mp] & 15;
                    if(code == 1) dir = ((int64_t)mp == wend) ? 0 : 1;       // C of a CpG
                    else if(code == 2) dir = -1;                              // G of a CpG
                }
                if(dir) {
                    int b;
                    if(rp < l_qseq) b = (seq[rp >> 1] >> ((~rp & 1) << 2)) & 15;
                    else if(l_qseq & 1) b = seq[rp >> 1] & 15;
                    else b = (qual[0] >> 4) & 15;
                    if(dir == 1 && odd) { if(b == 2) nm++; else if(b == 8) nu++; }
                    else if(dir == -1 && !odd) { if(b == 4) nm++; else if(b == 1) nu++; }
                }
                mp++; rp++; off++;
            } else { mp += c >> 4; k++; off = 0; }
        } else if(type & 1) { rp += c >> 4; k++; off = 0; }
        else { off = 0; k++; }
    }
    md_pr_count o; o.nmeth = nm; o.nunmeth = nu;
    return o;
}

// MDK_HOST_PROFILE=1: where the host threads' time inside the library goes (seconds and calls per site), printed by md_dev_profile_dump
enum { PF_UP_SYNC = 0, PF_UP_ALLOC, PF_UP_COPY, PF_LAUNCH, PF_FIN_WAIT, PF_DL_COPY, PF_DL_ORDER, PF_SETREF, PF_PIECE_SUBMIT, PF_PIECE_WAIT, PF_GRP_DEV, PF_GRP_TURN, PF_N };
MDK_HIDDEN bool mdk_prof_on();
MDK_HIDDEN void mdk_prof_add(int site, double seconds);
MDK_HIDDEN double mdk_now();
struct ProfScope { int site; double t0; ProfScope(int s) : site(s), t0(mdk_prof_on() ? mdk_now() : 0.0) {} ~ProfScope() { if(mdk_prof_on()) mdk_prof_add(site, mdk_now() - t0); } };
MDK_HIDDEN Slot *get_slot(md_dev *h, int slot);
MDK_HIDDEN void host_block_ensure_registered(const void *ptr);      // a huge-page staging block is registered with the runtime at its first upload
MDK_HIDDEN int launch_kernels(md_dev *h, Slot *s, bool time_pileup, hipStream_t on = nullptr);
MDK_HIDDEN int launch_group_on(md_dev *h, const int *slots, int n, hipStream_t on, bool cross_sync);
MDK_HIDDEN int64_t finish_count(md_dev *h, Slot *s);
MDK_HIDDEN int64_t finish_eval(md_dev *h, Slot *s);      // the status block is already on the host
MDK_HIDDEN int finish_group(md_dev *h, const int *slots, int n, int64_t *counts);
MDK_HIDDEN int prep_outcome(md_dev *h, Slot *s);
MDK_HIDDEN int prep_kernels_init();           // mdk_prep.hip: its code object loaded, the scan kernel's LDS limit set (once per process)
MDK_HIDDEN void inflate_kernels_warm();
MDK_HIDDEN hipStream_t mdk_piece_stream_take(int device);      // a low-priority stream for the device inflate (its own pool of hardware queues)
MDK_HIDDEN hipStream_t mdk_stream_take(int device);       // a stream made ahead by md_dev_warm, or a new one      // mdk_inflate.hip: its code object loaded
MDK_HIDDEN int enqueue_prep_group(md_dev *h, Slot *const *ss, int n, hipStream_t st);      // preparation kernels of up to MAXM uploaded raw slots, one launch each kernel
#endif

// mdk_comm.hip -- the exchange step of the interval-sharded path, and the resident-input benchmark loop that uses it.
//
// `extract` shards by interval with no data dependence between intervals: chunk k of the reference's schedule
// (extract.c:325-350) belongs to GPU k mod N, and the only exchange is the per-interval site buffers travelling to the GPU
// whose host writes the files -- a gather with no reduction, done with ncclSend/ncclRecv groups over xGMI.  The reference has
// no counterpart (its workers share one address space and write through outputMutex, extract.c:514-535).
//
// RCCL is loaded with dlopen the first time a communicator is asked for: a single-GPU run never pays for loading it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>
#include "mdk_hip_internal.hpp"

struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr; decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr; decltype(&ncclSend) Send = nullptr; decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr; decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
};
static Rccl *rccl() {
    static Rccl R; static int state = 0;       // 0 untried, 1 loaded, -1 unavailable
    if(state == 0) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for(const char *n : names) { R.so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if(R.so) break; }
        state = -1;
        if(R.so) {
#define SYM(f) R.f = (decltype(R.f))dlsym(R.so, "nccl" #f)
            SYM(GetUniqueId); SYM(CommInitRank); SYM(CommInitAll); SYM(CommDestroy); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString); SYM(AllReduce);
#undef SYM
            if(R.GetUniqueId && R.CommInitRank && R.CommInitAll && R.CommDestroy && R.Send && R.Recv && R.GroupStart && R.GroupEnd && R.GetErrorString && R.AllReduce) state = 1;
        }
    }
    if(state != 1) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "RCCL (librccl.so.1) could not be loaded: %s", dlerror() ? dlerror() : "symbols missing"); return nullptr; }
    return &R;
}
static int nfail(Rccl *R, const char *what, ncclResult_t r) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "%s: %s", what, R->GetErrorString(r)); return MDK_ERR_HIP; }
#define NCHK(call) do { ncclResult_t r_ = (call); if(r_ != ncclSuccess) return nfail(R, #call, r_); } while(0)

// A communicator over `world` ranks, root 0: one process per GPU, so this process drives exactly one of them (n_local == 1;
// the vectors are what is left of a one-process-many-devices mode that the process-per-GPU command replaced).
struct CommStage { DBuf<uint8_t> d; HBuf<uint8_t> raw; HBuf<md_site> site; HBuf<md_site_var> var; };      // where rank 0 receives one (rank, slot)'s result
struct md_comm {
    int world = 0, n_local = 0;
    std::vector<int> rank; std::vector<md_dev *> dev; std::vector<ncclComm_t> comm; std::vector<hipStream_t> stream; std::vector<hipEvent_t> ev;
    std::vector<CommStage> stage; int stage_slots = 0;
    bool ipc = false; md_comm_oob_fn oob = nullptr; void *oob_ctx = nullptr;      // ranks of different processes on one device: IPC mappings instead of RCCL
};

extern "C" int md_comm_unique_id(uint8_t *id) {
    Rccl *R = rccl(); if(!R) return MDK_ERR_NODEVICE;
    static_assert(sizeof(ncclUniqueId) == MD_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u; NCHK(R->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

static int comm_streams(md_comm *c) {
    c->stream.resize(c->n_local); c->ev.resize(c->n_local);
    for(int i = 0; i < c->n_local; i++) {
        HIPCHK(hipSetDevice(c->dev[i]->device));
        HIPCHK(hipStreamCreateWithFlags(&c->stream[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
    }
    return 0;
}

extern "C" int md_comm_open_rank(md_dev *h, int rank, int world, const uint8_t *id, md_comm **out) {
    if(!h || !out || !id || world < 1 || rank < 0 || rank >= world) return fail(MDK_ERR_ARG, "md_comm_open_rank", hipSuccess);
    *out = nullptr;
    Rccl *R = rccl(); if(!R) return MDK_ERR_NODEVICE;
    md_comm *c = new md_comm(); c->world = world; c->n_local = 1; c->rank = {rank}; c->dev = {h}; c->comm.resize(1);
    HIPCHK(hipSetDevice(h->device));
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    ncclResult_t r = R->CommInitRank(&c->comm[0], world, u, rank);
    if(r != ncclSuccess) { delete c; return nfail(R, "ncclCommInitRank", r); }
    int rc = comm_streams(c); if(rc) { delete c; return rc; }
    *out = c;
    return 0;
}

extern "C" int md_comm_open_rank_shared(md_dev *h, int rank, int world, md_comm_oob_fn oob, void *ctx, md_comm **out) {
    if(!h || !out || !oob || world < 1 || rank < 0 || rank >= world) return fail(MDK_ERR_ARG, "md_comm_open_rank_shared", hipSuccess);
    *out = nullptr;
    md_comm *c = new md_comm(); c->world = world; c->n_local = 1; c->rank = {rank}; c->dev = {h}; c->ipc = true; c->oob = oob; c->oob_ctx = ctx;
    int rc = comm_streams(c); if(rc) { delete c; return rc; }
    *out = c;
    return 0;
}

extern "C" void md_comm_close(md_comm *c) {
    if(!c) return;
    for(int i = 0; i < c->n_local; i++) {
        (void)hipSetDevice(c->dev[i]->device);
        if(i < (int)c->stream.size() && c->stream[i]) { (void)hipStreamSynchronize(c->stream[i]); (void)hipStreamDestroy(c->stream[i]); }
        if(i < (int)c->ev.size() && c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    }
    if(!c->stage.empty()) { (void)hipSetDevice(c->dev[0]->device); for(auto &g : c->stage) { g.d.release(); g.raw.release(); g.site.release(); g.var.release(); } }
    if(!c->comm.empty()) { Rccl *R = rccl(); if(R) for(ncclComm_t k : c->comm) if(k) (void)R->CommDestroy(k); }
    delete c;
}

extern "C" int md_comm_world(const md_comm *c) { return c ? c->world : MDK_ERR_ARG; }

// One exchange.  d_send/send_bytes are indexed by LOCAL rank; d_recv/recv_bytes by GLOBAL rank and only read where rank 0 is
// local.  Rank 0's own contribution is copied when d_recv[0] is given and differs from its send buffer.  Asynchronous on the
// communicator's streams: the caller guarantees the send buffers are complete (their kernels have been waited for) and
// calls md_comm_wait before touching either side again.
extern "C" int md_comm_gather(md_comm *c, const void *const *d_send, const uint64_t *send_bytes, void *const *d_recv, const uint64_t *recv_bytes) {
    if(!c || !d_send || !send_bytes) return fail(MDK_ERR_ARG, "md_comm_gather", hipSuccess);
    int root_local = -1;
    for(int i = 0; i < c->n_local; i++) if(c->rank[i] == 0) root_local = i;
    if(root_local >= 0 && (!d_recv || !recv_bytes)) return fail(MDK_ERR_ARG, "md_comm_gather: the root needs receive buffers", hipSuccess);
    {
        Rccl *R = rccl(); if(!R) return MDK_ERR_NODEVICE;
        NCHK(R->GroupStart());
        for(int i = 0; i < c->n_local; i++) {
            if(c->rank[i] == 0 || !send_bytes[i]) continue;
            ncclResult_t r = R->Send(d_send[i], (size_t)send_bytes[i], ncclUint8, 0, c->comm[i], c->stream[i]);
            if(r != ncclSuccess) { (void)R->GroupEnd(); return nfail(R, "ncclSend", r); }
        }
        if(root_local >= 0) {
            for(int r = 1; r < c->world; r++) {
                if(!recv_bytes[r]) continue;
                ncclResult_t q = R->Recv(d_recv[r], (size_t)recv_bytes[r], ncclUint8, r, c->comm[root_local], c->stream[root_local]);
                if(q != ncclSuccess) { (void)R->GroupEnd(); return nfail(R, "ncclRecv", q); }
            }
        }
        NCHK(R->GroupEnd());
        if(root_local >= 0 && d_recv[0] && d_recv[0] != d_send[root_local] && send_bytes[root_local]) {
            HIPCHK(hipSetDevice(c->dev[root_local]->device));
            HIPCHK(hipMemcpyAsync(d_recv[0], d_send[root_local], (size_t)send_bytes[root_local], hipMemcpyDeviceToDevice, c->stream[root_local]));
        }
    }
    for(int i = 0; i < c->n_local; i++) { HIPCHK(hipSetDevice(c->dev[i]->device)); HIPCHK(hipEventRecord(c->ev[i], c->stream[i])); }
    return 0;
}

extern "C" int md_comm_wait(md_comm *c) {
    if(!c) return fail(MDK_ERR_ARG, "md_comm_wait", hipSuccess);
    for(int i = 0; i < c->n_local; i++) { HIPCHK(hipSetDevice(c->dev[i]->device)); HIPCHK(hipEventSynchronize(c->ev[i])); }
    return 0;
}

// ---- one process per GPU: a chunk's result from the rank that computed it to rank 0 (sizes travel out of band first) ----
extern "C" int md_dev_pci_bus_id(const md_dev *h, char *buf, int cap) {
    if(!h || !buf || cap < 16) return fail(MDK_ERR_ARG, "md_dev_pci_bus_id", hipSuccess);
    HIPCHK(hipDeviceGetPCIBusId(buf, cap, h->device));
    return 0;
}
extern "C" int md_comm_result_header(md_dev *h, int slot, md_result_hdr *hdr) {
    if(!h || !hdr) return fail(MDK_ERR_ARG, "md_comm_result_header", hipSuccess);
    memset(hdr, 0, sizeof(*hdr));
    md_sites_dev dv; const int rc = md_dev_wait(h, slot, &dv);
    hdr->rc = rc; hdr->variant = h->variant ? 1 : 0;
    if(!rc) { hdr->n_slots = dv.n_slots; hdr->n_tiles = dv.n_tiles; }
    return 0;
}
static void result_layout(const md_result_hdr *h, size_t &nS, size_t &nV, size_t &nT, size_t &oV, size_t &oT, size_t &tot) {
    nS = (size_t)h->n_slots * sizeof(md_site); nV = h->variant ? (size_t)h->n_slots * sizeof(md_site_var) : 0; nT = (size_t)h->n_tiles * sizeof(md_tile_seg);
    oV = (nS + 255) & ~(size_t)255; oT = (oV + nV + 255) & ~(size_t)255; tot = oT + nT + 256;
}
extern "C" int md_comm_result_send(md_comm *c, int slot, const md_result_hdr *hdr) {
    if(!c || !hdr || c->n_local != 1 || c->comm.empty() || c->rank[0] == 0) return fail(MDK_ERR_ARG, "md_comm_result_send: needs a rank communicator (md_comm_open_rank) and a rank other than 0", hipSuccess);
    if(hdr->rc || !hdr->n_slots) return 0;                     // nothing follows such a header
    Rccl *R = rccl(); if(!R) return MDK_ERR_NODEVICE;
    md_dev *h = c->dev[0]; Slot *s = get_slot(h, slot); if(!s) return MDK_ERR_ARG;
    const md_site *d_site = s->b_site ? s->b_site : s->d_site.p; const md_site_var *d_var = s->b_site ? s->b_var : s->d_var.p; const md_tile_seg *d_seg = s->b_site ? s->b_seg : s->d_seg.p;
    size_t nS, nV, nT, oV, oT, tot; result_layout(hdr, nS, nV, nT, oV, oT, tot);
    HIPCHK(hipSetDevice(h->device));
    NCHK(R->GroupStart());
    ncclResult_t r = R->Send(d_site, nS, ncclUint8, 0, c->comm[0], c->stream[0]);
    if(r == ncclSuccess && nV) r = R->Send(d_var, nV, ncclUint8, 0, c->comm[0], c->stream[0]);
    if(r == ncclSuccess && nT) r = R->Send(d_seg, nT, ncclUint8, 0, c->comm[0], c->stream[0]);
    if(r != ncclSuccess) { (void)R->GroupEnd(); return nfail(R, "ncclSend", r); }
    NCHK(R->GroupEnd());
    HIPCHK(hipEventRecord(c->ev[0], c->stream[0]));
    return 0;
}
extern "C" int md_comm_result_recv(md_comm *c, int src, const md_result_hdr *hdr, md_sites *out) {
    if(!c || !hdr || !out || c->n_local != 1 || c->comm.empty() || c->rank[0] != 0 || src < 1 || src >= c->world) return fail(MDK_ERR_ARG, "md_comm_result_recv: rank 0 of a rank communicator only", hipSuccess);
    memset(out, 0, sizeof(*out));
    if(hdr->rc) return hdr->rc;
    if(!hdr->n_slots) return 0;
    Rccl *R = rccl(); if(!R) return MDK_ERR_NODEVICE;
    md_dev *h = c->dev[0];
    if(c->stage.size() < (size_t)c->world) c->stage.resize((size_t)c->world);
    CommStage &g = c->stage[(size_t)src];
    size_t nS, nV, nT, oV, oT, tot; result_layout(hdr, nS, nV, nT, oV, oT, tot);
    HIPCHK(hipSetDevice(h->device));
    if(g.d.need(tot) || g.raw.need(tot) || g.site.need((size_t)hdr->n_slots + 1) || (hdr->variant && g.var.need((size_t)hdr->n_slots + 1))) return MDK_ERR_NOMEM;
    NCHK(R->GroupStart());
    ncclResult_t r = R->Recv(g.d.p, nS, ncclUint8, src, c->comm[0], c->stream[0]);
    if(r == ncclSuccess && nV) r = R->Recv(g.d.p + oV, nV, ncclUint8, src, c->comm[0], c->stream[0]);
    if(r == ncclSuccess && nT) r = R->Recv(g.d.p + oT, nT, ncclUint8, src, c->comm[0], c->stream[0]);
    if(r != ncclSuccess) { (void)R->GroupEnd(); return nfail(R, "ncclRecv", r); }
    NCHK(R->GroupEnd());
    HIPCHK(hipMemcpyAsync(g.raw.p, g.d.p, oT + nT, hipMemcpyDeviceToHost, c->stream[0]));
    HIPCHK(hipStreamSynchronize(c->stream[0]));
    const int64_t n = md_sites_order((const md_site *)g.raw.p, hdr->variant ? (const md_site_var *)(g.raw.p + oV) : nullptr, (const md_tile_seg *)(g.raw.p + oT), hdr->n_tiles, hdr->n_slots, g.site.p, hdr->variant ? g.var.p : nullptr);
    if(n < 0) return fail(MDK_ERR_ARG, "md_comm_result_recv: inconsistent tile segments", hipSuccess);
    out->n_sites = n; out->site = g.site.p; out->var = hdr->variant ? g.var.p : nullptr;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Resident-input benchmark loop (bench.py): `n` uploaded slots holding different intervals are launched `group` at a time
// (md_dev_launch_group: one kernel over `group` chunks), the groups round robin, two launches in flight -- launch g is
// issued, then launch g-1 is collected (the site count of each of its chunks read back), as extract_main does -- and the
// kernels write straight into a send buffer that travels to rank 0 in one exchange per launch while the next launch is
// computed into the other buffer.
// ------------------------------------------------------------------------------------------------
struct md_bench {
    md_dev *h = nullptr; md_comm *comm = nullptr; std::vector<int> slots; int group = 0, ngroups = 0, world = 1, rank = 0;
    int64_t cap = 0, tcap = 0; size_t E = 0, off_var = 0, off_seg = 0;       // one chunk's region: sites, [var], tile segments
    uint8_t *send[2] = {nullptr, nullptr}; std::vector<uint8_t *> recv[2]; bool pending[2] = {false, false};
    hipStream_t stream = nullptr; hipEvent_t done[2] = {nullptr, nullptr};      // every launch goes to this one stream, followed by the copy of its status blocks and an event
    uint8_t *ipc_dst[2] = {nullptr, nullptr};    // (ipc communicator, rank > 0) rank 0's receive buffers for this rank, mapped here
    int64_t last_g = -1; bool prep = false;      // prep: every launch prepares its chunks again from their resident raw records
};

extern "C" void md_bench_close(md_bench *b) {
    if(!b) return;
    (void)hipSetDevice(b->h->device);
    if(b->comm) (void)md_comm_wait(b->comm);
    if(b->stream) (void)hipStreamSynchronize(b->stream);
    for(int i : b->slots) {
        (void)md_dev_bind_output(b->h, i, nullptr, nullptr, nullptr, 0, 0);
        Slot *s = get_slot(b->h, i);
        if(s && s->run == b->stream) s->run = nullptr;          // the launch stream goes away with the loop: the slot is collected on its own stream again
    }
    for(int x = 0; x < 2; x++) { if(b->ipc_dst[x]) (void)hipIpcCloseMemHandle(b->ipc_dst[x]); if(b->send[x]) (void)hipFree(b->send[x]); for(uint8_t *p : b->recv[x]) if(p) (void)hipFree(p); if(b->done[x]) (void)hipEventDestroy(b->done[x]); }
    if(b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    delete b;
}

extern "C" int md_bench_open(md_dev *h, md_comm *comm, const int *slots, int n, int group, md_bench **out) {
    if(!h || !slots || group < 1 || group > md_dev_group_max() || n < 2 * group || n % group || !out) return fail(MDK_ERR_ARG, "md_bench_open: needs at least two groups of uploaded slots", hipSuccess);
    if(comm && comm->n_local != 1) return fail(MDK_ERR_ARG, "md_bench_open: one process per GPU", hipSuccess);
    *out = nullptr;
    HIPCHK(hipSetDevice(h->device));
    md_bench *b = new md_bench(); b->h = h; b->comm = comm; b->group = group; b->ngroups = n / group;
    if(comm) { b->world = comm->world; b->rank = comm->rank[0]; }
    for(int i = 0; i < n; i++) {
        Slot *s = get_slot(h, slots[i]);
        if(!s || !s->uploaded) { delete b; return fail(MDK_ERR_ARG, "md_bench_open: slot not uploaded", hipSuccess); }
        int rc = md_dev_bind_output(h, slots[i], nullptr, nullptr, nullptr, 0, 0); if(rc) { delete b; return rc; }
        rc = launch_kernels(h, s, false); if(rc) { delete b; return rc; } s->launched = true;
        int64_t c = finish_count(h, s); if(c < 0) { delete b; return (int)c; }
        if(c > b->cap) b->cap = c;
        if(s->ntiles > b->tcap) b->tcap = s->ntiles;
        b->slots.push_back(slots[i]);
    }
    if(comm && comm->ipc) {      // the same agreement through the caller's out-of-band all-gather
        std::vector<long long> all((size_t)2 * comm->world); long long mine[2] = {(long long)b->cap, (long long)b->tcap};
        if(comm->oob(comm->oob_ctx, mine, all.data(), sizeof(mine))) { delete b; return fail(MDK_ERR_ARG, "md_bench_open: out-of-band all-gather failed", hipSuccess); }
        for(int r = 0; r < comm->world; r++) { b->cap = std::max<int64_t>(b->cap, all[2 * r]); b->tcap = std::max<int64_t>(b->tcap, all[2 * r + 1]); }
    }
    if(comm && !comm->ipc && !comm->comm.empty()) {
        // every rank sends one region per chunk and rank 0 posts a receive of the same size for it: the region must be sized by
        // the largest site count / tile count of ANY rank (the ranks hold different intervals), so the ranks agree on it here
        Rccl *R = rccl(); if(!R) { delete b; return MDK_ERR_NODEVICE; }
        long long hv[2] = {(long long)b->cap, (long long)b->tcap}, *dv = nullptr;
        hipError_t e = hipMalloc((void **)&dv, sizeof(hv));
        if(e == hipSuccess) e = hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice);
        if(e != hipSuccess) { if(dv) (void)hipFree(dv); delete b; return fail(MDK_ERR_NOMEM, "md_bench_open: capacity exchange", e); }
        ncclResult_t r = R->AllReduce(dv, dv, 2, ncclInt64, ncclMax, comm->comm[0], comm->stream[0]);
        if(r == ncclSuccess) { e = hipStreamSynchronize(comm->stream[0]); if(e == hipSuccess) e = hipMemcpy(hv, dv, sizeof(hv), hipMemcpyDeviceToHost); }
        (void)hipFree(dv);
        if(r != ncclSuccess) { delete b; return nfail(R, "ncclAllReduce(max of the ranks' site and tile counts)", r); }
        if(e != hipSuccess) { delete b; return fail(MDK_ERR_HIP, "md_bench_open: capacity exchange", e); }
        b->cap = (int64_t)hv[0]; b->tcap = (int64_t)hv[1];
    }
    HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    for(int x = 0; x < 2; x++) HIPCHK(hipEventCreateWithFlags(&b->done[x], hipEventDisableTiming));
    b->prep = true; for(int i : b->slots) if(!get_slot(h, i)->raw_layout) b->prep = false;
    b->cap += 64; b->tcap += 1;
    b->off_var = (size_t)b->cap * sizeof(md_site);
    b->off_seg = b->off_var + (h->variant ? (size_t)b->cap * sizeof(md_site_var) : 0);
    b->E = (b->off_seg + (size_t)b->tcap * sizeof(md_tile_seg) + 255) & ~(size_t)255;
    for(int x = 0; x < 2; x++) {
        hipError_t e = hipMalloc((void **)&b->send[x], b->E * (size_t)group);
        if(e == hipSuccess) e = hipMemset(b->send[x], 0, b->E * (size_t)group);
        if(e != hipSuccess) { md_bench_close(b); return fail(MDK_ERR_NOMEM, "hipMalloc(bench send buffer)", e); }
        if(comm && b->rank == 0) {
            b->recv[x].assign((size_t)b->world, nullptr);
            for(int r = 1; r < b->world; r++) {
                e = hipMalloc((void **)&b->recv[x][r], b->E * (size_t)group);
                if(e == hipSuccess) e = hipMemset(b->recv[x][r], 0, b->E * (size_t)group);
                if(e != hipSuccess) { md_bench_close(b); return fail(MDK_ERR_NOMEM, "hipMalloc(bench receive buffer)", e); }
            }
        }
    }
    HIPCHK(hipDeviceSynchronize());      // the memsets above have run before the bench stream (non-blocking) or a peer touches the buffers
    if(comm && comm->ipc) {      // rank 0 exports its receive buffers, every other rank maps the two that are meant for it
        const int W = comm->world; std::vector<hipIpcMemHandle_t> mine((size_t)2 * W), all((size_t)2 * W * W);
        memset(mine.data(), 0, sizeof(hipIpcMemHandle_t) * mine.size());
        if(b->rank == 0) for(int x = 0; x < 2; x++) for(int r = 1; r < W; r++) { hipError_t e = hipIpcGetMemHandle(&mine[(size_t)x * W + r], b->recv[x][r]); if(e != hipSuccess) { md_bench_close(b); return fail(MDK_ERR_HIP, "hipIpcGetMemHandle", e); } }
        if(comm->oob(comm->oob_ctx, mine.data(), all.data(), sizeof(hipIpcMemHandle_t) * mine.size())) { md_bench_close(b); return fail(MDK_ERR_ARG, "md_bench_open: out-of-band all-gather failed", hipSuccess); }
        if(b->rank > 0) for(int x = 0; x < 2; x++) { hipError_t e = hipIpcOpenMemHandle((void **)&b->ipc_dst[x], all[(size_t)x * W + b->rank], hipIpcMemLazyEnablePeerAccess); if(e != hipSuccess) { md_bench_close(b); return fail(MDK_ERR_HIP, "hipIpcOpenMemHandle", e); } }
    }
    *out = b;
    return 0;
}

extern "C" int64_t md_bench_region_bytes(const md_bench *b) { return b ? (int64_t)b->E : MDK_ERR_ARG; }
extern "C" int md_bench_set_prep(md_bench *b, int on) {
    if(!b) return MDK_ERR_ARG;
    if(on) for(int i : b->slots) if(!get_slot(b->h, i)->raw_layout) return fail(MDK_ERR_ARG, "md_bench_set_prep: the slots hold host-built batches", hipSuccess);
    b->prep = on != 0;
    return 0;
}

static int bench_exchange(md_bench *b, int x) {
    if(!b->comm) return 0;
    if(b->comm->ipc) {           // a device copy into rank 0's buffer, through the mapping; rank 0 has nothing to post
        md_comm *c = b->comm;
        if(b->rank > 0) HIPCHK(hipMemcpyAsync(b->ipc_dst[x], b->send[x], (size_t)b->E * (size_t)b->group, hipMemcpyDeviceToDevice, c->stream[0]));
        HIPCHK(hipEventRecord(c->ev[0], c->stream[0]));
        b->pending[x] = true;
        return 0;
    }
    const void *snd[1] = {b->send[x]}; uint64_t sb[1] = {(uint64_t)b->E * (uint64_t)b->group};
    std::vector<void *> rcv((size_t)b->world, nullptr); std::vector<uint64_t> rb((size_t)b->world, 0);
    if(b->rank == 0) for(int r = 1; r < b->world; r++) { rcv[r] = b->recv[x][r]; rb[r] = sb[0]; }
    int rc = md_comm_gather(b->comm, snd, sb, rcv.data(), rb.data());
    if(rc) return rc;
    b->pending[x] = true;
    return 0;
}
// launch g is finished when its event has fired: its status blocks are on the host by then
static int bench_collect(md_bench *b, int64_t g, int64_t *sites) {
    const int *gs = b->slots.data() + (g % b->ngroups) * b->group;
    HIPCHK(hipEventSynchronize(b->done[g & 1]));
    for(int i = 0; i < b->group; i++) { int64_t c = finish_eval(b->h, get_slot(b->h, gs[i])); if(c < 0) return (int)c; *sites = c; }
    return bench_exchange(b, (int)(g & 1));
}

// `launches` kernel launches, each one pass of the hot path over the next `group` resident intervals; returns when every
// launch has been collected and every exchange has completed.  The caller brackets this call with its barrier + device sync.
// All launches go to one stream, in order, each followed by the copy of its chunks' status blocks and an event; the host
// keeps two launches queued (launch g is issued, then launch g-1 is collected), so the GPU never waits for the host and a
// kernel never shares the GPU with its neighbour -- the durations rocprofv3 reports for it are those of the kernel alone.
extern "C" int md_bench_run(md_bench *b, int64_t launches, md_bench_run_result *out) {
    if(!b || launches < 0 || !out) return fail(MDK_ERR_ARG, "md_bench_run", hipSuccess);
    md_dev *h = b->h; const int K = b->group;
    HIPCHK(hipSetDevice(h->device));
    memset(out, 0, sizeof(*out));
    int64_t sites = 0;
    for(int64_t g = 0; g < launches; g++) {
        const int x = (int)(g & 1);
        if(b->pending[x]) { int rc = md_comm_wait(b->comm); if(rc) return rc; b->pending[0] = b->pending[1] = false; }     // this buffer is about to be overwritten
        const int *gs = b->slots.data() + (g % b->ngroups) * K;
        int lo = 0x7fffffff, hi = -1;
        for(int i = 0; i < K; i++) {
            uint8_t *base = b->send[x] + (size_t)i * b->E;
            int rc = md_dev_bind_output(h, gs[i], base, h->variant ? base + b->off_var : nullptr, base + b->off_seg, b->cap, b->tcap); if(rc) return rc;
            Slot *sl = get_slot(h, gs[i]); const int idx = sl->index; if(idx < lo) lo = idx; if(idx > hi) hi = idx;
            if(b->prep) sl->prep_pending = true;             // the launch below then runs the preparation kernels of its chunks first
        }
        int rc = launch_group_on(h, gs, K, b->stream, true); if(rc) return rc;
        HIPCHK(hipMemcpyAsync(h->h_status.p + lo, h->d_status.p + lo, sizeof(SlotStatus) * (size_t)(hi - lo + 1), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipEventRecord(b->done[x], b->stream));
        if(g) { rc = bench_collect(b, g - 1, &sites); if(rc) return rc; if(b->comm) out->exchanges++; }
    }
    if(launches) {
        int rc = bench_collect(b, launches - 1, &sites); if(rc) return rc;
        if(b->comm) out->exchanges++;
        b->last_g = launches - 1;
    }
    if(b->comm) { int rc = md_comm_wait(b->comm); if(rc) return rc; b->pending[0] = b->pending[1] = false; }
    out->launches = (uint64_t)launches; out->slots_last = (uint64_t)sites; out->bytes_per_exchange = b->comm ? (uint64_t)b->E * (uint64_t)K : 0;
    return 0;
}

// After md_bench_run: the sites the last launch left in the send buffer (ordered by tile) must be the sites md_dev_download
// gives for the same intervals; on rank 0 every peer's regions must hold sites too.  0 = verified.
extern "C" int md_bench_verify(md_bench *b) {
    if(!b || b->last_g < 0) return fail(MDK_ERR_ARG, "md_bench_verify: nothing was run", hipSuccess);
    md_dev *h = b->h; const int K = b->group; const int64_t g = b->last_g;
    HIPCHK(hipSetDevice(h->device));
    const int x = (int)(g & 1); const int *gs = b->slots.data() + (g % b->ngroups) * K;
    std::vector<uint8_t> reg(b->E); std::vector<md_site> ord((size_t)b->cap);
    for(int i = 0; i < K; i++) {
        HIPCHK(hipMemcpy(reg.data(), b->send[x] + (size_t)i * b->E, b->E, hipMemcpyDeviceToHost));
        Slot *s = get_slot(h, gs[i]);
        int64_t got = md_sites_order((const md_site *)reg.data(), nullptr, (const md_tile_seg *)(reg.data() + b->off_seg), s->ntiles, b->cap, ord.data(), nullptr);
        int rc = md_dev_bind_output(h, gs[i], nullptr, nullptr, nullptr, 0, 0); if(rc) return rc;
        rc = launch_kernels(h, s, false); if(rc) return rc; s->launched = true;
        md_sites ref; rc = md_dev_download(h, gs[i], &ref); if(rc) return rc;
        if(got != ref.n_sites || (got && memcmp(ord.data(), ref.site, (size_t)got * sizeof(md_site)))) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "bench: bound-output sites of chunk %d of the last launch differ from md_dev_download (%lld vs %lld)", i, (long long)got, (long long)ref.n_sites); return MDK_ERR_ARG; }
        if(b->comm && b->rank == 0) {
            for(int r = 1; r < b->world; r++) {
                HIPCHK(hipMemcpy(reg.data(), b->recv[x][r] + (size_t)i * b->E, b->E, hipMemcpyDeviceToHost));
                const md_tile_seg *ts = (const md_tile_seg *)(reg.data() + b->off_seg); uint64_t tot = 0;
                for(int64_t t = 0; t < b->tcap; t++) tot += ts[t].cnt;
                if(!tot) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "bench: nothing was received from rank %d", r); return MDK_ERR_ARG; }
            }
        }
    }
    return 0;
}
